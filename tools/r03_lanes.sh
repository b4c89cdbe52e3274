#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r03_pytest_lanes.log 2>&1; tail -4 gpurun_out/r03_pytest_lanes.log
rm -f gpurun_out/r03_lanes_sweep.jsonl
for K in 1 2 3 4 5 6 8; do
  timeout -k 10 200 python bench.py --lanes $K --steps 20 --warmup 5 --cpu-seconds 0 --e2e-reads 0 --strict-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lanes': d['config']['lanes'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'single_stream': (d.get('single_stream') or {}).get('value'), 'roofline_frac': d['roofline']['frac'], 'whole_step_tflops': d['roofline'].get('whole_step_tflops_timed_region')}))" | tee -a gpurun_out/r03_lanes_sweep.jsonl
done
for B in 3072 4096; do
  timeout -k 10 200 python bench.py --lanes 6 --batch $B --steps 12 --warmup 4 --cpu-seconds 0 --e2e-reads 0 --strict-steps 0 --no-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lanes': d['config']['lanes'], 'batch': d['config']['reads_per_step_per_gpu'], 'value': d['value'], 'ms_per_step': d['ms_per_step']}))" | tee -a gpurun_out/r03_lanes_sweep.jsonl
done
