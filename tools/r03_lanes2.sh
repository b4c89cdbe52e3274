#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "lanes or two_ranks" > gpurun_out/r03_pytest_lanes.log 2>&1; tail -3 gpurun_out/r03_pytest_lanes.log
show() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'tag': '$1', 'lanes': d['config']['lanes'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'single_stream': (d.get('single_stream') or {}).get('value')}))"; }
rocm-smi --showclocks --showpower 2>/dev/null | head -20
timeout -k 10 200 python bench.py --cpu-seconds 0 --e2e-reads 0 --strict-steps 0 2>/dev/null | show default
timeout -k 10 200 python bench.py --cpu-seconds 0 --e2e-reads 0 --strict-steps 0 --warmup 5 2>/dev/null | show warm5
timeout -k 10 200 python bench.py --cpu-seconds 0 --e2e-reads 0 --strict-steps 0 --lanes 4 --steps 20 --warmup 5 2>/dev/null | show flags
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python bench.py --cpu-seconds 0 --e2e-reads 0 --strict-steps 0 2>/dev/null | show hwq8
timeout -k 10 200 python tools/bench_two_streams.py --lanes 1 4 2>/dev/null
timeout -k 10 200 python bench.py --cpu-seconds 0 --e2e-reads 0 --strict-steps 0 --lanes 3 2>/dev/null | show lanes3
