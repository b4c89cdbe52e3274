#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "history or two_ranks or multi_rank" > gpurun_out/r03_pytest_hist.log 2>&1; tail -3 gpurun_out/r03_pytest_hist.log
timeout -k 10 600 python tools/host_budget.py --reads 200000 --gpus 1 2 4 --threads 2 4 --out gpurun_out/r03_host_budget.json > gpurun_out/r03_host_budget.log 2> gpurun_out/r03_host_budget.err; tail -5 gpurun_out/r03_host_budget.err; grep -c gpus gpurun_out/r03_host_budget.log
for T in 2 4; do
  timeout -k 10 400 python tools/bench_e2e.py --chop --reads 1000000 --container gzip --threads $T > gpurun_out/r03_e2e_chop_1M_t$T.json 2> gpurun_out/r03_e2e_chop_1M_t$T.err; tail -c 700 gpurun_out/r03_e2e_chop_1M_t$T.json; echo
done
