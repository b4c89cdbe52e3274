#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_intpaths.py -m gpu -x -q > gpurun_out/r03_pytest_int3.log 2>&1; tail -3 gpurun_out/r03_pytest_int3.log
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_int -o int -- python3 $R/tools/bench_intkernels.py > $R/gpurun_out/r03_int_kernels_under_rocprof.json 2> $R/gpurun_out/r03_prof_int.err
head -4 $R/gpurun_out/r03_prof_int/int_kernel_stats.csv | cut -c1-150
cd $R
timeout -k 10 400 python tools/bench_two_streams.py --lanes 1 2 3 4 6 8 > gpurun_out/r03_two_streams_sweep.json 2> gpurun_out/r03_two_streams.err; cat gpurun_out/r03_two_streams_sweep.json
timeout -k 10 300 python tools/bench_two_streams.py --reads 4096 --lanes 1 4 8 > gpurun_out/r03_two_streams_4096.json 2>> gpurun_out/r03_two_streams.err; cat gpurun_out/r03_two_streams_4096.json
