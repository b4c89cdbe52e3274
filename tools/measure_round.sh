#!/bin/bash
# One GPU-box call that collects the round's kernel evidence (run from the repo root via gpurun):
#   rocprofv3 kernel stats of the integer-kernel bench and of bench.py, the PMC FETCH_SIZE / WRITE_SIZE passes of
#   bench.py (separate passes, kernel-trace only -- the pool refuses --pmc next to the other trace domains), the
#   long-read FFT before/after, and the plain bench line.  Outputs under gpurun_out/$TAG*.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
step() { echo "== $1" ; }
step int-kernels-rocprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_int -o int -- python3 $R/tools/bench_intkernels.py > $O/${TAG}_int_kernels_under_rocprof.json 2> $O/${TAG}_prof_int.err || echo "FAILED int-kernels-rocprof"
if [ "${SKIP_LONG:-0}" != "1" ]; then
step long-reads
timeout -k 10 400 python3 $R/tools/bench_long_reads.py > $O/${TAG}_long_reads_fft15.jsonl 2> $O/${TAG}_long_reads.err || echo "FAILED long-reads"
fi
step bench-rocprof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_bench -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --e2e-reads 0 --legs "" > $O/${TAG}_bench_under_rocprofv3.json 2> $O/${TAG}_prof_bench.err || echo "FAILED bench-rocprof"
step pmc-fetch
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -o f -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-seconds 0 --strict-steps 0 --e2e-reads 0 --legs "" --no-profile > $O/${TAG}_pmc_fetch.json 2> $O/${TAG}_pmc_fetch.err || echo "FAILED pmc-fetch"
step pmc-write
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -o w -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-seconds 0 --strict-steps 0 --e2e-reads 0 --legs "" --no-profile > $O/${TAG}_pmc_write.json 2> $O/${TAG}_pmc_write.err || echo "FAILED pmc-write"
step ref-legs-rocprof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_ref -o ref -- python3 $R/tools/bench_ref_legs.py 60000 nanopore > $O/${TAG}_ref_nanopore_under_rocprofv3.json 2> $O/${TAG}_prof_ref.err || echo "FAILED ref-legs-rocprof"
step mixed-rocprof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_mixed -o mixed -- python3 $R/tools/bench_mixed.py > $O/${TAG}_config4_mixed_under_rocprofv3.json 2> $O/${TAG}_prof_mixed.err || echo "FAILED mixed-rocprof"
step cli-ref
( cd $R && timeout -k 10 500 python3 tools/bench_cli_ref.py 60000 > $O/${TAG}_cli_reference_mode.jsonl 2> $O/${TAG}_cli_ref.err ) || echo "FAILED cli-ref"
step bench-plain
cd $R
timeout -k 10 500 python3 bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || echo "FAILED bench-plain"
tail -c 600 $O/${TAG}_bench_default.json
python3 tools/pmc_summary.py $O/${TAG}_pmc_fetch mlp_fused gemm6 hyena > $O/${TAG}_pmc_fetch_summary.txt 2>&1
python3 tools/pmc_summary.py $O/${TAG}_pmc_write mlp_fused gemm6 hyena > $O/${TAG}_pmc_write_summary.txt 2>&1
# keep only the small summaries of the rocprof output directories (gpurun_out merges <= 64 MiB)
find $O/${TAG}_prof_int $O/${TAG}_prof_bench $O/${TAG}_prof_ref $O/${TAG}_prof_mixed -name "*kernel_stats.csv" | head
find $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_prof_int $O/${TAG}_prof_bench $O/${TAG}_prof_ref $O/${TAG}_prof_mixed -type f ! -name "*kernel_stats.csv" -size +2M -delete
du -sh $O/${TAG}_* | tail -12
