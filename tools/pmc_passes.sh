#!/bin/bash
# rocprofv3 PMC passes over bench.py (kernel-trace only next to --pmc: the pool refuses the other trace domains), one
# counter group per pass, summarised per kernel by tools/pmc_util.py: effective clock, MFMA-pipe busy share, and the
# split of wave cycles (parked / issue-stalled / VALU / LDS).  usage (on the GPU box, from the repo root):
#   bash tools/pmc_passes.sh TAG        -> gpurun_out/TAG_pmc_util.txt
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 4 --warmup 1 --cpu-seconds 0 --strict-steps 0 --e2e-reads 0 --legs \"\" --no-profile"
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_pmc_u$i -o u -- python3 $R/bench.py $ARGS \
    > $O/${TAG}_pmc_u$i.json 2> $O/${TAG}_pmc_u$i.err || echo "FAILED pass $i ($grp)"
done
cd $R
{
  for j in 1 2 3 4; do
    echo "== pass $j"
    python3 tools/pmc_util.py $O/${TAG}_pmc_u$j mlp_fused gemm6_kernel hyena
  done
} > $O/${TAG}_pmc_util.txt 2>&1
find $O/${TAG}_pmc_u? -type f -size +2M -delete
cat $O/${TAG}_pmc_util.txt
