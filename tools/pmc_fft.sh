#!/bin/bash
# The same PMC passes as tools/pmc_passes.sh over the config-4 length mix (tools/bench_mixed.py): where the FFT
# long-convolution kernels' wave cycles go.   bash tools/pmc_fft.sh TAG -> gpurun_out/TAG_pmc_fft.txt
set -u
TAG=${1:-r05b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_pmcf_u$i -o u -- python3 $R/tools/bench_mixed.py 6000 \
    > $O/${TAG}_pmcf_u$i.json 2> $O/${TAG}_pmcf_u$i.err || echo "FAILED pass $i ($grp)"
done
cd $R
{
  for j in 1 2 3 4; do
    echo "== pass $j"
    python3 tools/pmc_util.py $O/${TAG}_pmcf_u$j hyena_fft hyena4
  done
} > $O/${TAG}_pmc_fft.txt 2>&1
find $O/${TAG}_pmcf_u? -type f -size +2M -delete
cat $O/${TAG}_pmc_fft.txt
