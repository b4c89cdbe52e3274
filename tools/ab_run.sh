#!/bin/bash
# run on the GPU box: bench every prebuilt variant library (deepchopper_amd/lib/libdc_*.so), print per-stage ms
set -o pipefail
cd "$(dirname "$0")/.."
for lib in deepchopper_amd/lib/libdc_*.so; do
  name=$(basename "$lib" .so)
  for rep in 1 2; do
    if ! DC_HIP_LIB=$PWD/$lib timeout -k 5 200 python bench.py --steps 4 --warmup 2 --batch "${AB_BATCH:-1024}" --cpu-seconds 0 \
          --strict-steps 0 --e2e-reads 0 --legs "" ${AB_ARGS} 2>/dev/null | python tools/bench_stages.py "$name"; then
      echo "$name FAILED (bench.py exit status ${PIPESTATUS[0]})"
    fi
  done
done
