#!/bin/bash
# A/B of the FFT long convolution's size-class sets on the config-4 length mix (tools/bench_mixed.py, stage times on):
# one JSON line per variant.  bit k of DC_FFT_CLASS_MASK = kFftClass[k] (csrc/forward.h):
#   0 1024 | 1 2048 | 2 4096 | 3 8192 (2x4096) | 4 8192 direct | 5 12288 (3x4096) | 6 16384 (4x4096) | 7 16384 direct |
#   8 2^15 four-step
# (the sweep of the first build of the round, which also had 3x1024, 3x2048 and 5x2048 classes: profiles/r05b_fft_class_sweep.jsonl)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-20000}
run() {  # name, env...
  local name=$1; shift
  echo -n "{\"variant\": \"$name\", \"result\": "
  env DC_PROFILE_STAGES=1 "$@" timeout -k 10 200 python3 $R/tools/bench_mixed.py $N | tail -1 | tr -d '\n'
  echo "}"
}
run "power-of-two transforms (rounds 2-4)" DC_FFT_SPLIT=0
run "split classes (default)"
run "split classes, round-2 kernel for 4096 points" DC_FFT_STAGED1=0
run "split 8192 only (12288 / 16384 -> direct 2^14)" DC_FFT_CLASS_MASK=0x18f
run "split without the 12288 class" DC_FFT_CLASS_MASK=0x14f
