#!/bin/bash
# run on the GPU box: bench the in-tree library under several environment settings (A/B of runtime switches), two passes each.
# usage: tools/ab_env.sh "name1:VAR=val VAR2=val" "name2:" ...      (AB_BATCH, AB_ARGS as in ab_run.sh)
set -o pipefail
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for spec in "$@"; do
    name="${spec%%:*}"; envs="${spec#*:}"
    if ! env $envs timeout -k 5 200 python bench.py --steps 6 --warmup 2 --batch "${AB_BATCH:-2048}" --cpu-seconds 0 \
          --strict-steps 0 --e2e-reads 0 --legs "" ${AB_ARGS} 2>/dev/null | python tools/bench_stages.py "$name"; then
      echo "$name FAILED (bench.py exit status ${PIPESTATUS[0]})"
    fi
  done
done
