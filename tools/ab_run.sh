#!/bin/bash
# run on the GPU box: bench every prebuilt variant library, print per-stage ms
cd "$(dirname "$0")/.."
for lib in deepchopper_amd/lib/libdc_*.so; do
  name=$(basename $lib .so)
  for rep in 1 2; do
  DC_HIP_LIB=$PWD/$lib timeout -k 5 200 python bench.py --steps 4 --warmup 2 --batch ${AB_BATCH:-1024} --cpu-seconds 0 ${AB_ARGS} 2>/dev/null | python tools/bench_stages.py $name; continue; python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']
print('$name', 'reads/s', d['value'], ' '.join(f\"{k}={v['avg_ms']}\" for k,v in s.items() if k.startswith('gemm') or k in ('hyena','forward')))"
  done
done
