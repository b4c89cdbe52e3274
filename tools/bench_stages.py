#!/usr/bin/env python3
"""Pretty-print the per-stage timings of one bench.py JSON line read from stdin."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read())
s = d["stages"]
print(tag, "reads/s", d["value"], "ms/step", d["ms_per_step"],
      " ".join(f"{k}={v['avg_ms']}" + (f"({v['tflops']}TF)" if "tflops" in v else "") for k, v in s.items()))
