#!/usr/bin/env python3
"""profiles/rNN_pmc_hbm_traffic.json from the two rocprofv3 PMC passes of tools/measure_round.sh (FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled: gfx950 counts
64-byte units where the formula assumes 32).     python tools/pmc_traffic_json.py gpurun_out/r05 2056448 > profiles/r05_pmc_hbm_traffic.json"""
import csv
import glob
import json
import sys
from collections import defaultdict

tag, tokens = sys.argv[1], int(sys.argv[2])


def per_call(d, counter):
    acc, calls = defaultdict(float), defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            acc[r["Kernel_Name"]] += float(r["Counter_Value"])
            calls[r["Kernel_Name"]].add((f, r["Dispatch_Id"]))
    return {k: acc[k] / len(calls[k]) for k in acc}


fetch, write = per_call(tag + "_pmc_fetch", "FETCH_SIZE"), per_call(tag + "_pmc_write", "WRITE_SIZE")
ALG = {  # algorithmic HBM bytes per token of the kernels of a step
    "mlp_fused_kernel<true, 0>": ("mlp_fused_kernel<true,EPI_BIAS>: MLP + next in_proj", 5136),
    "mlp_fused_kernel<true, 3>": ("mlp_fused_kernel<true,EPI_BIAS_RELU_Q>: MLP + head linear1", 6164),
    "mlp_fused6_kernel<true, 0>": ("mlp_fused6_kernel<true,EPI_BIAS>: MLP + next in_proj, strict bf16x6", 5136),
    "mlp_fused6_kernel<true, 3>": ("mlp_fused6_kernel<true,EPI_BIAS_RELU_Q>: MLP + head linear1, strict bf16x6", 6164),
    "gemm6_kernel<5, false, 1>": ("gemm6_kernel<EPI_HEAD2_FUSED,false,1>: head linear2+3", 8208),
    "gemm6_kernel<6, false, 1>": ("gemm6_kernel<EPI_BIAS_RES_STATS,false,1>: out_proj", 3088),
    "hyena4_kernel<1, false, 0>": ("hyena4_kernel<1,false,0>: warp-specialised block-Toeplitz long convolution (layers 1-3)", 4096),
    "hyena4_kernel<1, true, 0>": ("hyena4_kernel<1,true,0>: the same with the layer-0 token-table gather (reads 1 B/token, writes y)", 1025),
}
out = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 "
                  "--strict-steps 0 --e2e-reads 0 --legs \"\" --no-profile (and the same with --pmc WRITE_SIZE, a separate pass); "
                  "tools/measure_round.sh, summed per kernel by tools/pmc_traffic_json.py",
       "unit_note": "FETCH_SIZE / WRITE_SIZE in KiB per call; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md); it counts L2 "
                    "fills from the fabric (Infinity-Cache hits included)",
       "padded_tokens_per_launch": tokens, "kernels": {}}
for key, (label, alg) in ALG.items():
    f = next((v for k, v in fetch.items() if key in k), None)
    w = next((v for k, v in write.items() if key in k), None)
    if f is None or w is None:
        continue
    b = (2 * f + w) * 1024
    out["kernels"][label] = {"FETCH_SIZE_KiB_per_call": round(f, 1), "WRITE_SIZE_KiB_per_call": round(w, 1), "bytes_per_call": int(b),
                             "bytes_per_token": round(b / tokens, 1), "algorithmic_bytes_per_token": alg,
                             "ratio_to_algorithmic": round(b / tokens / alg, 2)}
print(json.dumps(out, indent=1))
