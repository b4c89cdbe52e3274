#!/usr/bin/env python3
"""Per kernel: mean duration, effective clock (GRBM_GUI_ACTIVE / 8 / duration), MFMA pipe utilisation
(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles) and the wave-cycle split, from one rocprofv3 --pmc pass:
    python tools/pmc_util.py <dir> [name-substring ...]"""
import collections
import csv
import glob
import sys

d0 = sys.argv[1]
subs = sys.argv[2:] or ["mlp_fused", "gemm6_kernel", "hyena3"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
dur = collections.defaultdict(float)
for f in glob.glob(d0 + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in subs):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k in sorted(acc, key=lambda k: -dur[k]):
    c = len(seen[k])
    d = dur[k] / c
    a = {n: v / c for n, v in acc[k].items()}
    line = f"{k[:58]:58s} calls {c:3d} {d / 1e6:7.3f} ms"
    if "GRBM_GUI_ACTIVE" in a:
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        line += f"  clock {cyc / d:5.2f} GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            line += f"  MFMA busy {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:5.1f} %"
    wc = a.get("SQ_WAVE_CYCLES")
    if wc:
        for n, lab in (("SQ_WAIT_ANY", "parked"), ("SQ_WAIT_INST_ANY", "issue-stall"), ("SQ_ACTIVE_INST_VALU", "valu"),
                       ("SQ_ACTIVE_INST_LDS", "lds-inst"), ("SQ_WAIT_INST_LDS", "wait-lds")):
            if n in a:
                line += f"  {lab} {100 * a[n] / wc:4.1f}%"
    print(line)
    if "SQ_LDS_BANK_CONFLICT" in a and a.get("SQ_LDS_IDX_ACTIVE"):
        print(f"    LDS bank-conflict cycles / active cycles: {100 * a['SQ_LDS_BANK_CONFLICT'] / a['SQ_LDS_IDX_ACTIVE']:.1f} %")
