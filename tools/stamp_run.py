#!/usr/bin/env python3
"""Run a few forward steps with a DC_STAMP diagnostic library (DC_HIP_LIB=...) and print, per GEMM epilogue
kind, the average shader cycles a wave spends in prologue / k-loop / epilogue of gemm6_kernel."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deepchopper_amd import _lib, ops, synth
from deepchopper_amd.model import DeepChopper

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.cuda.set_device(0)
model = DeepChopper.synthetic(seed=42)
rng = np.random.default_rng(0)
ids = rng.integers(7, 11, size=(batch, 1001), dtype=np.int64).astype(np.uint8)
ids[:, -1] = 1
q = rng.random((batch, 1001), dtype=np.float32)
d_ids, d_q = ops.to_device(ids), ops.to_device(q)
L = _lib.lib()
for _ in range(2):
    model.net.forward(d_ids, d_q)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 56)()
L.dc_debug_read_stamps(buf)
for _ in range(3):
    model.net.forward(d_ids, d_q)
torch.cuda.synchronize()
L.dc_debug_read_stamps(buf)
names = {0: "in_proj", 2: "fc1(gelu)", 3: "head1", 5: "head2", 6: "out_proj(+fc2)", 7: "hyena3"}
for e, n in names.items():
    pro, loop, epi, waves = buf[4 * e], buf[4 * e + 1], buf[4 * e + 2], buf[4 * e + 3]
    if waves:
        tot = pro + loop + epi
        rt = buf[32 + e]
        if rt:
            print(f"   clock {tot / rt * 100:.0f} MHz (memtime / memrealtime), wave lifetime {rt / waves / 100:.1f} us")
        print(f"{n:14s} waves {waves:9d}  prologue {pro / waves:8.0f} ({100 * pro / tot:4.1f}%)  loop {loop / waves:8.0f} "
              f"({100 * loop / tot:4.1f}%)  epilogue {epi / waves:8.0f} ({100 * epi / tot:4.1f}%)  cycles/wave")

for r, n in enumerate(("hyena4 consumer", "hyena4 Z (z tiles)", "hyena4 Y (outputs)")):
    pro, work, wait, waves = (buf[40 + 4 * r + j] for j in range(4))
    if waves:
        tot = pro + work + wait
        print(f"{n:20s} waves {waves:9d}  prologue {pro / waves:8.0f} ({100 * pro / tot:4.1f}%)  working {work / waves:9.0f} "
              f"({100 * work / tot:4.1f}%)  at the barrier {wait / waves:9.0f} ({100 * wait / tot:4.1f}%)  cycles/wave")
if buf[40 + 15]:
    w = buf[40 + 15]
    print(f"hyena4 Y phases: load-wait + gate {buf[40 + 12] / w:9.0f}  patch-wait {buf[40 + 13] / w:9.0f}  cycles/wave")
