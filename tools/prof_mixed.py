"""Per-stage device time of one mixed-length batch (config 4 shape) through run_device_batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deepchopper_amd import ops, synth
from deepchopper_amd.model import DeepChopper
from deepchopper_amd.predict import PredictPipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 900
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 10000)
seqs, quals, ids = synth.synth_reads_mixed(n, lo, hi, seed=42)
torch.cuda.set_device(0)
pipe = PredictPipeline(DeepChopper.synthetic(seed=42))
sb, offs = ops.pack_reads(seqs)
qb, _ = ops.pack_reads(quals)
lens = np.diff(offs)
d = (ops.to_device(sb), ops.to_device(qb), ops.to_device(offs))
net = pipe.model.net
for _ in range(2):
    pipe.run_device_batch(*d, lens)
torch.cuda.synchronize()
net.set_profiling(True)
net.profile_reset()
for _ in range(3):
    pipe.run_device_batch(*d, lens)
torch.cuda.synchronize()
prof = net.profile()
tot = prof["forward"][0] / 3
print(f"reads {n}, bases {int(lens.sum())}, mean {lens.mean():.0f} bp, max {lens.max()}; forward {tot:.2f} ms -> "
      f"{lens.sum() / tot / 1e3:.2f} Mbases/s")
for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    if cnt and k != "forward":
        print(f"  {k:14s} {ms / 3:8.3f} ms/step  ({100 * ms / 3 / tot:4.1f} %)  {cnt // 3} launches")
