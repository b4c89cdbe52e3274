"""One launch geometry for counter passes over the FFT long convolution: 64 reads of 7 000 tokens (the 2 x 4096 split class) and
64 of 11 000 (3 x 4096), forward only, a few repetitions.  Prints the geometry (algorithmic bytes per launch for DESIGN)."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from deepchopper_amd import ops
from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict

torch.cuda.set_device(0)
model = ops.HipModel(synthetic_state_dict(3), HyenaDNAConfig())
rng = np.random.default_rng(0)
out = {}
for L in (7000, 11000):
    n = 64
    ids = rng.integers(7, 12, (n, L)).astype(np.uint8)
    ids[:, -1] = 1
    q = rng.random((n, L)).astype(np.float32)
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    for _ in range(3):
        model.forward(d_ids, d_q)
    torch.cuda.synchronize()
    M = 8192 if L <= 8192 else 12288
    # per (read, channel): x1, v, x0 rows in (12 B/token), y out (4 B/token), the class's filter spectrum (8 B x (M + 1))
    out[str(L)] = {"reads": n, "tokens": L, "points": M, "algorithmic_bytes_per_launch": n * 256 * (16 * L + 8 * (M + 1)),
                   "of_which_spectrum": n * 256 * 8 * (M + 1)}
print(json.dumps(out))
