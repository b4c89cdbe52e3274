"""Debug aid: per-token error pattern of the fused-MLP levels against the two-kernel form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepchopper_amd import ops
from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict

cfg = HyenaDNAConfig()
sd = synthetic_state_dict(11, cfg)
model = ops.HipModel(sd, cfg)
rng = np.random.default_rng(1024)
B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 5, int(sys.argv[2]) if len(sys.argv) > 2 else 333
ids = rng.integers(7, 12, (B, L)).astype(np.int64); ids[:, -1] = 1
q = rng.integers(1, 41, (B, L)).astype(np.float32); q[:, -1] = 0
q /= np.linalg.norm(q, axis=1, keepdims=True)
d_ids, d_q = ops.to_device(ids), ops.to_device(q)
outs = {}
for lvl in (0, 1, 2):
    model.set_mlp_fusion(lvl)
    outs[lvl] = model.forward(d_ids, d_q).cpu().numpy()
for lvl in (1, 2):
    d = np.abs(outs[lvl] - outs[0]).max(axis=2)
    print("level", lvl, "max", d.max(), "bad tokens", int((d > 1e-3).sum()), "of", d.size)
    for b in range(B):
        bad = np.flatnonzero(d[b] > 1e-3)
        if bad.size:
            print("  read", b, "first bad", bad[:12], "last", bad[-5:], "count", bad.size, "maxerr", d[b].max())
