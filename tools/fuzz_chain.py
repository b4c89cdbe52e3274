"""Random batch geometries (1-40 reads, 2-900 tokens, ragged starts): fusion level 2 (chained MLP kernel) against
level 0 (plain GEMM launches) of the same build.  Run on the GPU box: python tools/fuzz_chain.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from deepchopper_amd import ops
from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict
sd = synthetic_state_dict(5)
model = ops.HipModel(sd, HyenaDNAConfig())
rng = np.random.default_rng(123)
worst = 0.0
for it in range(30):
    B = int(rng.integers(1, 41)); L = int(rng.integers(2, 900))
    ids = rng.integers(7, 12, size=(B, L), dtype=np.int64).astype(np.uint8); ids[:, -1] = 1
    q = rng.random((B, L), dtype=np.float32)
    first = rng.integers(0, max(1, L - 1), size=B).astype(np.int32)
    ntok = (L - first).astype(np.int32)
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    outs = []
    for lvl in (2, 0):
        model.set_mlp_fusion(lvl)
        a = model.forward(d_ids, d_q).cpu().numpy()
        b = model.forward_ragged(d_ids, d_q, first, ntok).cpu().numpy()
        outs.append((a, b))
    mask = np.arange(L)[None, :] >= first[:, None]
    e1 = np.abs(outs[0][0] - outs[1][0]).max()
    e2 = np.abs((outs[0][1] - outs[1][1])[mask]).max()
    worst = max(worst, e1, e2)
    assert np.isfinite(outs[0][0]).all() and np.isfinite(outs[0][1][mask]).all()
    assert model.overflow() == 0
print("30 random batch geometries: max |chained - plain| =", worst)
assert worst < 1e-4
