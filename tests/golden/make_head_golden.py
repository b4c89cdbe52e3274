"""Golden vectors for the token-classification head (F3), produced by the REFERENCE's own
`TokenClassificationHead` (deepchopper/models/llm/head.py:39-102), loaded by file path -- torch
only, importable in the build container.  Run once here:

    python tests/golden/make_head_golden.py

The head weights are the head tensors of `deepchopper_amd.weights.synthetic_state_dict(seed=7)`
(too large to commit: 1.3 M parameters); the fixture stores their checksum so a generator drift is
detected, plus the inputs and the reference's outputs.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepchopper_amd.weights import HD, synthetic_state_dict  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_head", "/root/reference/deepchopper/models/llm/head.py")
ref_head = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_head)

sd = synthetic_state_dict(seed=7)
head = ref_head.TokenClassificationHead(256, 2, 1024, 1024, use_identity_layer_for_qual=True, use_qual=True)
with torch.no_grad():
    for name in ("linear1", "linear2", "linear3"):
        getattr(head, name).weight.copy_(torch.from_numpy(sd[HD + f"{name}.weight"]))
        getattr(head, name).bias.copy_(torch.from_numpy(sd[HD + f"{name}.bias"]))
rng = np.random.default_rng(2024)
# `pre` is a residual stream BEFORE the backbone's final LayerNorm; the head's input is ln_f(pre) as the HF
# backbone computes it (torch LayerNorm, eps 1e-5) -- so the vectors pin the build's "final LayerNorm folded
# into linear1 + head" kernels end to end (dc_head_forward), not only the head arithmetic.
pre = (rng.standard_normal((3, 50, 256)) * rng.uniform(0.2, 3.0, (3, 50, 1)) + rng.uniform(-1, 1, (3, 50, 1))).astype(np.float32)
gamma = rng.uniform(0.5, 1.5, 256).astype(np.float32)
beta = (0.1 * rng.standard_normal(256)).astype(np.float32)
x = torch.nn.functional.layer_norm(torch.from_numpy(pre), (256,), torch.from_numpy(gamma), torch.from_numpy(beta), 1e-5).numpy()
q = rng.integers(0, 41, (3, 50)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
with torch.no_grad():
    y = head(torch.from_numpy(x), torch.from_numpy(q)).numpy()
chk = np.array([float(np.sum(sd[HD + f"{n}.weight"].astype(np.float64))) for n in ("linear1", "linear2", "linear3")])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "head_golden.npz"),
                    pre=pre, ln_f_weight=gamma, ln_f_bias=beta, x=x, quals=q.astype(np.float32), logits=y,
                    weight_checksums=chk, n_params=np.array(sum(p.numel() for p in head.parameters())))
print("n_params", sum(p.numel() for p in head.parameters()), y.shape, np.abs(y).mean())
