"""CPU checks of the forward oracle: the head against golden vectors produced by the reference's
own head.py, the FFT long convolution against its time-domain statement, basic invariants."""
import os

import numpy as np
import torch

from deepchopper_amd.weights import HD, HyenaDNAConfig, required_keys, synthetic_state_dict
from oracle import forward_ref as fr


def test_head_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "head_golden.npz"))
    sd7 = synthetic_state_dict(seed=7)
    chk = np.array([float(np.sum(sd7[HD + f"{n}.weight"].astype(np.float64))) for n in ("linear1", "linear2", "linear3")])
    np.testing.assert_allclose(chk, g["weight_checksums"], rtol=1e-12)
    assert int(g["n_params"]) == 1314818
    tsd = {k: torch.as_tensor(v) for k, v in sd7.items()}
    out = fr.head_forward(tsd, torch.as_tensor(g["x"]), torch.as_tensor(g["quals"])).numpy()
    np.testing.assert_allclose(out, g["logits"], rtol=0, atol=2e-6)


def test_fftconv_equals_direct_causal_convolution():
    torch.manual_seed(0)
    for L in (1, 2, 31, 300):
        v, k, D = torch.randn(2, 6, L), torch.randn(6, L), torch.randn(6)
        np.testing.assert_allclose(fr.fftconv(v, k, D).numpy(), fr.causal_conv_direct(v, k, D).numpy(), atol=5e-5 * max(L, 8) ** 0.5)


def test_forward_shapes_and_causality():
    sd = synthetic_state_dict(0)
    cfg = HyenaDNAConfig()
    assert set(required_keys(cfg)) <= set(sd)
    rng = np.random.default_rng(0)
    ids = rng.integers(7, 12, (2, 40))
    q = rng.random((2, 40)).astype(np.float32)
    a = fr.forward(sd, ids, q).numpy()
    assert a.shape == (2, 40, 2) and np.isfinite(a).all()
    # causal: changing the last token leaves earlier logits unchanged
    ids2 = ids.copy()
    ids2[:, -1] = 1
    b = fr.forward(sd, ids2, q).numpy()
    np.testing.assert_allclose(a[:, :-1], b[:, :-1], atol=1e-5)
    # left padding changes a read's logits: the reference's output depends on batch composition
    ids3 = np.concatenate([np.full((2, 5), 4), ids], axis=1)
    q3 = np.concatenate([np.zeros((2, 5), np.float32), q], axis=1)
    c = fr.forward(sd, ids3, q3).numpy()[:, 5:]
    assert np.abs(c - a).max() > 1e-3
