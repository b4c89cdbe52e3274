"""Which model wrote the reference's committed prediction fixtures -- and therefore what they can pin.

tests/golden/eval_chunk*.npz are lossless copies of /root/reference/tests/data/eval/chunk{0,1}/{0,1,2}.pt
(tests/golden/make_pt_fixtures.py).  SURVEY 8(c) took them for HyenaDNA output; they are not."""
import os

import numpy as np
import pytest

import real_ckpt

RF = 12  # (7 - 1)/2 + (9 - 1)/2 + (11 - 1)/2: three stacked "same"-padded convolutions


@pytest.mark.parametrize("name", real_ckpt.fixture_names())
def test_reference_eval_fixtures_have_a_finite_receptive_field(golden_dir, name):
    """The stored logits are `BenchmarkCNN` outputs (deepchopper/models/cnn.py:24-37 with configs/model/cnn.yaml:
    `filter_sizes: [7, 9, 11]` -> receptive field exactly +-12 tokens; BatchNorm in eval mode is affine, quality 0 on pad),
    NOT HyenaDNA outputs.  Measured here on every left-padded batch:

    * inside a [PAD] run the logit pair is a transient for positions 0..11 and EXACTLY constant from position 12 up to 12
      positions before the run ends (a bounded look-back of 12);
    * two rows agree bit for bit over their shared [PAD] prefix until exactly 12 positions before the shorter run ends,
      and differ from there (a look-AHEAD of 12).

    HyenaDNA (deepchopper/models/llm/hyena.py:22,29-41; dc_hg.py:120-163) is causal with an unbounded look-back
    (`short_filter(...)[..., :L]`, 2L-point `fftconv`: oracle/forward_ref.py:68-77,106-113): its pad-run logits only
    converge asymptotically and never see the read ahead.  So these fixtures pin T3-T5 (left pad, SEP, -100 mask, quality
    normalisation, id row) and S0-S5 on real predictions -- those tests are model-agnostic -- and NOTHING about F2.  No
    HyenaDNA logits exist anywhere in /root/reference; F2 can only be pinned by the reference's own output with the real
    model (tests/test_gpu_real_checkpoint.py, $DEEPCHOPPER_REF_PT_DIR)."""
    d = np.load(os.path.join(golden_dir, name))
    p, seq = d["prediction"], d["seq"]
    B = p.shape[0]
    npad = real_ckpt.pad_counts(seq)
    for b in range(B):  # the layout the property is stated on: pads are a prefix
        assert (seq[b, : npad[b]] == real_ckpt.PAD_ID).all() and not (seq[b, npad[b]:] == real_ckpt.PAD_ID).any()
    # bounded look-back: exactly constant inside a pad run, RF positions away from both of its ends
    checked_rows = 0
    for b in range(B):
        if npad[b] > 2 * RF + 1:
            inner = p[b, RF: npad[b] - RF]
            assert np.array_equal(inner, np.broadcast_to(inner[0], inner.shape)), f"row {b}: pad-run logits are not constant"
            assert not np.array_equal(p[b, 0], inner[0])  # the transient at the left edge ("same" zero padding)
            checked_rows += 1
    assert checked_rows >= 8
    # look-ahead of exactly RF: first differing position of a row pair over the shared pad prefix
    pairs = 0
    for a in range(B):
        for b in range(a + 1, B):
            m = int(min(npad[a], npad[b]))
            if m <= RF or npad[a] == npad[b]:
                continue
            differs = np.any(p[a, :m] != p[b, :m], axis=1)
            assert differs.any(), "a causal model would show no difference here; this one must"
            assert int(np.argmax(differs)) == m - RF, f"rows {a},{b}: first difference at {int(np.argmax(differs))}, expected {m - RF}"
            pairs += 1
    assert pairs >= 40
    # and the harness therefore refuses these files as a pin for F2
    worst, n = real_ckpt.max_lookahead(p, seq)
    assert n >= pairs and worst > 10.0
    with pytest.raises(AssertionError, match="looks ahead"):
        real_ckpt.require_causal({"prediction": p, "seq": seq}, name)


def test_oracle_forward_is_causal_where_the_fixtures_are_not():
    """The same two measurements on the oracle's HyenaDNA restatement (small random batch, CPU): logits over the shared pad
    prefix agree between rows to fp32 FFT rounding, all the way to the end of the shorter run -- no look-ahead."""
    from deepchopper_amd.weights import synthetic_state_dict
    from oracle import forward_ref as fr

    rng = np.random.default_rng(5)
    B, L = 4, 160
    ids = rng.integers(7, 12, (B, L)).astype(np.int64)
    q = rng.random((B, L)).astype(np.float32)
    ids[:, -1] = 1
    for b, n in enumerate((90, 40, 64, 0)):
        ids[b, :n] = real_ckpt.PAD_ID
        q[b, :n] = 0
    got = fr.forward(synthetic_state_dict(0), ids, q).numpy()
    worst, pairs = real_ckpt.max_lookahead(got, ids)
    assert pairs == 3 and worst < 1e-4
    assert real_ckpt.require_causal({"prediction": got, "seq": ids}, "oracle") == 3


def test_fft_long_convolution_leaks_rounding_noise_backwards_in_the_reference_too():
    """Why tests/test_gpu_forward.py::test_forward_is_causal_... holds the FFT forms to a tolerance and only the block-Toeplitz
    form to bit-identity: the reference's own `fftconv` (2L-point rfft/irfft; oracle/forward_ref.py:68-77) is a circular
    convolution of the whole row in fp32, so what follows position t reaches the outputs before t as rounding noise --
    O(1e-5) on O(1) logits at 3 000 tokens.  Measured here on the oracle; far below the 1e-2 the causality gate of the
    replay harness allows, and nothing like a receptive field."""
    from deepchopper_amd.weights import synthetic_state_dict
    from oracle import forward_ref as fr

    rng = np.random.default_rng(3000)
    B, L = 2, 3000
    ids = rng.integers(7, 12, (B, L)).astype(np.int64)
    q = rng.integers(1, 41, (B, L)).astype(np.float32)
    ids[:, -1], q[:, -1] = 1, 0
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    for b, n in enumerate((L // 2, L // 5)):
        ids[b, :n] = real_ckpt.PAD_ID
        q[b, :n] = 0
    got = fr.forward(synthetic_state_dict(0), ids, q).numpy()
    worst, pairs = real_ckpt.max_lookahead(got, ids)
    assert pairs == 1 and 1e-7 < worst < 2e-4, worst
