"""GPU parity tests for the forward (F1-F3) through the C ABI against the torch-fp32 CPU oracle.

Tolerance: BASELINE.json's north_star asks for per-base logits within 1e-3; the fp32-MFMA path
is an exact fp32 FMA chain, so these tests hold it to 2e-4 absolute on O(1) logits and require the
argmax labels to be identical wherever the oracle's margin exceeds that tolerance."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 2e-4


@pytest.fixture(scope="module", params=["f32", "bf16x6", "fp16x3"])
def env(hip_lib, request):
    import torch

    from deepchopper_amd import _lib, ops
    from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict

    assert _lib.device_count() >= 1, "GPU tests need a HIP device; there is no CPU fallback"
    sd = synthetic_state_dict(0)
    model = ops.HipModel(sd, HyenaDNAConfig())
    model.set_gemm_mode(request.param)
    return torch, ops, sd, model


def _fft_points(n_tokens: int) -> int:
    """Transform size (complex points) of the FFT size class a sequence of n_tokens takes (csrc/forward.h: kFftClass)."""
    for m in (1024, 2048, 4096, 8192, 12288, 16384):
        if n_tokens <= m:
            return m
    return 32768


def _inputs(rng, B, L, pad_max=0):
    ids = rng.integers(7, 12, (B, L)).astype(np.int64)
    q = rng.integers(1, 41, (B, L)).astype(np.float32)
    ids[:, -1] = 1
    q[:, -1] = 0
    for b in range(B):
        p = int(rng.integers(0, pad_max + 1)) if pad_max else 0
        ids[b, :p] = 4
        q[b, :p] = 0
    q /= np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    return ids, q.astype(np.float32)


def _check(logits, ref):
    diff = np.abs(logits - ref)
    assert np.isfinite(logits).all()
    assert diff.max() < ATOL, f"max |dlogit| = {diff.max():.3e}"
    margin = np.abs(ref[..., 1] - ref[..., 0])
    sure = margin > 2 * ATOL
    assert np.array_equal((logits[..., 1] > logits[..., 0])[sure], (ref[..., 1] > ref[..., 0])[sure])


def test_filter_matches_oracle(env):
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    cfg = fr.HyenaDNAConfig()
    tsd = {k: torch.as_tensor(v) for k, v in sd.items()}
    for layer in (0, 3):
        L = 1500
        k = model.filter(layer, L).cpu().numpy()
        p = fr.BB + f"layers.{layer}.mixer.filter_fn."
        ref = fr.hyena_filter(tsd, p, L, cfg).numpy().T.copy()  # [D, L]
        ref[:, 0] += sd[p + "bias"]  # the skip term is folded into tap 0
        np.testing.assert_allclose(k, ref, rtol=0, atol=3e-5)


@pytest.mark.parametrize("B,L", [(1, 1), (1, 31), (2, 33), (3, 129), (33, 65), (5, 513), (2, 1001)])
def test_forward_dense_matches_oracle(env, B, L):
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(100 + B * 7 + L)
    ids, q = _inputs(rng, B, L)
    ref = fr.forward(sd, ids, q).numpy()
    out = model.forward(ops.to_device(ids), ops.to_device(q)).cpu().numpy()
    _check(out, ref)
    # compact uint8 ids give the same bits
    out8 = model.forward(ops.to_device(ids.astype(np.uint8)), ops.to_device(q)).cpu().numpy()
    assert np.array_equal(out8, out)


def test_forward_left_padded_batch_matches_oracle(env):
    """Reference-style batch: left pad tokens (id 4) are ordinary tokens to the model."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(42)
    ids, q = _inputs(rng, 6, 300, pad_max=120)
    ref = fr.forward(sd, ids, q).numpy()
    out = model.forward(ops.to_device(ids), ops.to_device(q)).cpu().numpy()
    _check(out, ref)


def test_forward_ragged_equals_per_read_oracle(env):
    """Native zero-padding mode == the reference at --batch-size 1 for every read."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(7)
    lens = [1, 17, 32, 33, 100, 257, 640, 77, 5, 999, 64, 300] + [int(x) for x in rng.integers(1, 400, 40)]
    B, stride = len(lens), max(lens)
    ids = np.full((B, stride), 4, np.int64)
    q = np.zeros((B, stride), np.float32)
    first = np.zeros(B, np.int32)
    for b, n in enumerate(lens):
        i1, q1 = _inputs(rng, 1, n)
        first[b] = stride - n
        ids[b, stride - n:] = i1[0]
        q[b, stride - n:] = q1[0]
    out = model.forward_ragged(ops.to_device(ids), ops.to_device(q), first, np.array(lens, np.int32)).cpu().numpy()
    for b, n in enumerate(lens):
        ref = fr.forward(sd, ids[b: b + 1, stride - n:], q[b: b + 1, stride - n:]).numpy()[0]
        _check(out[b, stride - n:], ref)
        assert np.all(out[b, : stride - n] == 0)


def test_head_against_reference_golden(env, golden_dir):
    """F3 pinned on the GPU by vectors generated from the reference's own head.py
    (deepchopper/models/llm/head.py:75-102, tests/golden/make_head_golden.py): the golden residual stream goes
    through the HIP final-LayerNorm + head kernels (`dc_head_forward`, the same launches dc_forward ends with)
    in the GEMM mode of the fixture and must reproduce the reference's logits."""
    torch, ops, sd, model = env
    from deepchopper_amd.weights import BB, HD, HyenaDNAConfig, synthetic_state_dict

    g = np.load(os.path.join(golden_dir, "head_golden.npz"))
    sd7 = dict(synthetic_state_dict(seed=7))
    chk = np.array([float(np.sum(sd7[HD + f"{n}.weight"].astype(np.float64))) for n in ("linear1", "linear2", "linear3")])
    np.testing.assert_allclose(chk, g["weight_checksums"], rtol=1e-12)
    sd7[BB + "ln_f.weight"], sd7[BB + "ln_f.bias"] = g["ln_f_weight"], g["ln_f_bias"]
    m7 = ops.HipModel(sd7, HyenaDNAConfig())
    m7.set_gemm_mode(model.gemm_mode)
    got = m7.head_forward(ops.to_device(g["pre"]), ops.to_device(g["quals"])).cpu().numpy()
    assert got.shape == g["logits"].shape and np.isfinite(got).all()
    err = np.abs(got - g["logits"]).max()
    assert err <= ATOL, f"HIP head vs reference head.py: max |dlogit| = {err:.3e} ({model.gemm_mode})"
    m7.close()


def test_forward_deterministic_and_batch_invariant(env):
    torch, ops, sd, model = env
    rng = np.random.default_rng(5)
    ids, q = _inputs(rng, 40, 257)
    a = model.forward(ops.to_device(ids), ops.to_device(q)).cpu().numpy()
    b = model.forward(ops.to_device(ids), ops.to_device(q)).cpu().numpy()
    assert np.array_equal(a, b)
    # a read's logits do not depend on its batch neighbours
    c = model.forward(ops.to_device(ids[3:4]), ops.to_device(q[3:4])).cpu().numpy()
    np.testing.assert_allclose(c[0], a[3], rtol=0, atol=1e-5)


CAUSAL_CASES = [  # (tokens per row, long-convolution path, stage that must have run, launches)
    (600, "block-Toeplitz MFMA, one chunk", None, 0),
    (1500, "block-Toeplitz MFMA, (output, input) chunk windows", None, 0),
    (3000, "in-LDS radix-16 FFT", "hyena_fft", 4),
    (20000, "four-step 2^15-point FFT", "hyena_fft15", 4),
]


@pytest.mark.parametrize("L,path,stage,launches", CAUSAL_CASES, ids=[str(c[0]) for c in CAUSAL_CASES])
def test_forward_is_causal_and_pad_prefix_is_row_independent(env, L, path, stage, launches):
    """The two properties HyenaDNA has and the model behind the reference's committed `.pt` fixtures has not
    (tests/test_reference_fixture_model.py: BenchmarkCNN, receptive field +-12), on the HIP path, in every GEMM mode and on
    every long-convolution path:

    (i)  in a left-padded batch (deepchopper/models/llm/tokenizer.py:71-87) the logits over the [PAD] prefix two rows share
         do not depend on the row -- all the way to the end of the shorter pad run: no look-ahead;
    (ii) logits at positions <= t do not change when tokens / qualities at positions > t change (`short_filter(...)[..., :L]`
         and the 2L-point `fftconv` of the reference's backbone, restated in oracle/forward_ref.py:68-77,106-113).

    The block-Toeplitz MFMA kernel is exactly causal (its blocks above the diagonal are exactly zero), so both hold BIT FOR
    BIT there.  The FFT forms compute what the reference's `fftconv` computes -- a circular convolution of the whole row
    -- so later tokens reach earlier outputs as fp32 rounding noise.  The reference has the same property: the torch-fp32
    oracle's own pad-prefix difference at 3 000 tokens is 5.6e-5 (tests/test_reference_fixture_model.py measures it on
    the CPU); the HIP FFT forms are held to 2e-4 here -- the parity tolerance of this file, 5 times inside north_star's
    1e-3 and five orders of magnitude below a real look-ahead (the CNN fixtures differ by ~80)."""
    torch, ops, sd, model = env
    rng = np.random.default_rng(L)
    B = 4
    ids, q = _inputs(rng, B, L)
    npad = [L // 2, L // 5, (2 * L) // 3, 0]
    for b, n in enumerate(npad):
        ids[b, :n] = 4
        q[b, :n] = 0
    model.set_profiling(True)
    model.profile_reset()
    try:
        a = model.forward(ops.to_device(ids.astype(np.uint8)), ops.to_device(q)).cpu().numpy()
        stages = model.profile()
    finally:
        model.set_profiling(False)
    assert np.isfinite(a).all()
    if stage is not None:
        assert stages.get(stage, (0, 0))[1] == launches, (path, stages)
    else:
        assert stages.get("hyena_fft", (0, 0))[1] == 0 and stages.get("hyena_fft15", (0, 0))[1] == 0, (path, stages)
    exact = stage is None
    tol = 0.0 if exact else ATOL

    def same(x, y, what):
        if exact:
            assert np.array_equal(x, y), f"{path}: {what}: max |d| = {np.abs(x - y).max():.3e} (must be bit-identical)"
        else:
            assert np.abs(x - y).max() <= tol, f"{path}: {what}: max |d| = {np.abs(x - y).max():.3e} > {tol}"

    # (i) shared pad prefix, every pair of padded rows
    for r in range(3):
        for s in range(r + 1, 3):
            m = min(npad[r], npad[s])
            same(a[r, :m], a[s, :m], f"pad prefix of rows {r},{s} (first {m} positions)")
    # (ii) change everything behind t: the bases, the qualities, and (row 1) the amount of padding that follows
    t = L // 3
    ids2, q2 = ids.copy(), q.copy()
    ids2[:, t + 1: L - 1] = rng.integers(7, 12, (B, L - t - 2))
    q2[:, t + 1:] = rng.random((B, L - t - 1)).astype(np.float32)
    b2 = model.forward(ops.to_device(ids2.astype(np.uint8)), ops.to_device(q2)).cpu().numpy()
    same(a[:, : t + 1], b2[:, : t + 1], f"logits at positions <= {t} after changing positions > {t}")
    assert np.abs(a[:, t + 1:] - b2[:, t + 1:]).max() > 1e-3  # ... and the change did reach the positions behind t


def test_forward_at_bench_shape_properties(env):
    """1 kb reads at a bench-like batch: finite, deterministic, spot-checked against the oracle."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(9)
    ids, q = _inputs(rng, 256, 1001)
    out = model.forward(ops.to_device(ids.astype(np.uint8)), ops.to_device(q))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert np.isfinite(out).all()
    for b in (0, 100, 255):
        ref = fr.forward(sd, ids[b: b + 1], q[b: b + 1]).numpy()[0]
        _check(out[b], ref)


def test_gemm_modes_error_vs_float64(hip_lib):
    """The bf16x6 GEMM path (error-free 3-way bf16 split) is fp32-equivalent: against a float64
    evaluation of the oracle its error is of the same size as the fp32-MFMA path's and the torch-fp32
    oracle's own."""
    import torch

    from deepchopper_amd import ops
    from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict
    from oracle import forward_ref as fr

    sd = synthetic_state_dict(3)
    model = ops.HipModel(sd, HyenaDNAConfig())
    rng = np.random.default_rng(77)
    ids, q = _inputs(rng, 4, 700)
    ref64 = fr.forward(sd, ids, q, dtype=torch.float64).numpy()
    ref32 = fr.forward(sd, ids, q).numpy()
    errs = {"oracle_f32": np.abs(ref32 - ref64).max()}
    for mode in ("f32", "bf16x6", "fp16x3"):
        model.set_gemm_mode(mode)
        out = model.forward(ops.to_device(ids), ops.to_device(q)).cpu().numpy()
        errs[mode] = np.abs(out - ref64).max()
    print("max |logit - float64 oracle|:", {k: float(v) for k, v in errs.items()})
    assert errs["f32"] < 1e-4 and errs["bf16x6"] < 1e-4 and errs["fp16x3"] < 2e-4
    assert errs["bf16x6"] < 4 * max(errs["f32"], errs["oracle_f32"]) + 1e-6


def test_long_convolution_forms_error_vs_float64(hip_lib):
    """Which long-convolution form is nearest the truth on long reads?  Reads of 5 000 and 11 000 tokens (the 2 x 4096 and
    3 x 4096 split transforms) through the block-Toeplitz MFMA kernel, the power-of-two FFT of rounds 2-4 and the split FFT,
    each against a float64 evaluation of the oracle: all three within 2e-4 of it, the split form no further from it than the
    power-of-two form (fewer passes, fewer roundings) beyond rounding noise, and the torch-fp32 oracle itself of the same size."""
    import torch

    from deepchopper_amd import ops
    from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict
    from oracle import forward_ref as fr

    sd = synthetic_state_dict(3)
    model = ops.HipModel(sd, HyenaDNAConfig())
    try:
        for L in (5000, 11000):
            rng = np.random.default_rng(L)
            ids, q = _inputs(rng, 1, L)
            ref64 = fr.forward(sd, ids, q, dtype=torch.float64).numpy()
            errs = {"oracle_f32": float(np.abs(fr.forward(sd, ids, q).numpy() - ref64).max())}
            d_ids, d_q = ops.to_device(ids.astype(np.uint8)), ops.to_device(q)
            for name, thr, split in (("toeplitz", 1 << 20, True), ("fft_pow2", 2560, False), ("fft_split", 2560, True)):
                model.set_fft_threshold(thr)
                model.set_fft_split(split)
                errs[name] = float(np.abs(model.forward(d_ids, d_q).cpu().numpy() - ref64).max())
            print(f"L = {L}: max |logit - float64 oracle|:", {k: f"{v:.2e}" for k, v in errs.items()})
            assert max(errs.values()) < 2e-4, errs
            assert errs["fft_split"] < 1.5 * errs["fft_pow2"] + 2e-5, errs
    finally:
        model.close()


@pytest.mark.parametrize("mode", ["fp16x3", "bf16x6"])
@pytest.mark.parametrize("d_inner,gelu", [(1024, "tanh"), (512, "tanh"), (1024, "none"), (384, "tanh")])
def test_mlp_fusion_matches_two_kernel_form(hip_lib, d_inner, gelu, mode):
    """The fused MLP kernel (fc1 -> GELU -> fc2 with the hidden activation in registers, fc2's k order permuted on the
    host) against the two-GEMM form of the same build and against the oracle, for the reference shape, a narrower MLP
    (4 hidden chunks), the erf GELU, and a width the fused kernel does not take (d_inner % 128 != 0 -> the switch is a
    no-op and both runs use the two-kernel form) -- in the default 2 x fp16 split AND, since round 6, in the strict
    error-free 3 x bf16 split (mlp_fused6_kernel: the mode the range guard falls back to)."""
    from deepchopper_amd import ops
    from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict
    from oracle import forward_ref as fr

    cfg = HyenaDNAConfig(d_inner=d_inner, gelu_approximate=gelu)
    sd = synthetic_state_dict(11, cfg)
    model = ops.HipModel(sd, cfg)
    assert model.gemm_mode == "fp16x3"
    model.set_gemm_mode(mode)
    rng = np.random.default_rng(d_inner)
    ids, q = _inputs(rng, 5, 333)
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    model.set_profiling(True)
    ran = lambda st, name: st.get(name, (0.0, 0))[1] > 0
    takes = d_inner % 128 == 0
    chained = model.forward(d_ids, d_q).cpu().numpy()  # default: MLP + the following GEMM in one kernel
    st2 = model.profile()
    assert ran(st2, "mlp_inproj_fused") == takes and ran(st2, "mlp_head1_fused") == takes
    assert ran(st2, "gemm_head1") == (not takes) and st2.get("gemm_inproj", (0, 0))[1] == (0 if takes else 3)
    model.profile_reset()
    model.set_mlp_fusion(1)
    fused = model.forward(d_ids, d_q).cpu().numpy()
    stages = model.profile()
    model.profile_reset()
    model.set_mlp_fusion(0)
    plain = model.forward(d_ids, d_q).cpu().numpy()
    stages_plain = model.profile()
    assert ran(stages, "mlp_fused") == takes and ran(stages, "gemm_fc1") == (not takes) and ran(stages, "gemm_head1")
    assert not ran(stages_plain, "mlp_fused") and ran(stages_plain, "gemm_fc1") and ran(stages_plain, "gemm_fc2")
    ref = fr.forward(sd, ids, q, fr.HyenaDNAConfig(d_inner=d_inner, gelu_approximate=gelu)).numpy()
    _check(chained, ref)
    _check(fused, ref)
    _check(plain, ref)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(fused - plain).max() < 5e-5 * scale  # same split arithmetic, different summation order only
    assert np.abs(chained - plain).max() < 5e-5 * scale
    assert model.overflow() == 0
    # identical labels wherever the margin exceeds the summation-order noise
    sure = np.abs(plain[..., 1] - plain[..., 0]) > 1e-4 * scale
    assert np.array_equal((chained[..., 1] > chained[..., 0])[sure], (plain[..., 1] > plain[..., 0])[sure])


@pytest.mark.parametrize("B,L", [(2, 2001), (1, 5000), (1, 10001), (2, 20000), (1, 32768), (1, 32769)])
def test_forward_long_reads_match_oracle(env, B, L):
    """BASELINE configs 3 and 4: 2 kb reads and reads up to 10 kb (several 32-block output chunks of the
    long-convolution kernel, filters longer than one LDS chunk); 32 769 tokens = the longest sequence the
    tokenizer emits (max_len_single_sentence + [SEP]), i.e. the full 32k filter.  Reads above 16 384 tokens take the
    four-step 2^15-point FFT (16 x 2048, csrc/hyena_fft.hip): asserted through the profile counters; 32 769 = 2^15 + 1
    exercises the single tap that form adds by hand, 32 768 the case that does not need it."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(L)
    ids, q = _inputs(rng, B, L)
    ref = fr.forward(sd, ids, q).numpy()
    model.set_profiling(True)
    model.profile_reset()
    try:
        out = model.forward(ops.to_device(ids.astype(np.uint8)), ops.to_device(q)).cpu().numpy()
        stages = model.profile()
    finally:
        model.set_profiling(False)
    _check(out, ref)
    if model.gemm_mode != "f32" or True:  # the FFT forms serve every GEMM mode
        want15 = 4 if L > 16384 else 0  # one launch pair per layer
        assert stages.get("hyena_fft15", (0, 0))[1] == want15, stages
        if 2560 < L <= 16384:
            assert stages.get("hyena_fft", (0, 0))[1] == 4, stages


def test_fft15_matches_toeplitz_and_mixes_with_other_sizes(hip_lib):
    """The four-step form against the block-Toeplitz MFMA kernel of the same build (which serves these lengths when the
    FFT is switched off), and a ragged batch that holds reads of the in-LDS sizes, the four-step size and the MFMA path
    at once: every read equals its own single-read result."""
    from deepchopper_amd import ops
    from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict

    sd = synthetic_state_dict(41)
    model = ops.HipModel(sd, HyenaDNAConfig())
    rng = np.random.default_rng(41)
    L = 24001
    ids, q = _inputs(rng, 1, L)
    d_ids, d_q = ops.to_device(ids.astype(np.uint8)), ops.to_device(q)
    try:
        model.set_fft_threshold(1 << 20)
        toep = model.forward(d_ids, d_q).cpu().numpy()
        model.set_fft_threshold(2560)
        fft = model.forward(d_ids, d_q).cpu().numpy()
        assert np.abs(fft - toep).max() < 1e-4
        lens = np.array([L, 700, 9000, 17000, 3000], np.int32)
        ids2 = np.full((len(lens), L), 4, np.uint8)
        q2 = np.zeros((len(lens), L), np.float32)
        singles = []
        for b, n in enumerate(lens):
            i1, q1 = (ids, q) if b == 0 else _inputs(rng, 1, int(n))
            ids2[b, L - n:], q2[b, L - n:] = i1[0], q1[0]
            singles.append(model.forward(ops.to_device(i1.astype(np.uint8)), ops.to_device(q1)).cpu().numpy()[0])
        first = (L - lens).astype(np.int32)
        out = model.forward_ragged(ops.to_device(ids2), ops.to_device(q2), first, lens).cpu().numpy()
        for b, n in enumerate(lens):
            assert np.abs(out[b, L - n:] - singles[b]).max() < 2e-5, (b, n)
    finally:
        model.set_fft_threshold(2560)


def test_forward_ragged_mixed_lengths_config4(env):
    """Config 4 shape: log-uniform lengths 100 bp - 10 kb in one ragged batch (groups of 32 reads sorted by
    length inside the long-convolution kernel); spot-check reads of every size class against the oracle."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(4)
    lens = np.exp(rng.uniform(np.log(100), np.log(10000), 70)).astype(int) + 1
    lens[:3] = [101, 10001, 3333]
    B, stride = len(lens), int(lens.max())
    ids = np.full((B, stride), 4, np.uint8)
    q = np.zeros((B, stride), np.float32)
    first = np.zeros(B, np.int32)
    for b, n in enumerate(lens):
        i1, q1 = _inputs(rng, 1, int(n))
        first[b] = stride - n
        ids[b, stride - n:] = i1[0]
        q[b, stride - n:] = q1[0]
    out = model.forward_ragged(ops.to_device(ids), ops.to_device(q), first, lens.astype(np.int32)).cpu().numpy()
    assert np.isfinite(out).all()
    order = np.argsort(lens)
    for b in [0, 1, 2, int(order[0]), int(order[len(order) // 2]), int(order[-1])]:
        n = int(lens[b])
        ref = fr.forward(sd, ids[b: b + 1, stride - n:].astype(np.int64), q[b: b + 1, stride - n:]).numpy()[0]
        _check(out[b, stride - n:], ref)


def test_forward_ragged_fuzz_one_handle_many_geometries(env):
    """Seeded sweep: ONE handle sees a series of unrelated ragged batches (1 .. 80 reads; lengths around the 4-slot /
    32-token / 1 024-token / FFT-threshold edges; workspace growing and shrinking; the cached geometry of an earlier
    batch coming back), every batch checked read by read against the per-read oracle."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(2024)
    edges = [1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 1026]
    batches = []
    for it in range(10):
        n = int(rng.integers(1, 81))
        lens = np.where(rng.random(n) < 0.35, rng.choice(edges, n), rng.integers(1, 420, n)).astype(np.int64)
        if it in (3, 7):
            lens[int(rng.integers(0, n))] = 2561 + int(rng.integers(0, 600))  # one read on the FFT path
        batches.append(lens)
    batches.append(batches[2].copy())  # same geometry key as an earlier batch, different contents
    for it, lens in enumerate(batches):
        B, stride = len(lens), int(lens.max())
        ids = np.full((B, stride), 4, np.uint8)
        q = np.zeros((B, stride), np.float32)
        first = (stride - lens).astype(np.int32)
        for b, n in enumerate(lens):
            i1, q1 = _inputs(rng, 1, int(n))
            ids[b, stride - n:] = i1[0]
            q[b, stride - n:] = q1[0]
        out = model.forward_ragged(ops.to_device(ids), ops.to_device(q), first, lens.astype(np.int32)).cpu().numpy()
        assert np.isfinite(out).all()
        # every short read, and the longest one
        check = [b for b in range(B) if lens[b] <= 300][:40] + [int(np.argmax(lens))]
        for b in check:
            n = int(lens[b])
            ref = fr.forward(sd, ids[b: b + 1, stride - n:].astype(np.int64), q[b: b + 1, stride - n:]).numpy()[0]
            _check(out[b, stride - n:], ref)
            assert np.all(out[b, : stride - n] == 0), f"batch {it}, read {b}: pad positions must stay zero"


def test_fp16x3_range_guard_and_fallback(hip_lib):
    """An activation beyond the fp16 range must not silently corrupt the fp16x3 path: the kernels raise
    the overflow counter and the pipeline redoes the batch with the bf16 split."""
    import torch

    from deepchopper_amd import ops, synth
    from deepchopper_amd.model import DeepChopper
    from deepchopper_amd.predict import PredictPipeline
    from deepchopper_amd.weights import BB, synthetic_state_dict
    from oracle import forward_ref as fr

    sd = synthetic_state_dict(5)
    sd[BB + "layers.1.mlp.fc1.bias"] = sd[BB + "layers.1.mlp.fc1.bias"] + np.float32(2.0e5)  # gelu(2e5) = 2e5 > 65504
    model = DeepChopper.from_state_dict(sd)
    assert model.net.gemm_mode == "fp16x3"
    rng = np.random.default_rng(1)
    ids, q = _inputs(rng, 3, 200)
    model.net.forward(ops.to_device(ids), ops.to_device(q))
    assert model.net.overflow() > 0 and model.net.overflow() == 0  # raised once, cleared by the read
    seq, qual, rid = synth.synth_reads_fixed(6, 200, seed=42)
    pipe = PredictPipeline(model)
    out, res = pipe.predict_reads(rid, [s.tobytes() for s in seq], [x.tobytes() for x in qual], reference_padding=True)
    ref = fr.forward(sd, res["tok"]["input_ids"].cpu().numpy().astype(np.int64), res["tok"]["input_quals"].cpu().numpy()).numpy()
    got = res["logits"].cpu().numpy()
    assert np.isfinite(got).all() and model.net.gemm_mode == "fp16x3"
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(got - ref).max() < 2e-4 * scale


def test_layer0_table_gather_is_bit_identical(hip_lib, monkeypatch):
    """Layer 0's in_proj output is a function of the token id alone; by default the long-convolution kernel gathers
    it from the 16-row table instead of reading a materialised [3D][T] buffer.  Both forms see the same fp32
    values, so the logits must be identical to the last bit (fixed-length and ragged batches, pad tokens, [UNK])."""
    from deepchopper_amd import ops
    from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict

    sd = synthetic_state_dict(21)
    model = ops.HipModel(sd, HyenaDNAConfig())
    rng = np.random.default_rng(21)
    ids, q = _inputs(rng, 6, 777)
    ids[0, :40] = 4   # left-pad tokens run through the model in reference padding
    ids[1, 100] = 6   # [UNK]
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    first = np.array([0, 5, 100, 333, 0, 776 - 31], np.int32)
    ntok = (777 - first).astype(np.int32)
    outs = {}
    for gather in (True, False):
        if gather:
            monkeypatch.delenv("DC_NO_L0_GATHER", raising=False)
        else:
            monkeypatch.setenv("DC_NO_L0_GATHER", "1")
        outs[gather] = (model.forward(d_ids, d_q).cpu().numpy(), model.forward_ragged(d_ids, d_q, first, ntok).cpu().numpy())
    assert np.array_equal(outs[True][0], outs[False][0])
    assert np.array_equal(outs[True][1], outs[False][1])


@pytest.mark.parametrize("mode", ["bf16x6", "fp16x3"])
@pytest.mark.parametrize("gpw", [None, "1", "3"])
def test_warp_specialised_long_convolution_is_bit_identical_to_hyena3(hip_lib, monkeypatch, mode, gpw):
    """csrc/hyena_ws_kernel.inc (producer waves build the gated-input tiles and finish the output blocks, consumer waves
    only run MFMAs; a workgroup streams several 32-read groups of its channel) against hyena3_kernel<32> of the same build
    and, for reads above 1 024 tokens (two or three output chunks per group), hyena3_kernel<16>
    (DC_H3_WS=0): same arithmetic in the same order, so the logits must be BIT-identical -- on ragged batches whose groups
    differ in length (1 .. 80 blocks per group, groups with empty read slots, reads that end inside a block), on a dense
    left-padded batch, with the layer-0 token-table gather and without it, and for every groups-per-workgroup setting."""
    from deepchopper_amd import ops
    from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict

    if gpw is not None:
        monkeypatch.setenv("DC_H4_GPW", gpw)
    sd = synthetic_state_dict(77)
    rng = np.random.default_rng(77)
    lens = np.concatenate([rng.integers(1, 1025, 150), [1, 2, 3, 4, 5, 31, 32, 33, 64, 65, 992, 993, 1023, 1024] * 2,
                           np.full(70, 1001),
                           # groups of two and three output chunks (the 16-block form of hyena3 on the other side)
                           rng.integers(1025, 2561, 60), [1025, 1056, 1057, 2047, 2048, 2049, 2080, 2081, 2559, 2560] * 2,
                           np.full(40, 2001)]).astype(np.int32)
    B, stride = len(lens), int(lens.max())
    ids = np.full((B, stride), 4, np.uint8)
    q = np.zeros((B, stride), np.float32)
    first = (stride - lens).astype(np.int32)
    for b, n in enumerate(lens):
        i1, q1 = _inputs(rng, 1, int(n))
        ids[b, stride - n:], q[b, stride - n:] = i1[0], q1[0]
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    dense_ids, dense_q = _inputs(rng, 37, 700, pad_max=300)
    dd_ids, dd_q = ops.to_device(dense_ids.astype(np.uint8)), ops.to_device(dense_q)
    out = {}
    for ws in ("1", "0"):
        for l0 in ("0", "1"):
            monkeypatch.setenv("DC_H3_WS", ws)
            if l0 == "1":
                monkeypatch.setenv("DC_NO_L0_GATHER", "1")
            else:
                monkeypatch.delenv("DC_NO_L0_GATHER", raising=False)
            model = ops.HipModel(sd, HyenaDNAConfig())
            model.set_gemm_mode(mode)
            model.set_profiling(True)
            model.profile_reset()
            a = model.forward_ragged(d_ids, d_q, first, lens).cpu().numpy()
            b = model.forward(dd_ids, dd_q).cpu().numpy()
            assert model.profile().get("hyena", (0, 0))[1] == 8 and model.profile().get("hyena_fft", (0, 0))[1] == 0
            assert model.overflow() == 0 if mode == "fp16x3" else True
            out[(ws, l0)] = (a, b)
            model.close()
    for l0 in ("0", "1"):
        for i in (0, 1):
            x, y = out[("1", l0)][i], out[("0", l0)][i]
            assert np.isfinite(x).all()
            assert np.array_equal(x, y), f"ws vs hyena3 ({mode}, no-l0-gather={l0}, {'ragged' if i == 0 else 'dense'}): max |d| = {np.abs(x - y).max():.3e}"
    for b, n in enumerate(lens):
        assert np.all(out[("1", "0")][0][b, : stride - n] == 0)


AB_SWITCHES = [("DC_NO_HEAD_FUSE", "1"), ("DC_NO_FUSED_STATS", "1"), ("DC_NO_L0_TABLE", "1"), ("DC_NO_L0_GATHER", "1"),
               ("DC_NO_MLP_CHAIN", "1"), ("DC_NO_MLP_FUSE", "1"), ("DC_HYENA", "2"), ("DC_H3_ONE_LAUNCH", "1"),
               ("DC_H3_CB16", "1"), ("DC_H3_WS", "0")]


@pytest.mark.parametrize("name,value", AB_SWITCHES, ids=[n for n, _ in AB_SWITCHES])
def test_ab_switch_paths_match_oracle(hip_lib, monkeypatch, name, value):
    """Every environment A/B switch that ships in the library selects a real, correct path (the unfused forms the
    other GEMM modes use anyway): each one is held to the same tolerance as the default path, on a ragged batch with
    reads on both sides of the 1 024-token chunk boundary of the long-convolution kernel."""
    from deepchopper_amd import ops
    from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict
    from oracle import forward_ref as fr

    sd = synthetic_state_dict(31)
    model = ops.HipModel(sd, HyenaDNAConfig())
    assert model.gemm_mode == "fp16x3"
    rng = np.random.default_rng(31)
    lens = np.array([40, 1500, 333, 1001, 2049, 64] + [200] * 30, np.int32)
    B, stride = len(lens), int(lens.max())
    ids = np.full((B, stride), 4, np.uint8)
    q = np.zeros((B, stride), np.float32)
    first = (stride - lens).astype(np.int32)
    for b, n in enumerate(lens):
        i1, q1 = _inputs(rng, 1, int(n))
        ids[b, stride - n:], q[b, stride - n:] = i1[0], q1[0]
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    base = model.forward_ragged(d_ids, d_q, first, lens).cpu().numpy()
    monkeypatch.setenv(name, value)
    out = model.forward_ragged(d_ids, d_q, first, lens).cpu().numpy()
    monkeypatch.delenv(name)
    assert np.abs(out - base)[ids != 4].max() < 1e-4
    for b in (0, 1, 3, 4):
        n = int(lens[b])
        ref = fr.forward(sd, ids[b: b + 1, stride - n:].astype(np.int64), q[b: b + 1, stride - n:]).numpy()[0]
        _check(out[b, stride - n:], ref)


def _robustness_cases():
    from deepchopper_amd.weights import BB, HD

    def outlier(key):
        def f(sd, rng):
            w = sd[key].copy()
            flat = w.reshape(-1)
            flat[int(rng.integers(0, flat.size))] *= np.float32(1.0e3)
            sd[key] = w
        return f

    def gamma_decades(sd, rng):
        for k in list(sd):
            if k.endswith(("norm1.weight", "norm2.weight", "ln_f.weight")):
                sd[k] = np.exp(rng.uniform(np.log(1e-2), np.log(1e2), sd[k].shape)).astype(np.float32) * np.sign(sd[k])

    def tiny_out_proj_input(sd, rng):
        for i in range(4):  # u scaled by 0.05 -> the gated long-convolution output y ~ u^3 ~ 1e-4
            for n in ("in_proj.weight", "in_proj.bias"):
                sd[BB + f"layers.{i}.mixer.{n}"] = sd[BB + f"layers.{i}.mixer.{n}"] * np.float32(0.05)

    def big_embeddings(sd, rng):
        sd[BB + "embeddings.word_embeddings.weight"] = sd[BB + "embeddings.word_embeddings.weight"] * np.float32(50.0)

    cases = {f"outlier:{k.split('.', 1)[1] if k.startswith(BB) else k}": outlier(k) for k in (
        BB + "layers.1.mixer.in_proj.weight", BB + "layers.1.mixer.out_proj.weight", BB + "layers.2.mlp.fc1.weight",
        BB + "layers.2.mlp.fc2.weight", HD + "linear1.weight", HD + "linear2.weight")}
    cases.update({"layernorm_gamma_1e-2..1e2": gamma_decades, "out_proj_input_1e-4": tiny_out_proj_input,
                  "embeddings_x50": big_embeddings})
    return cases


@pytest.mark.parametrize("case", list(_robustness_cases()))
def test_precision_robustness_vs_float64(hip_lib, case):
    """the default arithmetic (fp16x3: 22-bit products, per-tensor power-of-two weight scale) on
    weights a trained checkpoint may have -- one outlier x1e3 in a GEMM tensor (pushes the other weights' low parts
    towards fp16 subnormals), LayerNorm gains over four decades (folded into the next projection), tiny out_proj
    inputs (the `l` term vanishes), large embeddings.  Against a float64 evaluation of the oracle the fp16x3 error
    must stay within 4x the fp32-MFMA mode's (or the torch-fp32 oracle's own), or the range guard must fire and the
    bf16x6 fallback must meet that bound."""
    import torch

    from deepchopper_amd import ops
    from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict
    from oracle import forward_ref as fr

    sd = dict(synthetic_state_dict(13))
    _robustness_cases()[case](sd, np.random.default_rng(99))
    rng = np.random.default_rng(5)
    ids, q = _inputs(rng, 3, 333)
    ref64 = fr.forward(sd, ids, q, dtype=torch.float64).numpy()
    ref32 = fr.forward(sd, ids, q).numpy()
    scale = max(1.0, float(np.abs(ref64).max()))
    model = ops.HipModel(sd, HyenaDNAConfig())
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    errs = {"oracle_f32": float(np.abs(ref32 - ref64).max())}
    model.set_gemm_mode("f32")
    errs["f32"] = float(np.abs(model.forward(d_ids, d_q).cpu().numpy() - ref64).max())
    model.set_gemm_mode("bf16x6")
    errs["bf16x6"] = float(np.abs(model.forward(d_ids, d_q).cpu().numpy() - ref64).max())
    model.set_gemm_mode("fp16x3")
    counter = torch.zeros(1, dtype=torch.int32, device=d_ids.device)
    model.set_overflow_target(counter)
    raw = model.forward(d_ids, d_q).cpu().numpy()
    model.set_overflow_target(None)
    guard = int(counter.item())
    out = model.forward_checked(d_ids, d_q).cpu().numpy()
    errs["fp16x3"] = float(np.abs(out - ref64).max())
    errs["fp16x3_unchecked"] = float(np.abs(raw - ref64).max()) if np.isfinite(raw).all() else float("inf")
    print(case, "scale", scale, "guard", guard, {k: f"{v:.3e}" for k, v in errs.items()})
    yard = max(errs["f32"], errs["oracle_f32"])
    assert np.isfinite(out).all()
    assert errs["bf16x6"] <= 4 * yard + 1e-6 * scale, errs
    assert errs["fp16x3"] <= 4 * yard + 1e-6 * scale, errs   # after the guard's fallback, if it fired
    if guard == 0:
        assert errs["fp16x3_unchecked"] == errs["fp16x3"]
    assert errs["fp16x3"] <= 1e-3 * scale                      # north_star's tolerance, with margin to spare
    model.close()


@pytest.mark.parametrize("L", [1500, 2048, 2049, 3000, 4096, 4097, 8192, 8193, 10001, 12288, 12289, 16384])
def test_fft_long_convolution_matches_toeplitz_and_oracle(env, L):
    """The O(L log L) FFT long convolution (reads above the length threshold, csrc/hyena_fft.hip) against the
    block-Toeplitz MFMA kernel of the same build and against the oracle's `fftconv`: every size class (1 024 .. 16 384
    complex points: one radix-16 transform up to 4 096, the split forms 2 x 4096, 3 x 4096, 4 x 4096 above), lengths at and
    next to the class boundaries, and a ragged batch in which short reads keep the MFMA kernel while long ones take the
    FFT."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(L)
    ids, q = _inputs(rng, 1, L)
    d_ids, d_q = ops.to_device(ids.astype(np.uint8)), ops.to_device(q)
    try:
        model.set_fft_threshold(1 << 20)
        toep = model.forward(d_ids, d_q).cpu().numpy()
        model.set_fft_threshold(1024)
        model.set_profiling(True)
        model.profile_reset()
        fft = model.forward(d_ids, d_q).cpu().numpy()
        stages = model.profile()
        model.set_profiling(False)
        assert stages.get("hyena_fft", (0, 0))[1] == 4, stages  # one launch per layer
        ref = fr.forward(sd, ids, q).numpy()
        _check(fft, ref)
        _check(toep, ref)
        assert np.abs(fft - toep).max() < 1e-4
        # ragged: a short read (MFMA path) next to this one (FFT path), each equal to its own single-read result
        Ls = 700
        ids2 = np.full((2, L), 4, np.uint8)
        q2 = np.zeros((2, L), np.float32)
        i_s, q_s = _inputs(rng, 1, Ls)
        ids2[0, L - Ls:], q2[0, L - Ls:] = i_s[0], q_s[0]
        ids2[1], q2[1] = ids[0], q[0]
        first, ntok = np.array([L - Ls, 0], np.int32), np.array([Ls, L], np.int32)
        out = model.forward_ragged(ops.to_device(ids2), ops.to_device(q2), first, ntok).cpu().numpy()
        assert np.abs(out[1] - fft[0]).max() < 2e-5
        _check(out[0, L - Ls:], fr.forward(sd, i_s, q_s).numpy()[0])
    finally:
        model.set_fft_threshold(2560)  # the shipped default (forward_api.hip: fft_min_tokens)
        model.set_profiling(False)


def test_split_fft_classes_match_the_power_of_two_transforms(env):
    """dc_model_set_fft_split: the split / mixed-radix size classes (default) against the power-of-two transforms of rounds
    2-4 on one ragged launch that holds a read of every class (and one on the MFMA path): same logits to fp32 rounding, the
    launch counts show which classes ran, and both are within the parity tolerance of the oracle."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(77)
    lens = [900, 2600, 4096, 5000, 8192, 9000, 12288, 14000, 16384]
    stride = max(lens)
    ids = np.full((len(lens), stride), 4, np.uint8)
    q = np.zeros((len(lens), stride), np.float32)
    for r, n in enumerate(lens):
        i_r, q_r = _inputs(rng, 1, n)
        ids[r, :n], q[r, :n] = i_r[0], q_r[0]
    first, ntok = np.zeros(len(lens), np.int32), np.array(lens, np.int32)
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    try:
        outs, launches = {}, {}
        for split in (False, True):
            model.set_fft_split(split)
            model.set_profiling(True)
            model.profile_reset()
            outs[split] = model.forward_ragged(d_ids, d_q, first, ntok).cpu().numpy()
            launches[split] = model.profile().get("hyena_fft", (0, 0))[1]
            model.set_profiling(False)
        # classes present: power of two 2^12, 2^13, 2^14 (3 per layer); split 4096, 8192, 12288, 16384 (4)
        assert launches[False] == 4 * 3 and launches[True] == 4 * 4, launches
        worst = 0.0
        for r, n in enumerate(lens):
            worst = max(worst, float(np.abs(outs[True][r, :n] - outs[False][r, :n]).max()))
        assert worst < 1e-4, worst
        for r in (3, 5, 7):  # one read per split form against the oracle
            n = lens[r]
            _check(outs[True][r, :n], fr.forward(sd, ids[r: r + 1, :n].astype(np.int64), q[r: r + 1, :n]).numpy()[0])
    finally:
        model.set_fft_split(True)
        model.set_profiling(False)


def test_fft_pair_maps_match_the_unpack_multiply_repack_form(env):
    """Round 6: between its two transforms the split / staged FFT kernel applies ONE precomputed 2 x 2 complex map per conjugate
    pair (hyena_fft_pairmap_kernel: unpack, filter product, repack and the 1 / M of the inverse folded into A, A', B; 8 packed
    FMAs per pair) instead of round 5's unpack -> multiply -> repack (DC_FFT_PAIRMAP=0 keeps that form).  Same mathematics:
    every size class the kernel serves (4 096 one row, 2 / 3 / 4 x 4 096), dense rows and a left-padded batch in both
    shared-prefix forms, agree with the round-5 form to fp32 rounding and stay inside the parity tolerance of the oracle."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(606)
    lens = [2600, 4095, 4096, 4097, 7000, 8192, 8193, 12288, 12289, 16383, 16384]
    stride = max(lens)
    ids = np.full((len(lens), stride), 4, np.uint8)
    q = np.zeros((len(lens), stride), np.float32)
    for r, n in enumerate(lens):
        i_r, q_r = _inputs(rng, 1, n)
        ids[r, :n], q[r, :n] = i_r[0], q_r[0]
    first, ntok = np.zeros(len(lens), np.int32), np.array(lens, np.int32)
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    pids, pq, pfirst, pntok, pnpad = _padded_launch(rng, [[9000, 3000, 5000, 700], [12000, 2700]])
    d_pids, d_pq = ops.to_device(pids), ops.to_device(pq)
    before = os.environ.get("DC_FFT_PAIRMAP")
    outs = {}
    try:
        for pm in ("0", "1"):
            os.environ["DC_FFT_PAIRMAP"] = pm
            o = {"ragged": model.forward_ragged(d_ids, d_q, first, ntok).cpu().numpy()}
            for form in ("exact", "linear"):
                model.set_prefix_mode(form)
                o[form] = model.forward_shared_prefix(d_pids, d_pq, pfirst, pntok, pnpad).cpu().numpy()
            outs[pm] = o
    finally:
        model.set_prefix_mode("exact")
        if before is None:
            os.environ.pop("DC_FFT_PAIRMAP", None)
        else:
            os.environ["DC_FFT_PAIRMAP"] = before
    worst = {}
    for k in ("ragged", "exact", "linear"):
        assert np.isfinite(outs["1"][k]).all()
        worst[k] = float(np.abs(outs["1"][k] - outs["0"][k]).max())
    print(f"pair maps vs unpack-multiply-repack ({model.gemm_mode}): max |dlogit| {worst}")
    assert max(worst.values()) < 1e-4, worst
    for r in (1, 4, 7, 10):  # one read per size class against the oracle
        n = lens[r]
        _check(outs["1"]["ragged"][r, :n], fr.forward(sd, ids[r: r + 1, :n].astype(np.int64), q[r: r + 1, :n]).numpy()[0])


# ----------------------------------------------------------------------------------------------------------------
# dc_forward_shared_prefix: the reference's left-padded batches with the [PAD] runs of all rows computed once per launch
# ----------------------------------------------------------------------------------------------------------------
def _padded_launch(rng, batches):
    """`batches`: list of lists of read lengths in TOKENS (incl. [SEP]); every inner list is one reference batch, left-padded
    to its longest read (deepchopper/models/llm/tokenizer.py:71-87); all batches share one dense [n, row_stride] launch as
    `fastq.fuse_chunks` + `PredictPipeline.run_device_batch(sub_sizes=...)` lay them out.  Returns ids (uint8), quals,
    first, n_tokens (= the padded length of the row's batch), n_pad."""
    lens = np.array([x for b in batches for x in b], np.int32)
    n, stride = len(lens), int(lens.max())
    ids = np.full((n, stride), 4, np.uint8)
    q = np.zeros((n, stride), np.float32)
    ntok = np.empty(n, np.int32)
    r = 0
    for b in batches:
        ntok[r: r + len(b)] = max(b)
        r += len(b)
    for r, L in enumerate(lens.tolist()):
        ids[r, stride - L:] = rng.integers(7, 12, L)
        ids[r, -1] = 1
        qq = rng.integers(1, 41, L).astype(np.float32)
        qq[-1] = 0
        q[r, stride - L:] = qq / max(float(np.linalg.norm(qq)), 1e-12)
    return ids, q, (stride - ntok).astype(np.int32), ntok, (ntok - lens).astype(np.int32)


def _nanopore_lens(rng, n, cap):
    return np.clip(np.exp(np.log(900.0) + 0.75 * rng.standard_normal(n)), 100, cap).astype(np.int32) + 1


SHARED_PREFIX_CASES = {
    # name: (batches builder, every row on a block-Toeplitz form -> bit-identical)
    "uniform_500_1500": (lambda rng: [list(rng.integers(501, 1502, 12)) for _ in range(6)], True),
    "nanopore_like_capped_2500": (lambda rng: [list(_nanopore_lens(rng, 12, 2500)) for _ in range(6)], True),
    # pad runs = 0 (the longest row), 1, 2, 3 (owned by the row in full), 4, 5, 6, 7 (mod 4 = 0..3), 31, 32, 33, 63, 95 (mod 32)
    "pad_run_residues": (lambda rng: [[700 - p for p in (0, 1, 2, 3, 4, 5, 6, 7, 31, 32, 33, 63, 95, 127, 640, 697)]], True),
    "single_row_batches_and_no_padding": (lambda rng: [[300], [411, 411], [257, 100, 31, 2]], True),
    "padded_length_above_2560": (lambda rng: [[3100, 900, 2800, 150, 2561 - 4, 3099], [400, 380, 210]], False),
    "nanopore_like": (lambda rng: [list(_nanopore_lens(rng, 12, 12000)) for _ in range(4)], False),
    "padded_length_above_16384": (lambda rng: [[17000, 500, 16385 - 8, 9000], [3000, 2999, 64]], False),
}


@pytest.mark.parametrize("case", list(SHARED_PREFIX_CASES))
def test_shared_prefix_equals_the_padded_launch(env, case):
    """rows `[PAD] x p ++ read` (quality 0 on the pads) through dc_forward_shared_prefix give the
    logits of dc_forward_ragged on the whole padded rows -- over the real tokens AND over the pad columns -- bit for
    bit while every row takes a block-Toeplitz long convolution, <= 2e-4 (the parity tolerance) where rows take an FFT
    form, in every GEMM mode; and the compact launch really is compact (token slots of the GEMMs)."""
    torch, ops, sd, model = env
    build, exact = SHARED_PREFIX_CASES[case]
    rng = np.random.default_rng(len(case))
    ids, q, first, ntok, npad = _padded_launch(rng, build(rng))
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    padded = model.forward_ragged(d_ids, d_q, first, ntok).cpu().numpy()
    model.set_prefix_mode("exact")
    model.set_profiling(True)
    model.profile_reset()
    try:
        shared = model.forward_shared_prefix(d_ids, d_q, first, ntok, npad).cpu().numpy()
        stages = model.profile()
    finally:
        model.set_profiling(False)
    assert model.prefix_violations() == 0
    assert np.isfinite(shared).all()
    mask = np.zeros(ids.shape, bool)
    for r in range(len(ntok)):
        mask[r, first[r]: first[r] + ntok[r]] = True
    assert np.array_equal(shared[~mask], padded[~mask])  # (both leave the positions outside the rows untouched: zeros)
    d = np.abs(shared - padded)[mask]
    if exact:
        assert stages.get("hyena_fft", (0, 0))[1] == 0 and stages.get("hyena_fft15", (0, 0))[1] == 0, stages
        assert d.max() == 0.0, f"{case} ({model.gemm_mode}): max |d| = {d.max():.3e}, must be bit-identical"
    else:
        assert stages.get("hyena_fft", (0, 0))[1] + stages.get("hyena_fft15", (0, 0))[1] > 0, stages
        assert d.max() <= ATOL, f"{case} ({model.gemm_mode}): max |d| = {d.max():.3e}"
    if npad.max() >= 4:
        assert stages.get("pad_fill", (0, 0))[1] == 1, stages
    # same entry point with int64 ids (the `.pt` wire format's dtype)
    shared64 = model.forward_shared_prefix(ops.to_device(ids.astype(np.int64)), d_q, first, ntok, npad).cpu().numpy()
    assert np.array_equal(shared64, shared)


def test_shared_prefix_matches_batched_oracle(env):
    """... and against the oracle run on the left-padded batches themselves (what the reference computes)."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(314)
    batches = [[400, 130, 397, 262], [95, 333, 120], [2700, 610, 1200]]
    ids, q, first, ntok, npad = _padded_launch(rng, batches)
    out = model.forward_shared_prefix(ops.to_device(ids), ops.to_device(q), first, ntok, npad).cpu().numpy()
    r = 0
    for b in batches:
        c0 = ids.shape[1] - max(b)
        ref = fr.forward(sd, ids[r: r + len(b), c0:].astype(np.int64), q[r: r + len(b), c0:]).numpy()
        _check(out[r: r + len(b), c0:], ref)
        r += len(b)


def test_shared_prefix_contract_violation_is_counted(env):
    """The pad positions are never read by the model: a caller that declares a pad run over tokens that are not
    ([PAD], quality 0) gets a count, not silence."""
    torch, ops, sd, model = env
    rng = np.random.default_rng(2)
    ids, q, first, ntok, npad = _padded_launch(rng, [[300, 120, 200]])
    assert model.prefix_violations() == 0
    model.forward_shared_prefix(ops.to_device(ids), ops.to_device(q), first, ntok, npad)
    assert model.prefix_violations() == 0
    bad = ids.copy()
    bad[1, first[1] + 17] = 9
    model.forward_shared_prefix(ops.to_device(bad), ops.to_device(q), first, ntok, npad)
    assert model.prefix_violations() > 0
    assert model.prefix_violations() == 0  # (reading resets the counter)
    badq = q.copy()
    badq[2, first[2] + 40] = 0.25
    model.forward_shared_prefix(ops.to_device(ids), ops.to_device(badq), first, ntok, npad)
    assert model.prefix_violations() > 0
    with pytest.raises(Exception, match="pad run must be shorter"):
        model.forward_shared_prefix(ops.to_device(ids), ops.to_device(q), first, ntok, ntok)


@pytest.mark.parametrize("case", list(SHARED_PREFIX_CASES))
def test_shared_prefix_linear_form_equals_the_padded_launch(env, case):
    """The linear form (dc_model_set_prefix_mode 2, VERDICT r4 next #1 "stretch" taken one step further): the long
    convolution is linear, so with ONE shared [PAD] row giving z_pad and F = k * z_pad a row is
    y = ((k * (z - z_pad)) + F) * gate with z - z_pad = 0 over its pad run -- the long convolution covers the read's own
    tokens, whatever the batch's longest read.  Same mathematics in another summation order: held to the parity
    tolerance against the padded launch over real tokens AND pad columns, in every GEMM mode, and the long-convolution
    class of a row follows ITS length (a short read in a long batch stays on the block-Toeplitz kernel)."""
    torch, ops, sd, model = env
    build, _ = SHARED_PREFIX_CASES[case]
    rng = np.random.default_rng(len(case))
    batches = build(rng)
    ids, q, first, ntok, npad = _padded_launch(rng, batches)
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    padded = model.forward_ragged(d_ids, d_q, first, ntok).cpu().numpy()
    model.set_prefix_mode("linear")
    model.set_profiling(True)
    model.profile_reset()
    try:
        lin = model.forward_shared_prefix(d_ids, d_q, first, ntok, npad).cpu().numpy()
        stages = model.profile()
        again = model.forward_shared_prefix(d_ids, d_q, first, ntok, npad).cpu().numpy()
    finally:
        model.set_profiling(False)
        model.set_prefix_mode("exact")
    assert model.prefix_violations() == 0
    assert np.isfinite(lin).all() and np.array_equal(lin, again)
    mask = np.zeros(ids.shape, bool)
    for r in range(len(ntok)):
        mask[r, first[r]: first[r] + ntok[r]] = True
    assert np.array_equal(lin[~mask], padded[~mask])
    d = np.abs(lin - padded)[mask]
    assert d.max() <= ATOL, f"{case} ({model.gemm_mode}): max |d| = {d.max():.3e}"
    print(f"linear form vs padded launch, {case} ({model.gemm_mode}): max |dlogit| = {d.max():.2e}")
    if npad.max() >= 4:
        assert stages.get("hyena_pad_row", (0, 0))[1] == 4, stages  # one shared row, once per layer
        # FFT launches: one per layer and transform size among the reads THEMSELVES
        own = np.array([x for b in batches for x in b]) + 4
        sizes = {_fft_points(n) for n in own[(npad >= 4) & (own > 2560)]} | {_fft_points(n) for n in ntok[(npad < 4) & (ntok > 2560)]}
        small = len({z for z in sizes if z <= 16384})
        assert stages.get("hyena_fft", (0, 0))[1] == 4 * small, (stages, sizes)  # (the shared row's own launch is inside "hyena_pad_row")


def test_shared_prefix_forms_error_vs_float64(hip_lib):
    """Which of them is nearer the truth?  A left-padded batch whose longest row takes an FFT form, against a float64
    evaluation of the oracle on the padded batch: the padded launch, the exact shared-prefix form, the linear form and
    the torch-fp32 oracle itself.  The linear form transforms the reads' own tokens only (shorter transforms, the short
    rows on the exactly causal block-Toeplitz kernel), so its error must not exceed the padded launch's by more than
    fp32 noise -- the differences between the forms that the tests above bound by 2e-4 are rounding, not signal."""
    import torch

    from deepchopper_amd import ops
    from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict
    from oracle import forward_ref as fr

    sd = synthetic_state_dict(3)
    model = ops.HipModel(sd, HyenaDNAConfig())
    rng = np.random.default_rng(64)
    ids, q, first, ntok, npad = _padded_launch(rng, [[5200, 700, 1900, 3100]])
    ref64 = fr.forward(sd, ids.astype(np.int64), q, dtype=torch.float64).numpy()
    ref32 = fr.forward(sd, ids.astype(np.int64), q).numpy()
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    errs = {"oracle_f32": float(np.abs(ref32 - ref64).max())}
    for mode in ("f32", "bf16x6", "fp16x3"):
        model.set_gemm_mode(mode)
        errs[f"{mode}/padded"] = float(np.abs(model.forward_ragged(d_ids, d_q, first, ntok).cpu().numpy() - ref64).max())
        for form in ("exact", "linear"):
            model.set_prefix_mode(form)
            out = model.forward_shared_prefix(d_ids, d_q, first, ntok, npad).cpu().numpy()
            errs[f"{mode}/{form}"] = float(np.abs(out - ref64).max())
    print("max |logit - float64 oracle| on a left-padded batch (longest row 5 200 tokens):", errs)
    for mode in ("f32", "bf16x6", "fp16x3"):
        assert errs[f"{mode}/linear"] <= 1.5 * max(errs[f"{mode}/padded"], errs["oracle_f32"]) + 2e-5, errs
        assert errs[f"{mode}/exact"] <= 1.5 * max(errs[f"{mode}/padded"], errs["oracle_f32"]) + 2e-5, errs
        assert errs[f"{mode}/linear"] < ATOL and errs[f"{mode}/padded"] < ATOL


# ----------------------------------------------------------------------------------------------------------------
# the linear prefix form where it is worst, and over many seeds
# ----------------------------------------------------------------------------------------------------------------
_F64_CACHE: dict = {}


def _form_errors(model, ops, ids, q, first, ntok, npad):
    """max |dlogit| over the rows' positions of both shared-prefix forms against the padded launch; returns the three logit
    arrays as well."""
    d_ids, d_q = ops.to_device(ids), ops.to_device(q)
    padded = model.forward_ragged(d_ids, d_q, first, ntok).cpu().numpy()
    out = {}
    try:
        for form in ("exact", "linear"):
            model.set_prefix_mode(form)
            out[form] = model.forward_shared_prefix(d_ids, d_q, first, ntok, npad).cpu().numpy()
    finally:
        model.set_prefix_mode("exact")
    mask = np.zeros(ids.shape, bool)
    for r in range(len(ntok)):
        mask[r, first[r]: first[r] + ntok[r]] = True
    return padded, out, mask


def test_linear_prefix_form_at_the_longest_pad_run(env):
    """A ~100-token read behind a ~32 600-token pad run (the batch's longest read fills the model's whole context: the shared
    row takes the four-step 2^15 transform), the case in which `k * (z - z_pad) + F` leans hardest on F: both forms against
    the padded launch AND against a float64 evaluation of the oracle on the padded batch.  The linear form must be no
    further from the float64 truth than the padded launch is (its own reads run on shorter, exactly causal kernels)."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    rng = np.random.default_rng(32769)
    batches = [[32769, 100, 4000], [32768 - 3, 150]]
    ids, q, first, ntok, npad = _padded_launch(rng, batches)
    assert int(npad.max()) >= 32600
    padded, forms, mask = _form_errors(model, ops, ids, q, first, ntok, npad)
    assert model.prefix_violations() == 0
    ref64 = _F64_CACHE.get("longest_pad_run")  # (the same weights and inputs in every GEMM mode: evaluated once)
    if ref64 is None:
        ref64 = np.zeros_like(padded, dtype=np.float64)
        r = 0
        for b in batches:
            c0 = ids.shape[1] - max(b)
            ref64[r: r + len(b), c0:] = fr.forward(sd, ids[r: r + len(b), c0:].astype(np.int64), q[r: r + len(b), c0:], dtype=torch.float64).numpy()
            r += len(b)
        _F64_CACHE["longest_pad_run"] = ref64
    e = {"padded": float(np.abs(padded - ref64)[mask].max())}
    for form in ("exact", "linear"):
        e[f"{form}_vs_padded"] = float(np.abs(forms[form] - padded)[mask].max())
        e[form] = float(np.abs(forms[form] - ref64)[mask].max())
    # the short reads' own tokens (what the chop stage consumes) on their own
    own = np.zeros_like(mask)
    for r in range(len(ntok)):
        own[r, first[r] + npad[r]: first[r] + ntok[r]] = True
    e["linear_own_tokens"] = float(np.abs(forms["linear"] - ref64)[own].max())
    e["padded_own_tokens"] = float(np.abs(padded - ref64)[own].max())
    print(f"longest pad run ({model.gemm_mode}): max |dlogit| {({k: f'{v:.2e}' for k, v in e.items()})}")
    assert e["linear_vs_padded"] <= ATOL and e["exact_vs_padded"] <= ATOL, e
    assert e["linear"] <= ATOL and e["exact"] <= ATOL and e["padded"] <= ATOL, e
    assert e["linear"] <= 1.5 * e["padded"] + 2e-5, e
    lab = lambda x: x[..., 1] > x[..., 0]
    sure = (np.abs(ref64[..., 1] - ref64[..., 0]) > 2 * ATOL) & mask
    assert np.array_equal(lab(forms["linear"])[sure], lab(ref64)[sure])


def test_linear_prefix_form_over_twenty_seeds(env):
    """The driver's reference_b12_nanopore check saw 1.64e-4 against the torch-fp32 oracle on one seed (82 % of the suite's
    2e-4).  Twenty seeds of the same nanopore-like length mix in the reference's batches of 12: max |dlogit| of the linear
    form against the padded launch (every seed) and against the float64 oracle (five seeds, one batch each), printed as a
    distribution; argmax labels of the linear form, the exact form and the padded launch agree wherever the margin exceeds the
    observed difference."""
    torch, ops, sd, model = env
    from oracle import forward_ref as fr

    vs_padded, vs_f64, padded_vs_f64, flips = [], [], [], 0
    for seed in range(20):
        rng = np.random.default_rng(1000 + seed)
        batches = [list(_nanopore_lens(rng, 12, 12000)) for _ in range(3)]
        ids, q, first, ntok, npad = _padded_launch(rng, batches)
        padded, forms, mask = _form_errors(model, ops, ids, q, first, ntok, npad)
        d = np.abs(forms["linear"] - padded)
        vs_padded.append(float(d[mask].max()))
        margin = np.abs(padded[..., 1] - padded[..., 0])
        lab = lambda x: x[..., 1] > x[..., 0]
        diff = lab(forms["linear"]) != lab(padded)
        assert not (diff & mask & (margin > 2 * d.max(axis=-1))).any()
        flips += int((diff & mask).sum())
        if seed % 4 == 0:  # float64 truth for the first batch (12 rows, its own padded width)
            b = batches[0]
            c0 = ids.shape[1] - max(b)
            ref64 = _F64_CACHE.get(("seeds", seed))
            if ref64 is None:
                ref64 = _F64_CACHE[("seeds", seed)] = fr.forward(sd, ids[: len(b), c0:].astype(np.int64), q[: len(b), c0:], dtype=torch.float64).numpy()
            vs_f64.append(float(np.abs(forms["linear"][: len(b), c0:] - ref64).max()))
            padded_vs_f64.append(float(np.abs(padded[: len(b), c0:] - ref64).max()))
    assert model.prefix_violations() == 0
    v = np.array(vs_padded)
    print(f"linear form vs padded launch over 20 seeds ({model.gemm_mode}): median {np.median(v):.2e}, p90 {np.quantile(v, 0.9):.2e}, "
          f"max {v.max():.2e}; vs float64 oracle (5 seeds): linear max {max(vs_f64):.2e}, padded launch max {max(padded_vs_f64):.2e}; "
          f"labels differing inside the noise margin: {flips}")
    assert v.max() <= ATOL, v
    assert max(vs_f64) <= ATOL and max(vs_f64) <= 1.5 * max(padded_vs_f64) + 2e-5, (vs_f64, padded_vs_f64)
