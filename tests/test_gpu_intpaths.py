"""GPU parity tests for the integer / byte kernels (tokenise, k-mer, smoothing) through the C ABI.
Bit-exact against the CPU oracle on the same seeded inputs, against the reference's committed
fixtures, and through size-independent properties at larger sizes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(hip_lib):
    import torch

    from deepchopper_amd import _lib, ops

    assert _lib.device_count() >= 1, "GPU tests need a HIP device; there is no CPU fallback"
    return torch, ops


def _rand_reads(rng, n, lo, hi, alphabet=b"ACGTN", qlo=34, qhi=74):
    seqs, quals = [], []
    al = np.frombuffer(alphabet, dtype=np.uint8)
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        seqs.append(al[rng.integers(0, len(al), L)].tobytes())
        quals.append(rng.integers(qlo, qhi, L).astype(np.uint8).tobytes())
    return seqs, quals


# ------------------------------------------------------------------------------------------
# tokenise
# ------------------------------------------------------------------------------------------
def _oracle_batch(seqs, quals, ids, max_length):
    from oracle import dc_oracle as o

    feats = []
    for i, s, q in zip(ids, seqs, quals):
        rec = o.parse_fastq_record(i, s.decode("latin1"), q.decode("latin1"))
        feats.append(o.tokenize_and_align_labels_and_quals_ids(rec["id"], rec["seq"], rec["qual"], rec["target"], max_length))
    return o.collate(feats), feats


@pytest.mark.parametrize("ids_i64", [True, False])
def test_tokenize_batch_matches_oracle(env, ids_i64):
    torch, ops = env
    rng = np.random.default_rng(11)
    seqs, quals = _rand_reads(rng, 37, 1, 700, alphabet=b"ACGTNacgtnUuRYKM-.~X")
    ids = [f"r{i}" for i in range(len(seqs))]
    ob, feats = _oracle_batch(seqs, quals, ids, 32769)
    sb, offs = ops.pack_reads(seqs)
    qb, _ = ops.pack_reads(quals)
    lmax = ob["input_ids"].shape[1]
    out = ops.tokenize_batch(ops.to_device(sb), ops.to_device(qb), ops.to_device(offs), max_length=32769,
                             row_stride=lmax, pad_left=True, ids_i64=ids_i64)
    torch.cuda.synchronize()
    assert np.array_equal(out["input_ids"].cpu().numpy().astype(np.int64), ob["input_ids"])
    assert np.array_equal(out["labels"].cpu().numpy(), ob["labels"])
    got_q = out["input_quals"].cpu().numpy()
    assert np.array_equal(got_q.view(np.uint32), ob["input_quals"].view(np.uint32)), "quals must be bit-exact vs oracle"
    assert out["n_tokens"].cpu().numpy().tolist() == [len(f["input_ids"]) for f in feats]
    assert np.all(out["status"].cpu().numpy() == 0)


@pytest.mark.parametrize("lens,stride_extra", [
    ([1, 2, 15, 16, 17, 31, 33, 1], 0),        # every read shorter than two 16-byte chunks; first / last windows leave the buffer
    ([1000, 7, 1023, 1], 0),                    # row of 1 024 columns: the one-step form at its limit
    ([1024, 1, 999], 0),                        # 1 025 columns: first two-pass row
    ([2500, 16, 3000, 48], 0),                  # several steps per wave
    ([40, 700, 3], 37),                         # rows wider than the longest read (extra left pad)
])
@pytest.mark.parametrize("pad_left", [True, False])
def test_tokenize_chunk_edges(env, lens, stride_extra, pad_left):
    """The round-3 tokenizer moves 16 bases per lane: reads shorter than a chunk, windows that would leave the packed
    buffers (first / last read), rows that end inside a chunk, the one-step (<= 1 024 columns) and the two-pass form,
    left and right padding -- all bit-exact against the oracle's per-read features."""
    torch, ops = env
    rng = np.random.default_rng(sum(lens) + stride_extra)
    al = np.frombuffer(b"ACGTNacgtnU-X ", dtype=np.uint8)
    seqs = [al[rng.integers(0, len(al) - 1, n)].tobytes() for n in lens]  # (no blanks: they raise a status flag)
    quals = [rng.integers(34, 74, n).astype(np.uint8).tobytes() for n in lens]
    ids = [f"e{i}" for i in range(len(lens))]
    _, feats = _oracle_batch(seqs, quals, ids, 32769)
    sb, offs = ops.pack_reads(seqs)
    qb, _ = ops.pack_reads(quals)
    stride = max(lens) + 1 + stride_extra
    for ids_i64 in (False, True):
        out = ops.tokenize_batch(ops.to_device(sb), ops.to_device(qb), ops.to_device(offs), max_length=32769,
                                 row_stride=stride, pad_left=pad_left, ids_i64=ids_i64)
        got_ids = out["input_ids"].cpu().numpy().astype(np.int64)
        got_q = out["input_quals"].cpu().numpy()
        got_l = out["labels"].cpu().numpy()
        assert np.all(out["status"].cpu().numpy() == 0)
        for r, f in enumerate(feats):
            n = len(f["input_ids"])
            a = stride - n if pad_left else 0
            assert np.array_equal(got_ids[r, a: a + n], f["input_ids"]), (r, ids_i64)
            assert np.array_equal(got_q[r, a: a + n].view(np.uint32), np.asarray(f["input_quals"], np.float32).view(np.uint32))
            assert np.array_equal(got_l[r, a: a + n], np.asarray(f["labels"], np.int8))
            rest = np.ones(stride, bool)
            rest[a: a + n] = False
            assert np.all(got_ids[r, rest] == 4) and np.all(got_q[r, rest] == 0) and np.all(got_l[r, rest] == -100)


def test_tokenize_truncation_and_flags(env):
    torch, ops = env
    from deepchopper_amd import _lib

    rng = np.random.default_rng(3)
    seqs, quals = _rand_reads(rng, 5, 30, 60)
    seqs[1] = seqs[1][:10] + b" " + seqs[1][11:]       # whitespace -> normalize drops it
    quals[2] = quals[2][:5] + b"\x10" + quals[2][6:]   # quality byte below the offset
    max_length = 40
    ob, feats = _oracle_batch([seqs[0], seqs[3], seqs[4]], [quals[0], quals[3], quals[4]], ["a", "b", "c"], max_length)
    sb, offs = ops.pack_reads(seqs)
    qb, _ = ops.pack_reads(quals)
    out = ops.tokenize_batch(ops.to_device(sb), ops.to_device(qb), ops.to_device(offs), max_length=max_length,
                             row_stride=max_length, pad_left=True, ids_i64=True)
    st = out["status"].cpu().numpy()
    ntok = out["n_tokens"].cpu().numpy()
    for r, s in enumerate(seqs):
        assert bool(st[r] & 1) == (len(s) >= max_length)
        assert ntok[r] == min(len(s), max_length - 1) + 1
    assert st[1] & 2 and st[2] & 4
    ids = out["input_ids"].cpu().numpy()
    for k, r in enumerate((0, 3, 4)):
        n = len(feats[k]["input_ids"])
        assert np.array_equal(ids[r, max_length - n:], feats[k]["input_ids"])
        assert np.all(ids[r, : max_length - n] == 4)
        np.testing.assert_array_equal(out["input_quals"].cpu().numpy()[r, max_length - n:], feats[k]["input_quals"])


@pytest.mark.parametrize("qual_offset", [33, 64, 128, 129, 200])
def test_tokenize_quality_offsets_and_underflow_flag(env, qual_offset):
    """encode_qual with other offsets (src/python.rs:26: `q - offset` as u8): the word-wise "a byte is below the offset" test
    holds for offsets <= 128 and hands over to a per-byte test above; shifting qualities and offset together must not change
    one bit, and a single byte below the offset -- in the first, a middle or the last chunk of a read -- raises the flag of
    that read only."""
    torch, ops = env
    rng = np.random.default_rng(qual_offset)
    lens = [5, 16, 17, 100, 1000, 1023]
    al = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [al[rng.integers(0, 4, n)].tobytes() for n in lens]
    ph = [rng.integers(1, 41, n).astype(np.uint8) for n in lens]
    sb, offs = ops.pack_reads(seqs)
    stride = max(lens) + 1

    def run(off, phreds):
        qb, _ = ops.pack_reads([(p + off).astype(np.uint8).tobytes() for p in phreds])
        out = ops.tokenize_batch(ops.to_device(sb), ops.to_device(qb), ops.to_device(offs), max_length=32769, row_stride=stride,
                                 pad_left=True, ids_i64=False, qual_offset=off)
        return out["input_quals"].cpu().numpy(), out["status"].cpu().numpy()

    base_q, base_st = run(33, ph)
    q, st = run(qual_offset, ph)
    assert np.array_equal(q.view(np.uint32), base_q.view(np.uint32)) and np.all(st == 0) and np.all(base_st == 0)
    for r, pos in ((3, 0), (4, 500), (5, 1022), (0, 4)):
        bad = [p.astype(np.int64) for p in ph]
        bad[r][pos] = -1  # one byte just below the offset
        _, st = run(qual_offset, bad)
        assert [bool(x & 4) for x in st] == [i == r for i in range(len(lens))], (r, pos, st)


def test_tokenize_against_reference_fixture(env, golden_dir):
    """Real reads (decoded from the reference's .pt fixture): our kernel reproduces the
    reference's own input_ids / labels rows and its quality normalisation."""
    torch, ops = env
    from oracle import dc_oracle as o

    d = np.load(os.path.join(golden_dir, "eval_chunk0_1.npz"))
    seq, tgt, qual = d["seq"], d["target"].astype(np.int64), d["qual"]
    reads, phreds = [], []
    for b in range(seq.shape[0]):
        npad = int(np.argmax(seq[b] != 4))
        reads.append(o.id_list2seq(seq[b, npad:-1]).encode())
        v = qual[b, npad:-1].astype(np.float64)
        vmin = v[v > 0].min()
        for m in range(1, 94):
            nrm = m / vmin
            cand = np.rint(v * nrm)
            if np.max(np.abs(cand - v * nrm)) < 0.02 and abs(np.sqrt(np.sum(cand**2)) - nrm) < 1e-3 * nrm:
                break
        phreds.append((cand + 33).astype(np.uint8).tobytes())
    sb, offs = ops.pack_reads(reads)
    qb, _ = ops.pack_reads(phreds)
    out = ops.tokenize_batch(ops.to_device(sb), ops.to_device(qb), ops.to_device(offs), max_length=32769,
                             row_stride=seq.shape[1], pad_left=True, ids_i64=True)
    assert np.array_equal(out["input_ids"].cpu().numpy(), seq.astype(np.int64))
    assert np.array_equal(out["labels"].cpu().numpy().astype(np.int64), tgt)
    np.testing.assert_allclose(out["input_quals"].cpu().numpy(), qual, rtol=2e-6, atol=1e-9)


def test_normalize_and_encode_qual(env):
    torch, ops = env
    from oracle import dc_oracle as o

    rng = np.random.default_rng(5)
    reads = [bytes(rng.integers(0, 256, int(rng.integers(0, 900))).astype(np.uint8)) for _ in range(23)]
    sb, offs = ops.pack_reads(reads)
    for iupac in (False, True):
        out, out_len = ops.normalize_seq_batch(ops.to_device(sb), ops.to_device(offs), iupac)
        out, out_len = out.cpu().numpy(), out_len.cpu().numpy()
        for r, raw in enumerate(reads):
            exp = o.normalize_seq(raw, iupac)
            assert out_len[r] == len(exp)
            assert out[offs[r]: offs[r] + out_len[r]].tobytes() == exp
    q = ops.encode_qual_bytes(ops.to_device(sb), 33).cpu().numpy()
    assert np.array_equal(q, (sb - np.uint8(33)))


# ------------------------------------------------------------------------------------------
# k-mer encoder
# ------------------------------------------------------------------------------------------
def _gpu_tensor_encode(ops, torch, recs, k, bases, vectorized, max_width=0, max_seq_len=0):
    from deepchopper_amd import rs_api

    return rs_api.tensor_encode_records(recs, k, bases, 33, vectorized, max_width, max_seq_len)


def test_kmer_encoder_reference_files(env, kats, golden_dir):
    torch, ops = env
    from oracle import dc_oracle as o

    for c in kats["tensor_encoder"]["cases"]:
        recs = o.read_fastq_records(os.path.join(golden_dir, c["file"]))
        exp = o.tensor_encode(recs, 3, b"ATCGN", 33, c["vectorized_target"], c["max_width"], c["max_seq_len"])
        got = _gpu_tensor_encode(ops, torch, recs, 3, b"ATCGN", c["vectorized_target"], c["max_width"], c["max_seq_len"])
        for g, e in zip(got, exp):
            assert g.dtype == np.int32 and g.shape == e.shape
            assert np.array_equal(g, e)
        if "shapes" in c:
            assert [list(x.shape) for x in got] == c["shapes"]


def test_kmer_encoder_random_and_dropped_reads(env):
    torch, ops = env
    from oracle import dc_oracle as o

    rng = np.random.default_rng(17)
    for k, bases in ((3, b"ACGTN"), (1, b"ACGT"), (5, b"ATCGN"), (8, b"ACGT")):
        seqs, quals = _rand_reads(rng, 40, 1, 3000, alphabet=b"ACGT")
        seqs[3] = seqs[3][:7] + b"X" + seqs[3][8:]  # not in bases -> read dropped
        seqs[5] = b"AC"[: min(2, k - 1)] if k > 1 else seqs[5]  # shorter than k -> skipped
        quals[5] = quals[5][: len(seqs[5])]
        quals[7] = bytes([126]) * len(quals[7])  # phred 93: the u8 window sum wraps for k >= 3
        recs = [(b"id%d|3:9" % i, s, q) for i, (s, q) in enumerate(zip(seqs, quals))]
        for vec in (False, True):
            exp = o.tensor_encode(recs, k, bases, 33, vec)
            got = _gpu_tensor_encode(ops, torch, recs, k, bases, vec)
            for g, e in zip(got, exp):
                assert np.array_equal(g, e)


# ------------------------------------------------------------------------------------------
# smoothing / intervals
# ------------------------------------------------------------------------------------------
def _check_against_oracle(ops, torch, label_list, window, min_iv, approved):
    from oracle import dc_oracle as o

    lab, offs = ops.pack_reads([l.astype(np.int8).tobytes() for l in label_list])
    dl = ops.to_device(lab.view(np.int8))
    out = ops.smooth_intervals_from_labels(dl, ops.to_device(offs), window=window, min_interval=min_iv,
                                           approved=approved, max_intervals=max(approved, 1))
    sm = out["smooth"].cpu().numpy()
    iv = out["intervals"].cpu().numpy()
    niv = out["n_intervals"].cpu().numpy()
    keep = out["keep"].cpu().numpy()
    nkeep = out["n_keep"].cpu().numpy()
    for r, l in enumerate(label_list):
        e_sm = o.majority_voting(l, window)
        assert np.array_equal(sm[offs[r]: offs[r + 1]], e_sm), f"read {r} smoothed labels differ"
        e_iv = o.smooth_label_region(l, window, min_iv, approved)
        assert niv[r] == len(e_iv), f"read {r}: {niv[r]} intervals, oracle {len(e_iv)}"
        assert [tuple(x) for x in iv[r, : niv[r]].tolist()] == e_iv
        if len(l) > 0:
            e_keep = o.generate_unmaped_intervals(e_iv, len(l))
            assert [tuple(x) for x in keep[r, : nkeep[r]].tolist()] == e_keep


def test_smooth_reference_kats(env, kats):
    torch, ops = env
    from oracle import dc_oracle as o

    for c in kats["majority_voting"]["cases"]:
        if not c["labels"]:
            continue
        lab = np.array(c["labels"], dtype=np.int8)
        out = ops.smooth_intervals_from_labels(ops.to_device(lab), ops.to_device(np.array([0, len(lab)], np.int64)),
                                               window=c["window"], min_interval=0, approved=64,
                                               apply_approved_rule=False, max_intervals=64)
        assert out["smooth"].cpu().numpy().tolist() == c["expected"]
    for c in kats["get_label_region"]["cases"]:
        if not c["labels"]:
            continue
        lab = np.array(c["labels"], dtype=np.int8)
        out = ops.smooth_intervals_from_labels(ops.to_device(lab), ops.to_device(np.array([0, len(lab)], np.int64)),
                                               window=1, min_interval=0, approved=64, apply_approved_rule=False,
                                               max_intervals=64)
        n = int(out["n_intervals"].cpu()[0])
        assert out["intervals"].cpu().numpy()[0, :n].tolist() == c["expected"]
    for c in kats["generate_unmaped_intervals"]["cases"]:
        n = c["total_length"]
        lab = np.zeros(n, np.int8)
        for s, e in c["intervals"]:
            lab[s:e] = 1
        out = ops.smooth_intervals_from_labels(ops.to_device(lab), ops.to_device(np.array([0, n], np.int64)),
                                               window=1, min_interval=1, approved=20, max_intervals=20)
        k = int(out["n_keep"].cpu()[0])
        assert out["keep"].cpu().numpy()[0, :k].tolist() == c["expected"]


def test_smooth_edge_cases(env):
    torch, ops = env
    rng = np.random.default_rng(23)
    cases = [np.array(x, dtype=np.int8) for x in (
        [1], [0], [1, 1], [1, 0], [0, 1], [1, 1, 1, 0], [1, 0, 1, 1], [1] * 64, [1] * 65, [0] * 63 + [1] * 66,
        [1] * 128, [0, 1] * 100, [1, 1, 0] * 70)]
    cases += [(rng.random(n) < p).astype(np.int8) for n in (2, 20, 21, 22, 63, 64, 65, 127, 128, 129, 1000, 4097)
              for p in (0.1, 0.5, 0.9)]
    for window, min_iv, approved in ((21, 13, 20), (1, 1, 20), (3, 2, 2), (22, 13, 20), (101, 5, 3), (5, 0, 20)):
        _check_against_oracle(ops, torch, cases, window, min_iv, approved)


def test_smooth_synthetic_label_streams(env):
    torch, ops = env
    from deepchopper_amd import synth

    labs = [synth.synth_label_stream(n, seed=43, stream=i) for i, n in enumerate((150, 512, 1000, 2000, 9999, 32768, 65536))]
    _check_against_oracle(ops, torch, labs, 21, 13, 20)
    # approved-count rule: many planted runs -> empty result
    many = synth.synth_label_stream(20000, seed=43, run_every=150, mean_run=40.0, stream=99)
    _check_against_oracle(ops, torch, [many], 21, 13, 20)


@pytest.mark.parametrize("name", [f"eval_chunk{c}_{i}.npz" for c in (0, 1) for i in (0, 1, 2)])
def test_smooth_from_logits_reference_fixtures(env, golden_dir, name):
    """S0-S5 on the reference's real predictions, starting from fp32 logits."""
    torch, ops = env
    from oracle import dc_oracle as o

    d = np.load(os.path.join(golden_dir, name))
    batch = {k: d[k] for k in d.files}
    preds = o.load_predicts_from_batch(batch)
    B, L, _ = batch["prediction"].shape
    tgt = batch["target"].astype(np.int64)
    first = np.array([int(np.argmax(tgt[b] != -100)) for b in range(B)], dtype=np.int32)
    length = np.array([int(np.sum(tgt[b] != -100)) for b in range(B)], dtype=np.int32)
    out = ops.smooth_intervals_from_logits(ops.to_device(batch["prediction"]), ops.to_device(first),
                                           ops.to_device(length), window=21, min_interval=13, approved=20)
    offs = out["offsets"].cpu().numpy()
    pred = out["pred"].cpu().numpy()
    sm = out["smooth"].cpu().numpy()
    niv = out["n_intervals"].cpu().numpy()
    iv = out["intervals"].cpu().numpy()
    idm = batch["id"].astype(np.int64)
    for b in range(B):
        rid = "".join(chr(c) for c in idm[b, 2: 2 + idm[b, 0]])
        p = preds[rid]
        assert np.array_equal(pred[offs[b]: offs[b + 1]], p.prediction)
        assert np.array_equal(sm[offs[b]: offs[b + 1]], p.smooth_label(21))
        e_iv = p.smooth_and_select_intervals(21, 13, 20)
        assert niv[b] == len(e_iv) and [tuple(x) for x in iv[b, : niv[b]].tolist()] == e_iv


@pytest.mark.parametrize("stride", [1001, 2048, 2049, 4096, 4097])
@pytest.mark.parametrize("window", [1, 2, 3, 21, 30, 31, 32, 33, 101])
def test_smooth_from_logits_forms_and_windows(env, stride, window):
    """Every form of the logits-mode smoothing kernel (64 threads / 2 048 positions, 128 / 4 096, 256 / 65 536, chosen by
    row stride) with both window evaluations (32-bit register windows up to 31, word-prefix counts above), reads whose
    lengths sit on the chunk (16), word (64) and form boundaries, first columns > 0, exact ties in the logits."""
    torch, ops = env
    from oracle import dc_oracle as o

    rng = np.random.default_rng(stride * 131 + window)
    lens = [x for x in (1, 15, 16, 17, 63, 64, 65, 150, 999, 1000, 1024, 2047, 2048, 4095, 4096, stride - 1, stride) if x <= stride]
    B = len(lens)
    logits = rng.standard_normal((B, stride, 2)).astype(np.float32)
    logits[:, ::7, 1] = logits[:, ::7, 0]  # exact ties -> label 0
    for b in range(B):  # runs long enough to survive the length filter
        for s0 in range(30, lens[b] - 40, 211):
            logits[b, stride - lens[b] + s0: stride - lens[b] + s0 + 35, 1] = logits[b, stride - lens[b] + s0: stride - lens[b] + s0 + 35, 0] + 1.0
    first = np.array([stride - n for n in lens], np.int32)
    length = np.array(lens, np.int32)
    out = ops.smooth_intervals_from_logits(ops.to_device(logits), ops.to_device(first), ops.to_device(length), window=window,
                                           min_interval=13, approved=20)
    offs = out["offsets"].cpu().numpy()
    pred, sm = out["pred"].cpu().numpy(), out["smooth"].cpu().numpy()
    niv, iv = out["n_intervals"].cpu().numpy(), out["intervals"].cpu().numpy()
    nk, kp = out["n_keep"].cpu().numpy(), out["keep"].cpu().numpy()
    for b, n in enumerate(lens):
        lg = logits[b, stride - n:]
        lab = (lg[:, 1] > lg[:, 0]).astype(np.int8)
        assert np.array_equal(pred[offs[b]: offs[b + 1]], lab), (b, n)
        assert np.array_equal(sm[offs[b]: offs[b + 1]], o.majority_voting(lab, window)), (b, n)
        e_iv = o.smooth_label_region(lab, window, 13, 20)
        assert niv[b] == len(e_iv) and [tuple(x) for x in iv[b, : niv[b]].tolist()] == e_iv, (b, n)
        assert [tuple(x) for x in kp[b, : nk[b]].tolist()] == o.generate_unmaped_intervals(e_iv, n), (b, n)


def test_smooth_properties_at_scale(env):
    """Size-independent properties at bench scale: idempotence of w=1, complement coverage."""
    torch, ops = env
    from deepchopper_amd import synth

    n_reads, L = 4096, 1000
    lab = np.concatenate([synth.synth_label_stream(L * 256, seed=43, stream=s) for s in range(n_reads // 256)])
    offs = np.arange(n_reads + 1, dtype=np.int64) * L
    dl, doffs = ops.to_device(lab), ops.to_device(offs)
    ident = ops.smooth_intervals_from_labels(dl, doffs, window=1, min_interval=0, approved=20,
                                             apply_approved_rule=False, max_intervals=4)
    assert torch.equal(ident["smooth"], dl)
    out = ops.smooth_intervals_from_labels(dl, doffs, window=21, min_interval=13, approved=20)
    sm = out["smooth"].cpu().numpy().reshape(n_reads, L)
    iv = out["intervals"].cpu().numpy()
    niv = out["n_intervals"].cpu().numpy()
    keep = out["keep"].cpu().numpy()
    nkeep = out["n_keep"].cpu().numpy()
    for r in range(0, n_reads, 37):
        cov = sum(e - s for s, e in iv[r, : niv[r]]) + sum(e - s for s, e in keep[r, : nkeep[r]])
        assert cov in (L, L - 1)
        for s, e in iv[r, : niv[r]]:
            assert e - s >= 13 and np.all(sm[r, s:e] == 1) and (e == L or sm[r, e] == 0)
    # smoothing twice with w=21 is not the identity in general, but it is deterministic
    out2 = ops.smooth_intervals_from_labels(dl, doffs, window=21, min_interval=13, approved=20)
    assert torch.equal(out2["smooth"], out["smooth"]) and torch.equal(out2["intervals"], out["intervals"])


def test_summary_predict(env, kats):
    torch, ops = env
    from oracle import dc_oracle as o

    rng = np.random.default_rng(9)
    pred = rng.integers(0, 2, (17, 700)).astype(np.int8)
    lab = rng.integers(0, 2, (17, 700)).astype(np.int8)
    lab[rng.random((17, 700)) < 0.3] = -100
    op, ol, offs = ops.summary_predict_dense(ops.to_device(pred), ops.to_device(lab), -100)
    op, ol, offs = op.cpu().numpy(), ol.cpu().numpy(), offs.cpu().numpy()
    ep, el = o.summary_predict(pred, lab, -100)
    for r in range(17):
        assert np.array_equal(op[offs[r]: offs[r + 1]], ep[r]) and np.array_equal(ol[offs[r]: offs[r + 1]], el[r])


@pytest.mark.parametrize("row_stride", [600, 3000, 5000])  # the 64-, 128- and 256-thread forms
def test_smooth_flags_reads_whose_lengths_break_the_contract(env, row_stride):
    """include/deepchopper_hip.h: first[r] + len[r] <= row_stride.  A caller's inconsistent length must not come back as
    truncated labels / intervals that look valid: the read is flagged n_intervals = n_keep = -1, its neighbours are
    computed as usual (the kernel used to clamp n to its form's capacity silently)."""
    torch, ops = env
    from oracle import dc_oracle as o

    rng = np.random.default_rng(row_stride)
    B = 5
    logits = rng.standard_normal((B, row_stride, 2)).astype(np.float32)
    first = np.array([0, 10, 5, 0, row_stride - 40], np.int32)
    length = np.array([row_stride, row_stride - 10, row_stride, row_stride + 4000, 40], np.int32)  # rows 2, 3 overrun
    first_bad = first.copy()
    out = ops.smooth_intervals_from_logits(ops.to_device(logits), ops.to_device(first_bad), ops.to_device(length),
                                           window=21, min_interval=13, approved=20, want_labels=False)
    niv, nkeep = out["n_intervals"].cpu().numpy(), out["n_keep"].cpu().numpy()
    iv = out["intervals"].cpu().numpy()
    assert niv[2] == -1 and niv[3] == -1 and nkeep[2] == -1 and nkeep[3] == -1
    for b in (0, 1, 4):
        lab = (logits[b, first[b]: first[b] + length[b], 1] > logits[b, first[b]: first[b] + length[b], 0]).astype(np.int8)
        want = o.smooth_label_region(lab, 21, 13, 20)
        assert niv[b] == len(want) and [tuple(x) for x in iv[b, : niv[b]].tolist()] == want
