"""The oracle against every known-answer vector / fixture the reference's own tests hold for the
hot path.  CPU only."""
import os

import numpy as np
import pytest

from oracle import dc_oracle as o


def test_majority_voting(kats):
    for c in kats["majority_voting"]["cases"]:
        assert o.majority_voting(c["labels"], c["window"]).tolist() == c["expected"]
        assert o.majority_voting_literal(c["labels"], c["window"]) == c["expected"]


def test_id_list2seq(kats):
    for c in kats["id_list2seq"]["cases"]:
        assert o.id_list2seq(c["ids"]) == c["expected"]


def test_summary_predict(kats):
    for c in kats["summary_predict"]["cases"]:
        p, l = o.summary_predict(c["predictions"], c["labels"], c["ignore"])
        assert [x.tolist() for x in p] == c["expected_predictions"]
        assert [x.tolist() for x in l] == c["expected_labels"]


def test_get_label_region(kats):
    for c in kats["get_label_region"]["cases"]:
        exp = [tuple(x) for x in c["expected"]]
        assert o.get_label_region(c["labels"]) == exp
        assert o.get_label_region_literal(c["labels"]) == exp


def test_get_label_region_index0_quirk():
    # utils.rs:680-683: `start == 0` doubles as "no open run"
    assert o.get_label_region_literal([1, 1, 1, 0]) == [(1, 3)]
    assert o.get_label_region([1, 1, 1, 0]) == [(1, 3)]
    assert o.get_label_region_literal([1, 0, 1, 1]) == [(2, 4)]
    assert o.get_label_region([1, 0, 1, 1]) == [(2, 4)]
    assert o.get_label_region([1]) == [] == o.get_label_region_literal([1])


def test_remove_intervals_and_unmapped(kats):
    for c in kats["remove_intervals_and_keep_left"]["cases"]:
        parts, _ = o.remove_intervals_and_keep_left(c["seq"].encode(), [tuple(x) for x in c["intervals"]])
        assert [p.decode() for p in parts] == c["expected"]
    for c in kats["generate_unmaped_intervals"]["cases"]:
        got = o.generate_unmaped_intervals([tuple(x) for x in c["intervals"]], c["total_length"])
        assert got == [tuple(x) for x in c["expected"]]


def test_kmer_kats(kats):
    for c in kats["seq_to_kmers"]["cases"]:
        assert len(o.seq_to_kmers(c["seq"].encode(), c["k"], c["overlap"])) == c["expected_len"]
    for c in kats["generate_kmers"]["cases"]:
        assert [k.decode() for k in o.generate_kmers(c["bases"], c["k"])] == c["expected"]
    for c in kats["generate_kmers_table"]["cases"]:
        assert {k.decode(): v for k, v in o.generate_kmers_table(c["bases"], c["k"]).items()} == c["expected"]
    for c in kats["kmers_roundtrip"]["cases"]:
        s = c["seq"].encode()
        assert o.kmers_to_seq(o.seq_to_kmers(s, c["k"], True)) == s
    for c in kats["to_kmer_target_region"]["cases"]:
        if "error" in c:
            with pytest.raises(o.EncodingError, match=c["error"]):
                o.to_kmer_target_region(c["start"], c["end"], c["k"], c["seq_len"])
        else:
            assert list(o.to_kmer_target_region(c["start"], c["end"], c["k"], c["seq_len"])) == c["expected"]
    for c in kats["to_original_targtet_region"]["cases"]:
        assert list(o.to_original_targtet_region(c["start"], c["end"], c["k"])) == c["expected"]
    for c in kats["seq_to_kmers_and_offset"]["cases"]:
        r = o.seq_to_kmers_and_offset(c["seq"].encode(), c["k"], c["overlap"])
        assert len(r) == c["expected_len"]
        assert (r[0][0].decode(), list(r[0][1])) == (c["first"][0], c["first"][1])
        assert (r[1][0].decode(), list(r[1][1])) == (c["second"][0], c["second"][1])
        if "last" in c:
            assert (r[-1][0].decode(), list(r[-1][1])) == (c["last"][0], c["last"][1])
    for c in kats["vectorize_target"]["cases"]:
        if "error" in c:
            with pytest.raises(o.EncodingError, match=c["error"]):
                o.vectorize_target(c["start"], c["end"], c["length"])
        else:
            assert o.vectorize_target(c["start"], c["end"], c["length"]).tolist() == c["expected"]
    for c in kats["vectorize_targets"]["cases"]:
        if "error" in c:
            with pytest.raises(o.EncodingError, match=c["error"]):
                o.vectorize_targets(c["targets"], c["length"])
        else:
            assert o.vectorize_targets(c["targets"], c["length"]).tolist() == c["expected"]
    for c in kats["parse_target_from_id"]["cases"]:
        assert o.parse_target_from_id(c["id"]) == [tuple(x) for x in c["expected"]]


def test_tensor_encoder_kats(kats, golden_dir):
    for c in kats["tensor_encoder"]["cases"]:
        recs = o.read_fastq_records(os.path.join(golden_dir, c["file"]))
        inp, tgt, qual = o.tensor_encode(recs, 3, b"ATCGN", 33, c["vectorized_target"], c["max_width"], c["max_seq_len"])
        if "shapes" in c:
            assert [list(inp.shape), list(tgt.shape), list(qual.shape)] == c["shapes"]
        if "target_0_0" in c:
            assert tgt[0, 0].tolist() == c["target_0_0"]
        if "ones" in c:
            s, e = c["ones"]
            assert np.all(tgt[0, 0, s:e] == 1) and tgt[0, 0].sum() == e - s
            table = {v: k for k, v in o.generate_kmers_table(b"ATCGN", 3).items()}
            assert o.kmerids_to_seq(inp[0, 0, s:e].tolist(), table).decode() == c["kmer_seq_of_target"]


def _load_fixture(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    return {k: d[k] for k in d.files}


FIXTURES = [f"eval_chunk{c}_{i}.npz" for c in (0, 1) for i in (0, 1, 2)]


def test_load_predict_kat(kats, golden_dir):
    for c in kats["load_predict"]["cases"]:
        p = o.load_predicts_from_batch(_load_fixture(golden_dir, c["file"]))
        assert len(p) == c["n_predicts"]


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_layout_pins_tokeniser(golden_dir, name):
    """The reference's real CustomWriter outputs pin T3-T5: left pad id 4, [SEP]=1 last, labels
    -100 exactly on pad + SEP, quals L2-normalised, id row = [len, truncated, ascii...]."""
    d = _load_fixture(golden_dir, name)
    seq, tgt, qual, idm = d["seq"], d["target"].astype(np.int64), d["qual"], d["id"].astype(np.int64)
    assert set(np.unique(seq).tolist()) <= {1, 4, 7, 8, 9, 10, 11}
    for b in range(seq.shape[0]):
        row = seq[b]
        npad = int(np.argmax(row != 4))
        assert np.all(row[:npad] == 4) and row[-1] == 1 and np.all(row[npad:-1] >= 7)
        assert np.all(tgt[b, :npad] == -100) and tgt[b, -1] == -100 and np.all(tgt[b, npad:-1] != -100)
        assert np.all(qual[b, :npad] == 0) and qual[b, -1] == 0
        assert abs(float(np.sum(qual[b].astype(np.float64) ** 2)) - 1.0) < 1e-5
        n = int(idm[b, 0])
        assert idm[b, 1] == 0 and np.all(idm[b, 2 + n :] == 0) and np.all(idm[b, 2 : 2 + n] > 0)
        # re-tokenising the decoded read reproduces the row (oracle T3/T5 vs the fixture)
        read = o.id_list2seq(row[npad:-1]).encode()
        assert np.array_equal(o.tokenize_chars(read), row[npad:].astype(np.int64))
        # quality values are phred / ||phred||: recover the integer phred row and re-normalise
        v = qual[b, npad:].astype(np.float64)
        vmin = v[v > 0].min()
        phred = None
        for m in range(1, 94):
            nrm = m / vmin
            cand = np.rint(v * nrm)
            if np.max(np.abs(cand - v * nrm)) < 0.02 and abs(np.sqrt(np.sum(cand**2)) - nrm) < 1e-3 * nrm:
                phred = cand
                break
        assert phred is not None
        ours = o.l2_normalize_quals(phred.astype(np.float32))
        np.testing.assert_allclose(ours, qual[b, npad:], rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_smoothing_consistency(golden_dir, name):
    """S0->S5 on real predictions: vectorised oracle == literal restatement; invariants hold."""
    preds = o.load_predicts_from_batch(_load_fixture(golden_dir, name))
    for p in preds.values():
        lab = p.prediction
        assert len(lab) == len(p.seq)
        sm = o.majority_voting(lab, 21)
        assert sm.tolist() == o.majority_voting_literal(lab.tolist(), 21)
        regs = o.get_label_region(sm)
        assert regs == o.get_label_region_literal(sm.tolist())
        ivs = p.smooth_and_select_intervals(21, 13, 20)
        assert all(e - s >= 13 for s, e in ivs) and len(ivs) <= 20
        keep = o.generate_unmaped_intervals(ivs, len(lab))
        covered = sum(e - s for s, e in keep) + sum(e - s for s, e in ivs)
        assert covered in (len(lab), len(lab) - 1)  # the tail segment drops the last base


def test_model_input_fixture(golden_dir):
    d = _load_fixture(golden_dir, "model_inputs.npz")
    ids, q = d["input_ids"], d["input_quals"]
    assert ids.shape == q.shape == (12, 3223)
    assert set(np.unique(ids).tolist()) <= {1, 4, 7, 8, 9, 10, 11}
    assert np.all(ids[:, -1] == 1)
    np.testing.assert_allclose(np.sum(q.astype(np.float64) ** 2, axis=1), 1.0, atol=1e-5)


def test_normalize_and_tokenise_semantics():
    assert o.normalize_seq("ACGTNacgtnUu.~- \t\r\nXRY", False) == b"ACGTNACGTNTT---NNN"
    assert o.normalize_seq("RYKMbdhv", True) == b"RYKMBDHV"
    assert o.tokenize_chars(b"ACGTN-").tolist() == [7, 8, 9, 10, 11, 6, 1]
    assert o.encode_qual("!+5?I", 33).tolist() == [0, 10, 20, 30, 40]
    assert o.encode_qual(b" ", 33).tolist() == [255]  # u8 wrap, release profile
    t = o.tokenize_and_align_labels_and_quals_ids("r1", b"ACGT", o.encode_qual("IIII"), max_length=4)
    assert t["input_ids"].tolist() == [7, 8, 9, 1] and t["id"][1] == 1 and t["labels"].tolist() == [0, 0, 0, -100]
    feats = [o.tokenize_and_align_labels_and_quals_ids(f"r{i}", s, o.encode_qual("I" * len(s))) for i, s in enumerate([b"ACG", b"A"])]
    b = o.collate(feats)
    assert b["input_ids"].tolist() == [[7, 8, 9, 1], [4, 4, 7, 1]]
    assert b["labels"].tolist() == [[0, 0, 0, -100], [-100, -100, 0, -100]]
    assert b["id"][1, :4].tolist() == [2, 0, ord("r"), ord("1")]
    with pytest.raises(RuntimeError):
        o.collate([o.tokenize_and_align_labels_and_quals_ids("x" * 130, b"ACG", o.encode_qual("III"))])


def test_l2_normalize_matches_torch():
    import torch

    rng = np.random.default_rng(0)
    for n in (1, 5, 513, 2001, 32769):
        q = rng.integers(0, 60, n).astype(np.float32)
        q[-1] = 0
        ref = torch.nn.functional.normalize(torch.from_numpy(q), dim=0).numpy()
        np.testing.assert_allclose(o.l2_normalize_quals(q), ref, rtol=3e-7, atol=0)
    assert np.all(o.l2_normalize_quals(np.zeros(4, np.float32)) == 0)


def test_vectorised_vs_literal_random():
    rng = np.random.default_rng(5)
    for n in (1, 2, 5, 20, 21, 22, 64, 65, 300):
        for w in (1, 2, 3, 21, 22, 51):
            lab = (rng.random(n) < 0.4).astype(np.int8)
            assert o.majority_voting(lab, w).tolist() == o.majority_voting_literal(lab.tolist(), w)
            assert o.get_label_region(lab) == o.get_label_region_literal(lab.tolist())


def test_chop_record_semantics():
    seq = b"A" * 100 + b"C" * 30 + b"G" * 100
    qual = b"I" * 230
    lab = np.zeros(230, np.int8)
    lab[100:130] = 1
    p = o.Predict(lab, seq.decode(), "r", False)
    out = o.chop_read(p, b"r", seq, qual)
    assert [r[0] for r in out] == [b"r|0:100|I", b"r|130:229|I"]
    assert out[1][1] == b"G" * 99
    lab2 = np.zeros(230, np.int8)
    lab2[200:230] = 1
    out = o.chop_read(o.Predict(lab2, seq.decode(), "r", False), b"r", seq, qual)
    assert [r[0] for r in out] == [b"r|0:200|T"]
    assert o.chop_read(o.Predict(lab[:100], seq[:100].decode(), "r", False), b"r", seq[:100], qual[:100]) == [(b"r", seq[:100], qual[:100])]
