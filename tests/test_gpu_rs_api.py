"""The PyO3-mirror API (`deepchopper_amd.rs_api`) exercised the way the reference's own tests exercise
`deepchopper.*` (tests/test_rust.py, the `#[test]` vectors in src/): same names, same arguments, same
answers -- computed on the GPU."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def r(hip_lib):
    from deepchopper_amd import _lib, rs_api

    assert _lib.device_count() >= 1
    return rs_api


def test_majority_voting_vectors(r, kats):
    for c in kats["majority_voting"]["cases"]:
        assert r.majority_voting(c["labels"], c["window"]) == c["expected"]


def test_get_label_region_vectors(r, kats):
    for c in kats["get_label_region"]["cases"]:
        assert r.get_label_region(c["labels"]) == [tuple(x) for x in c["expected"]]
    # many regions: the binding grows its buffer transparently
    lab = [0, 1] * 300
    assert r.get_label_region(lab) == [(i, i + 1) for i in range(1, 600, 2)]


def test_smooth_label_region_and_predict_class(r):
    lab = [0] * 50 + [1] * 40 + [0] * 30 + [1] * 5 + [0] * 40
    assert r.smooth_label_region(lab, 21, 13, 20) == [(50, 90)]
    p = r.Predict(lab, "A" * len(lab), "read1", False)
    assert p.prediction_region() == [(50, 90), (120, 125)]
    assert p.smooth_and_select_intervals(21, 13, 20) == [(50, 90)]
    assert p.smooth_label(1) == lab and p.seq_len() == len(lab)
    # Predict.show_info (smooth/predict.rs:127-146): Rust's {:?} of the two interval lists + the read highlighted twice
    info = p.show_info(p.smooth_and_select_intervals(21, 13, 20), 60)
    head = info.split("\n")
    assert head[:3] == ["id: read1", "prediction: [50..90, 120..125]", "smooth_intervals: [(50, 90)]"]
    plain = info.replace("\x1b[1;35m", "").replace("\x1b[0m", "")
    assert plain.split("\n\n")[1].replace("\n", "") == "A" * len(lab) and info.count("\x1b[1;35m") == 3
    assert r.smooth_label_region([0, 1] * 200, 1, 1, 20) == []  # more than `approved` intervals -> []
    with pytest.raises(Exception, match="binary"):
        r.majority_voting([0, 2, 1], 3)


def test_summary_predict_vectors(r, kats):
    for c in kats["summary_predict"]["cases"]:
        p, l = r.summary_predict(c["predictions"], c["labels"], c["ignore"])
        assert p == c["expected_predictions"] and l == c["expected_labels"]


def test_encode_qual_and_normalize_seq(r):
    assert r.encode_qual("!+5?I", 33) == [0, 10, 20, 30, 40]
    assert r.encode_qual("", 33) == []
    assert r.normalize_seq("ACGTNacgtnUu.~- \t\r\nXRY", False) == "ACGTNACGTNTT---NNN"
    assert r.normalize_seq("RYKMbdhv", True) == "RYKMBDHV"
    assert r.seq_to_kmers("ATCGT", 2, True) == ["AT", "TC", "CG", "GT"]
    assert r.seq_to_kmers("atcgt", 2, False) == ["AT", "CG", "T"]


def test_encode_fq_path_to_tensor_like_reference_test(r, golden_dir):
    """tests/test_rust.py::test_encode_fqs_to_tensor (k=3, bases "ACGTN", offset 33, vectorized target)."""
    inputs, target, qual, kmer2idx = r.encode_fq_path_to_tensor(os.path.join(golden_dir, "one_record.fq"), 3, "ACGTN", 33, True)
    assert inputs.shape == (1, 2, 1347) and target.shape == (1, 1, 1347) and qual.shape == (1, 1349)
    assert inputs.dtype == target.dtype == qual.dtype == np.int32
    assert len(kmer2idx) == 125 and kmer2idx["AAA"] == 0 and kmer2idx["NNN"] == 124 and kmer2idx["AAC"] == 1
    assert target[0, 0].sum() == 64 and np.all(target[0, 0, 462:526] == 1)
    ins, tg, q, _ = r.encode_fq_paths_to_tensor([os.path.join(golden_dir, "twenty_five_records.fq")], 3, "ATCGN", 33, True, False)
    assert ins.shape == (25, 2, 4741) and tg.shape == (25, 1, 4741) and q.shape == (25, 4743)
    # records without a target annotation are dropped by the encoder (to_kmer_target_region rejects [0, 0))
    with pytest.raises(Exception):
        r.encode_fq_path_to_tensor(os.path.join(golden_dir, "simple.fq"), 3, "ACGTN", 33, False)
