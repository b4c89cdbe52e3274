"""CPU tests of the native chop stage (`dc_chop_write_chunk`, include/deepchopper_io.h) against the oracle's
`chop_read` (process_chunk, src/bin/predict.rs:130-192; split.rs:138-226): gating, record splitting, naming,
`--ct`, `--ocq`, descriptions of pass-through records.  No GPU: the per-read intervals are computed by the
oracle and handed over in the layout the smoothing kernel produces."""
import gzip

import numpy as np
import pytest

from deepchopper_amd import _iolib
from deepchopper_amd.chop import chop_params
from deepchopper_amd.fastq import BgzfWriter, FastqReader
from deepchopper_amd.predict import SmoothParams
from oracle import dc_oracle as o


def _case_reads(rng):
    """(header, seq, labels): labels are the per-base predictions the intervals are derived from."""
    def rnd(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))

    def lab(n, runs):
        a = np.zeros(n, np.int8)
        for s, e in runs:
            a[s:e] = 1
        return a

    cases = [
        (b"short ch=1", rnd(100), lab(100, [(10, 60)])),                      # < 150: passes through, keeps description
        (b"clean runid=ab ch=2", rnd(600), lab(600, [])),                      # no interval
        (b"internal ch=3 start_time=x", rnd(800), lab(800, [(300, 360)])),     # two kept segments -> |I
        (b"terminal3", rnd(800), lab(800, [(740, 800)])),                      # one kept segment -> |T
        (b"terminal5 d", rnd(800), lab(800, [(0, 50)])),                       # index-0 quirk: run starts at 1, [0,1) kept then filtered
        (b"many", rnd(900), lab(900, [(100 + 150 * k, 130 + 150 * k) for k in range(5)])),   # > 4 intervals
        (b"toomany x", rnd(3000), lab(3000, [(50 + 100 * k, 70 + 100 * k) for k in range(25)])),  # > approved 20
        (b"tiny", rnd(500), lab(500, [(200, 210)])),                           # run shorter than min interval 13
        (b"lower desc", b"acgtnNuUxX.-" * 50, lab(600, [(100, 180)])),         # letters come from the prediction's sequence
        (b"close", rnd(700), lab(700, [(200, 260), (270, 330)])),              # 10-base middle segment is filtered
        (b"tailone", rnd(400), lab(400, [(300, 399)])),                        # tail [399, 399) is empty: nothing after
        (b"noise", rnd(1000), lab(1000, [(500, 560)]) | (rng.random(1000) < 0.03).astype(np.int8)),
        (b"four", rnd(1500), lab(1500, [(100, 140), (400, 440), (800, 860), (1200, 1230)])),
    ]
    return cases


def _run_native(tmp_path, cases, sp, chop_type, ocq, pred_len=None, name="out.fq.gz"):
    rng = np.random.default_rng(5)
    headers = b"".join(h for h, _, _ in cases)
    ho = np.zeros(len(cases) + 1, np.int64)
    np.cumsum([len(h) for h, _, _ in cases], out=ho[1:])
    nlen = np.array([len(h.split(b" ")[0]) for h, _, _ in cases], np.int32)
    seq = np.frombuffer(b"".join(s for _, s, _ in cases), np.uint8)
    quals = [bytes(rng.integers(34, 74, len(s)).astype(np.uint8)) for _, s, _ in cases]
    qual = np.frombuffer(b"".join(quals), np.uint8)
    offs = np.zeros(len(cases) + 1, np.int64)
    np.cumsum([len(s) for _, s, _ in cases], out=offs[1:])
    max_iv = max(sp.approved, 1)
    iv = np.zeros((len(cases), max_iv, 2), np.int32)
    niv = np.zeros(len(cases), np.int32)
    keep = np.zeros((len(cases), max_iv + 1, 2), np.int32)
    nkeep = np.zeros(len(cases), np.int32)
    plen = np.array([len(l) for _, _, l in cases], np.int32) if pred_len is None else np.asarray(pred_len, np.int32)
    for r, (_, s, labels) in enumerate(cases):
        ivs = o.smooth_label_region(labels[: plen[r]], sp.window, sp.min_interval, sp.approved)
        kp = o.generate_unmaped_intervals(ivs, int(plen[r]))
        niv[r], nkeep[r] = len(ivs), len(kp)
        if ivs:
            iv[r, : len(ivs)] = ivs
        if kp:
            keep[r, : len(kp)] = kp
    path = str(tmp_path / name)
    per_read = np.zeros(len(cases), np.int32)
    with BgzfWriter(path, threads=2) as w:
        total = w.chop_write_chunk(chop_params(sp, chop_type, ocq), len(cases), np.frombuffer(headers, np.uint8), ho, nlen,
                                   seq, qual, offs, plen, niv, iv, nkeep, keep, per_read)
    assert total == int(per_read.sum())
    return path, quals, plen, per_read


def _expected(cases, quals, plen, sp, chop_type, ocq):
    exp = []
    for (h, s, labels), q, pl in zip(cases, quals, plen):
        rid = h.split(b" ")[0]
        pseq = o.id_list2seq(o.tokenize_chars(o.normalize_seq(s.upper()))[:-1][:pl])
        p = o.Predict(np.asarray(labels[:pl], np.int8), pseq, rid.decode(), pl != len(s))
        exp += o.chop_read(p, rid, s, q, sp.window, sp.min_interval, sp.approved, sp.max_process_intervals,
                           sp.min_read_length_after_chop, chop_type, ocq, fq_header=h)
    return exp


@pytest.mark.parametrize("chop_type", ["all", "terminal", "internal"])
@pytest.mark.parametrize("ocq", [False, True])
def test_native_chop_equals_oracle(tmp_path, chop_type, ocq):
    sp = SmoothParams()
    cases = _case_reads(np.random.default_rng(1))
    path, quals, plen, per_read = _run_native(tmp_path, cases, sp, chop_type, ocq)
    exp = _expected(cases, quals, plen, sp, chop_type, ocq)
    text = gzip.open(path, "rb").read()
    assert text == b"".join(b"@" + i + b"\n" + s + b"\n+\n" + q + b"\n" for i, s, q in exp)
    if not ocq and chop_type == "all":
        ids = [i for i, _, _ in exp]
        assert b"short ch=1" in ids and b"clean runid=ab ch=2" in ids                    # descriptions kept
        assert b"internal|0:300|I" in ids and b"internal|360:799|I" in ids               # n-1 tail quirk
        assert b"terminal3|0:740|T" in ids
        assert b"terminal5|50:799|I" in ids and not any(i.startswith(b"terminal5|0:1") for i in ids)
        assert b"many" in ids and b"toomany x" in ids and b"tiny" in ids                 # passed through
        assert b"close|0:200|I" in ids and b"close|330:699|I" in ids and not any(i.startswith(b"close|260") for i in ids)
        assert sum(i.startswith(b"four|") for i in ids) == 5
        low = [s for i, s, _ in exp if i.startswith(b"lower|")]
        assert low and all(set(s) <= set(b"ACGTN") for s in low)                         # Predict.seq letters
    if ocq:
        assert b"internal|300:360" in [i for i, _, _ in exp]
    # the reader of this repo reads the output back record for record
    with FastqReader(path, full_header=True) as rd:
        got = [h for c in rd.chunks() for h in [c._names.tobytes()]]
    assert b"".join(got) == b"".join(i for i, _, _ in exp)
    assert int(per_read.sum()) == len(exp)


def test_native_chop_truncated_prediction_passes_through(tmp_path):
    """predict.rs:160-164: a prediction shorter than the read (the tokenizer truncated it) is not processed."""
    sp = SmoothParams()
    cases = _case_reads(np.random.default_rng(2))[2:4]
    plen = [700, 800]  # first read's prediction covers 700 of 800 bases
    path, quals, plen, per_read = _run_native(tmp_path, cases, sp, "all", False, pred_len=plen)
    exp = _expected(cases, quals, plen, sp, "all", False)
    assert exp[0][0] == b"internal ch=3 start_time=x" and per_read.tolist() == [1, 1]
    assert gzip.open(path, "rb").read() == b"".join(b"@" + i + b"\n" + s + b"\n+\n" + q + b"\n" for i, s, q in exp)


def test_native_chop_rejects_bad_arguments(tmp_path):
    sp = SmoothParams()
    with pytest.raises(ValueError, match="Invalid chop type"):
        chop_params(sp, "both")
    cases = _case_reads(np.random.default_rng(3))[2:3]
    p = chop_params(sp)
    h = np.frombuffer(cases[0][0], np.uint8)
    ho = np.array([0, len(h)], np.int64)
    s = np.frombuffer(cases[0][1], np.uint8)
    offs = np.array([0, len(s)], np.int64)
    iv = np.zeros((1, 20, 2), np.int32)
    keep = np.zeros((1, 21, 2), np.int32)
    keep[0, 0] = (0, 9000)  # segment beyond the read
    with BgzfWriter(str(tmp_path / "x.gz")) as w, pytest.raises(ValueError, match="interval is invalid"):
        w.chop_write_chunk(p, 1, h, ho, None, s, s, offs, np.array([len(s)], np.int32), np.array([1], np.int32), iv,
                           np.array([1], np.int32), keep)


def test_full_header_reader(tmp_path):
    fq = tmp_path / "d.fq"
    fq.write_bytes(b"@r1 runid=1 ch=2\nACGT\n+\nIIII\n@r2\nAC\n+r2\nII\n@r3\tx y\r\nA\r\n+\r\nI\r\n")
    with FastqReader(str(fq), full_header=True) as rd:
        c = rd.read_chunk()
        assert c.names() == ["r1", "r2", "r3"] and c._name_len.tolist() == [2, 2, 2]
        assert c._names.tobytes() == b"r1 runid=1 ch=2r2r3\tx y"
    with FastqReader(str(fq)) as rd:
        c = rd.read_chunk()
        assert c.names() == ["r1", "r2", "r3"] and c._name_len is None
    assert "dc_chop_write_chunk" in _iolib.header_symbols() and "dc_fq_read_chunk_full" in _iolib.header_symbols()
