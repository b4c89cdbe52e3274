"""GPU end-to-end tests: FASTQ -> tokenise -> forward -> smooth/intervals through the product API,
the reference's `.pt` wire format, and the `predict` command line."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(hip_lib):
    import torch

    from deepchopper_amd import _lib
    from deepchopper_amd.model import DeepChopper
    from deepchopper_amd.predict import PredictPipeline
    from deepchopper_amd.weights import synthetic_state_dict

    assert _lib.device_count() >= 1
    sd = synthetic_state_dict(0)
    pipe = PredictPipeline(DeepChopper.from_state_dict(sd))
    return torch, sd, pipe


def _oracle_reads(sd, ids, seqs, quals, batch=None):
    """Oracle predict: `batch=None` -> every read alone (zero padding); else reference batches."""
    from oracle import dc_oracle as o
    from oracle import forward_ref as fr

    feats = []
    for i, s, q in zip(ids, seqs, quals):
        rec = o.parse_fastq_record(i, s.decode(), q.decode())
        feats.append(o.tokenize_and_align_labels_and_quals_ids(rec["id"], rec["seq"], rec["qual"], rec["target"]))
    out = {}
    step = batch or 1
    for b0 in range(0, len(feats), step):
        bt = o.collate(feats[b0: b0 + step])
        logits = fr.forward(sd, bt["input_ids"], bt["input_quals"]).numpy()
        for rid, p in o.load_predicts_from_batch({"prediction": logits, "target": bt["labels"], "seq": bt["input_ids"],
                                                  "id": bt["id"]}).items():
            out[rid] = (p, logits)
    return out


def _margin_ok(pred, ref_pred):
    # labels may only differ where the oracle's own logit margin is within the fp32 noise floor
    return np.mean(pred != ref_pred) < 2e-3


def _check_smoothing(o, r, window=21, min_interval=13, approved=20):
    """S1-S5 of one pipeline result against the oracle run on the same (GPU) labels."""
    labels = np.asarray(r["prediction"], np.int8)
    assert np.array_equal(r["smooth"], o.majority_voting(labels, window))
    ivs = o.smooth_label_region(labels, window, min_interval, approved)
    assert r["intervals"] == ivs
    assert r["keep"] == o.generate_unmaped_intervals(ivs, len(labels))


def test_mixed_length_native_equals_per_read_oracle(env):
    torch, sd, pipe = env
    from deepchopper_amd import synth

    seqs, quals, ids = synth.synth_reads_mixed(24, 100, 1500, seed=42)
    out, res = pipe.predict_reads(ids, seqs, quals)
    ref = _oracle_reads(sd, ids, seqs, quals)
    from oracle import dc_oracle as o

    for r in out:
        p, _ = ref[r["id"]]
        assert len(r["prediction"]) == len(p.prediction)
        assert _margin_ok(r["prediction"], p.prediction)
        # the integer half is checked UNCONDITIONALLY: oracle smoothing / intervals / kept segments applied to
        # the GPU's own labels (near-tie logits may flip a label; everything after the argmax must still agree)
        _check_smoothing(o, r)


def test_reference_padding_equals_batched_oracle(env):
    torch, sd, pipe = env
    from deepchopper_amd import synth

    seqs, quals, ids = synth.synth_reads_mixed(12, 150, 700, seed=7)
    out, res = pipe.predict_reads(ids, seqs, quals, reference_padding=True)
    ref = _oracle_reads(sd, ids, seqs, quals, batch=12)
    got = res["logits"].cpu().numpy()
    _, ref_logits = ref[ids[0]]
    assert np.abs(got - ref_logits).max() < 2e-4
    from oracle import dc_oracle as o

    for r in out:
        p, _ = ref[r["id"]]
        assert _margin_ok(r["prediction"], p.prediction)
        _check_smoothing(o, r)


def test_pt_wire_format_roundtrip(env, tmp_path):
    torch, sd, pipe = env
    from deepchopper_amd import rs_api, synth
    from deepchopper_amd.predict import write_pt_batch
    from oracle import dc_oracle as o

    seq, qual, ids = synth.synth_reads_fixed(12, 400, seed=42)
    seqs, quals = [s.tobytes() for s in seq], [q.tobytes() for q in qual]
    out, res = pipe.predict_reads(ids, seqs, quals, reference_padding=True)
    path = str(tmp_path / "0" / "0_0.pt")
    write_pt_batch(path, res, ids)
    d = torch.load(path, weights_only=True)
    assert d["prediction"].dtype == torch.float32 and d["prediction"].shape == (12, 401, 2)
    assert d["target"].dtype == torch.int64 and d["seq"].dtype == torch.int64 and d["id"].shape == (12, 256)
    assert d["qual"].dtype == torch.float32 and d["id"].dtype == torch.int64
    mine = rs_api.load_predicts_from_batch_pt(path)                         # GPU argmax + mask
    theirs = o.load_predicts_from_batch({k: v.numpy() for k, v in d.items()})  # oracle of the Rust loader
    assert set(mine) == set(theirs) == set(ids)
    for r in out:
        assert mine[r["id"]].prediction == theirs[r["id"]].prediction.tolist() == r["prediction"].tolist()
        assert mine[r["id"]].seq == theirs[r["id"]].seq == seqs[ids.index(r["id"])].decode()
        assert mine[r["id"]].smooth_and_select_intervals(21, 13, 20) == r["intervals"]


def test_cli_predict(env, tmp_path):
    torch, sd, pipe = env
    from deepchopper_amd import synth
    from deepchopper_amd.weights import save_state_dict

    ckpt = tmp_path / "ckpt" / "model.safetensors"
    save_state_dict(sd, str(ckpt))
    seq, qual, ids = synth.synth_reads_fixed(30, 300, seed=42)
    fq = tmp_path / "reads.fq"
    fq.write_bytes(synth.to_fastq(seq, qual, ids))
    out = tmp_path / "predictions"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "deepchopper_amd.cli", "predict", str(fq), "-g", "1", "-o", str(out), "-b", "12",
           "-m", str(ckpt.parent)]
    proc = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert sorted(os.listdir(out / "0")) == ["0_0.pt", "0_1.pt", "0_2.pt"]  # reference batching: 12 + 12 + 6
    d = torch.load(out / "0" / "0_2.pt", weights_only=True)
    assert d["prediction"].shape[0] == 6
    out2 = tmp_path / "native"
    cmd2 = cmd[:7] + ["-o", str(out2)] + cmd[9:] + ["--native", "--max-reads", "16"]
    proc = subprocess.run(cmd2, cwd=root, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert sorted(os.listdir(out2 / "0")) == ["0.intervals.tsv", "0_0.pt", "0_1.pt"]  # token-budget batches: 16 + 14
    lines = (out2 / "0" / "0.intervals.tsv").read_text().strip().split("\n")
    assert [l.split("\t")[0] for l in lines] == ids
    # no-GPU flag is rejected loudly
    proc = subprocess.run(cmd[:5] + ["-g", "0"], cwd=root, capture_output=True, text=True, timeout=120)
    assert proc.returncode != 0 and "no CPU fallback" in proc.stderr


def test_chop_driver_matches_oracle(env, tmp_path):
    """predict -> .pt -> chop end to end; every output record equals the oracle's `chop_read`
    (gating S4, intervals S3, complement S5, record naming S6)."""
    torch, sd, pipe = env
    import gzip

    from deepchopper_amd import rs_api, synth
    from deepchopper_amd.chop import chop
    from deepchopper_amd.fastq import iter_fastq
    from deepchopper_amd.predict import write_pt_batch
    from oracle import dc_oracle as o

    seq, qual, ids = synth.synth_reads_fixed(36, 600, seed=42)
    short = synth.synth_reads_fixed(4, 100, seed=9, start_index=900)  # < 150 bp: passed through
    seqs = [s.tobytes() for s in seq] + [s.tobytes() for s in short[0]]
    quals = [q.tobytes() for q in qual] + [q.tobytes() for q in short[1]]
    ids = ids + short[2]
    fq = tmp_path / "reads.fq"
    fq.write_bytes(synth.to_fastq(seqs, quals, ids))
    pdir = tmp_path / "pred" / "0"
    for b, b0 in enumerate(range(0, len(ids), 12)):
        sl = slice(b0, b0 + 12)
        _, res = pipe.predict_reads(ids[sl], seqs[sl], quals[sl], reference_padding=True)
        write_pt_batch(str(pdir / f"0_{b}.pt"), res, ids[sl])
    # synthetic weights give noisy labels -> use a permissive interval policy so that reads really get chopped
    from deepchopper_amd.predict import SmoothParams

    sp = SmoothParams(window=21, min_interval=5, approved=40, max_process_intervals=40)
    os.chdir(tmp_path)
    out = chop([str(pdir)], str(fq), sp)
    preds = rs_api.load_predicts_from_batch_pts(str(pdir))
    assert os.path.basename(out).startswith(f"reads.{len(preds)}pd.") and out.endswith("record.chop.fq.gz")
    got = list(iter_fastq(out))
    exp = []
    n_chopped = 0
    for rid, s, q in zip(ids, seqs, quals):
        p = preds[rid]
        recs = o.chop_read(o.Predict(np.asarray(p.prediction, np.int8), p.seq, rid, p.is_truncated), rid.encode(), s, q,
                           21, 5, 40, 40, 20, "all")
        n_chopped += len(recs) != 1 or recs[0][0] != rid.encode()
        exp += [(r[0].decode(), r[1], r[2]) for r in recs]
    assert n_chopped > 5, "the test should exercise real chopping"
    assert got == exp
    assert int(os.path.basename(out).split(".")[2].replace("record", "")) == len(got)


@pytest.mark.parametrize("chunk", [0, 1])
def test_chop_over_the_reference_pt_directories_equals_the_oracle(hip_lib, tmp_path, chunk):
    """`deepchopper chop <dir> reads.fq` over the REFERENCE's own prediction files (tests/golden/pt/chunk{0,1}:
    its tests/data/eval, the files its loader is pinned on, src/smooth/predict.rs:323-330) -- read natively (chunk1's older
    int8 layout included), smoothed on the GPU, split and written by the host library -- against the oracle's chop_read on the
    same 36 reads, byte for byte.  The FASTQ that produced the files is not in the reference's repository: it is synthesised
    from their `seq` / `id` rows (letters of the unmasked positions; a constant quality)."""
    import gzip

    import torch

    from deepchopper_amd import rs_api
    from deepchopper_amd.chop import chop
    from oracle import dc_oracle as o

    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pt", f"chunk{chunk}")
    records, preds = [], {}
    for name in sorted(os.listdir(d)):
        t = {k: v.numpy() for k, v in torch.load(os.path.join(d, name), map_location="cpu", weights_only=True).items()}
        got = o.load_predicts_from_batch({"prediction": t["prediction"], "target": t["target"].astype(np.int64), "seq": t["seq"].astype(np.int64),
                                          "id": t["id"].astype(np.int64)})
        for rid, p in got.items():
            preds[rid] = p
            records.append((rid.encode(), p.seq.encode(), b"I" * len(p.seq)))
    assert len(records) == 36
    fq = tmp_path / "reads.fq"
    fq.write_bytes(b"".join(b"@" + i + b" ch=7\n" + s + b"\n+\n" + q + b"\n" for i, s, q in records))
    store = rs_api.LabelStore.load(d, threads=2)
    assert store.fallback_files == [] and len(store) == 36
    out = chop([d], str(fq), output_prefix=str(tmp_path / "out"), threads=2)
    want = []
    n_chopped = 0
    for rid, s, q in records:
        recs = o.chop_read(preds[rid.decode()], rid, s, q, fq_header=rid + b" ch=7")
        n_chopped += len(recs) != 1 or recs[0][0] != rid + b" ch=7"
        want += [b"@" + i + b"\n" + rs + b"\n+\n" + rq + b"\n" for i, rs, rq in recs]
    assert gzip.open(out, "rb").read() == b"".join(want)
    assert os.path.basename(out) == f"out.36pd.{len(want)}record.chop.fq.gz"
    assert n_chopped > 0, "the reference's real predictions must chop something"


def test_native_chop_equals_the_per_record_form(env, tmp_path, monkeypatch):
    """`chop` (native `.pt` loader -> one GPU smoothing launch per chunk -> dc_chop_write_chunk) against its per-record form
    (DC_CHOP_PYTHON=1: `chop_chunk`, the line-by-line restatement of process_chunk, src/bin/predict.rs:130-192) on the same
    files, byte for byte (decompressed): `.pt` files in another order than the FASTQ, reads without a prediction (skipped,
    predict.rs:139-142) in the middle of a chunk, short reads (passed through), small chunks, --ocq and every --ct."""
    import gzip

    torch, sd, pipe = env
    from deepchopper_amd import synth
    from deepchopper_amd.chop import chop
    from deepchopper_amd.predict import PtWriter, SmoothParams

    seqs, quals, ids = synth.synth_reads_mixed(90, 100, 1800, seed=11)
    long_s, long_q, long_i = synth.synth_reads_fixed(1, 33500, seed=3, start_index=7000)  # longer than the model's context: its
    seqs[5], quals[5], ids[5] = long_s[0].tobytes(), long_q[0].tobytes(), long_i[0]      # prediction is truncated -> passes through
    fq = tmp_path / "reads.fq"
    fq.write_bytes(synth.to_fastq(seqs, quals, ids))
    pdir = tmp_path / "pred" / "0"
    order = list(range(0, 90, 12))[::-1]  # batches written last to first; batch 3 (reads 36..47) has no file at all
    with PtWriter(threads=2) as w:
        for b, b0 in enumerate(order):
            if b0 == 36:
                continue
            sl = slice(b0, b0 + 12)
            _, res = pipe.predict_reads(ids[sl], seqs[sl], quals[sl], reference_padding=True)
            w.submit(str(pdir), 0, b, res, ids[sl])
    sp = SmoothParams(window=21, min_interval=5, approved=40, max_process_intervals=40)
    for chop_type, ocq in (("all", False), ("terminal", False), ("internal", False), ("all", True)):
        outs = {}
        for form in ("native", "python"):
            monkeypatch.setenv("DC_CHOP_PYTHON", "1" if form == "python" else "0")
            out = chop([str(tmp_path / "pred")], str(fq), sp, chop_type, output_prefix=str(tmp_path / f"{form}_{chop_type}_{int(ocq)}"),
                       chunk_size=17, output_chopped_seqs=ocq)
            outs[form] = (os.path.basename(out).split(".", 1)[1], gzip.open(out, "rb").read())
        assert outs["native"] == outs["python"], (chop_type, ocq)
        assert b"@" + ids[5].encode() + b"\n" + seqs[5] + b"\n" in outs["native"][1]  # the truncated read: as read
        assert outs["native"][0].startswith("78pd.") and outs["native"][1].count(b"\n") >= 4 * 40  # (78 of the 90 reads have a prediction)


def test_predict_stream_matches_predict_reads(env, tmp_path):
    """FASTQ file -> native reader chunks -> predict_stream == predict_reads on the same reads (labels,
    smoothed labels, intervals, kept segments), for gzip input and ragged chunk sizes."""
    torch, sd, pipe = env
    import gzip

    from deepchopper_amd import synth
    from deepchopper_amd.fastq import FastqReader

    seqs, quals, ids = synth.synth_reads_mixed(90, 100, 3000, seed=11)
    fq = tmp_path / "mixed.fq.gz"
    with gzip.open(fq, "wb") as fh:
        fh.write(synth.to_fastq(seqs, quals, ids))
    ref, _ = pipe.predict_reads(ids, seqs, quals)
    got = []
    with FastqReader(str(fq)) as rd:
        for res in pipe.predict_stream(rd.chunks(max_reads=16, cap_bases=30_000)):
            o = res["offsets"]
            assert len(res["names"]) <= 16 and int(res["lengths"].sum()) <= 30_000
            for r, name in enumerate(res["names"]):
                got.append({"id": name, "prediction": res["pred"][o[r]: o[r + 1]], "smooth": res["smooth"][o[r]: o[r + 1]],
                            "intervals": [tuple(x) for x in res["intervals"][r, : res["n_intervals"][r]].tolist()],
                            "keep": [tuple(x) for x in res["keep"][r, : res["n_keep"][r]].tolist()]})
    assert [g["id"] for g in got] == ids
    for g, e in zip(got, ref):
        assert np.array_equal(g["prediction"], e["prediction"]) and np.array_equal(g["smooth"], e["smooth"])
        assert g["intervals"] == e["intervals"] and g["keep"] == e["keep"]
    # a parse error in the reader thread surfaces in the consumer
    bad = tmp_path / "bad.fq"
    bad.write_bytes(synth.to_fastq(seqs[:3], quals[:3], ids[:3]) + b"@x\nACGT\n+\nII\n")
    with FastqReader(str(bad)) as rd, pytest.raises(ValueError, match="length mismatch"):
        list(pipe.predict_stream(rd.chunks(max_reads=2)))


def test_cli_two_ranks_shard_the_file(env, tmp_path):
    """Two CLI processes with RANK=0/1 (one after the other on this one GPU): byte-range shards of the
    same FASTQ; the union of their outputs equals the single-rank run (no exchange step)."""
    torch, sd, pipe = env
    from deepchopper_amd import synth
    from deepchopper_amd.weights import save_state_dict

    ckpt = tmp_path / "ckpt" / "model.safetensors"
    save_state_dict(sd, str(ckpt))
    seqs, quals, ids = synth.synth_reads_mixed(41, 150, 900, seed=12)
    fq = tmp_path / "reads.fq"
    fq.write_bytes(synth.to_fastq(seqs, quals, ids))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(out, rank, world):
        envv = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        cmd = [sys.executable, "-m", "deepchopper_amd.cli", "predict", str(fq), "-g", str(world), "-o", str(out), "-b", "12",
               "-m", str(ckpt.parent), "--native", "--no-pt", "--max-reads", "8"]
        proc = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300, env=envv)
        assert proc.returncode == 0, proc.stderr[-2000:]
        return (out / "0" / f"{rank}.intervals.tsv").read_text()

    single = run(tmp_path / "w1", 0, 1)
    two = run(tmp_path / "w2", 0, 2) + run(tmp_path / "w2", 1, 2)
    assert [l.split("\t")[0] for l in single.strip().split("\n")] == ids
    assert two == single
    assert sorted(os.listdir(tmp_path / "w2" / "0")) == ["0.intervals.tsv", "1.intervals.tsv"]


def test_single_read_api_matches_oracle(env):
    """ui/main.py:53-106 `predict(text)`: intervals JSON + highlighted segments for one FASTQ record."""
    torch, sd, pipe = env
    from deepchopper_amd import synth, ui
    from oracle import dc_oracle as o

    seq, qual, ids = synth.synth_reads_fixed(1, 900, seed=7)
    s, q = seq[0].tobytes().decode(), qual[0].tobytes().decode()
    text = f"@{ids[0]}\n{s}\n+\n{q}\n"
    # permissive policy: synthetic weights give many short intervals
    js, hl = ui.predict(text, 21, 5, 200, 200, model=pipe.model)
    ref, _ = pipe.predict_reads([ids[0]], [s.encode()], [q.encode()])
    labels = ref[0]["prediction"]
    exp = o.smooth_label_region(labels, 21, 5, 200)
    assert [(d["start"], d["end"]) for d in js] == exp and len(exp) > 0
    assert all(tag in ("ada", None) for _, tag in hl)
    kept = o.generate_unmaped_intervals(exp, len(s))
    assert [t for t, tag in hl if tag == "ada"] == [s[a:b] for a, b in exp]
    assert [t for t, tag in hl if tag is None] == [s[a:b] for a, b in kept]
    assert ui.predict(text, 21, 13, 20, 0, model=pipe.model) == ([], [])  # more intervals than max_process_intervals
    with pytest.raises(ValueError):
        list(ui.parse_fq_record("@r\nACGT\n+\n"))


# ------------------------------------------------------------------------------------------------------
# fused predict -> chop (no .pt in between), BASELINE config 5's flow
# ------------------------------------------------------------------------------------------------------
def _designed_reads(rng):
    """Reads whose adapters are runs of N: with `token_gated_state_dict` the per-base label is 1 exactly on N
    (logit margin ~1), so the expected records do not depend on near-tie logits."""
    def rnd(n, runs=()):
        a = bytearray(rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes())
        for s, e in runs:
            a[s:e] = b"N" * (e - s)
        return bytes(a)

    reads = [
        (b"short ch=1", rnd(100, [(10, 60)])),                                # < 150 bp: passes through with its description
        (b"clean runid=ab ch=2", rnd(600)),                                   # no interval
        (b"internal ch=3 start_time=x", rnd(800, [(300, 360)])),              # |I x 2
        (b"terminal3", rnd(800, [(740, 800)])),                               # |T
        (b"terminal5 d", rnd(800, [(0, 50)])),                                # index-0 quirk
        (b"many", rnd(900, [(100 + 150 * k, 130 + 150 * k) for k in range(5)])),        # > 4 intervals: passes through
        (b"toomany x", rnd(990, [(20 + 38 * k, 35 + 38 * k) for k in range(25)])),     # > 20 approved: passes through
        (b"tiny", rnd(500, [(200, 210)])),                                    # shorter than min interval
        (b"truncated ch=9", rnd(1200, [(300, 360)])),                         # >= max_length (1000): prediction truncated
        (b"lower desc", b"acgtuU.-" * 25 + b"n" * 40 + b"acgt" * 90),            # letters come from the prediction
        (b"close", rnd(700, [(200, 260), (270, 330)])),                       # 10-base middle segment is filtered
        (b"isolatedN", rnd(640, [(100, 101), (300, 301), (420, 480)])),       # lone N's are smoothed away
        (b"four", rnd(990, [(100, 140), (300, 340), (600, 660), (900, 930)])),
    ]
    for i in range(20):
        n = int(rng.integers(150, 990))
        runs = []
        if i % 3 != 2:
            s0 = int(rng.integers(20, n - 100))
            runs.append((s0, s0 + int(rng.integers(15, 70))))
        if i % 5 == 0:
            runs.append((n - 45, n))
        reads.append((b"mix%02d ch=%d" % (i, i), rnd(n, runs)))
    out = []
    for h, s in reads:
        out.append((h, s, bytes(rng.integers(34, 74, len(s)).astype(np.uint8))))
    return out


@pytest.fixture(scope="module")
def gated(hip_lib):
    from deepchopper_amd.model import DeepChopper
    from deepchopper_amd.predict import PredictPipeline
    from deepchopper_amd.weights import token_gated_state_dict

    sd = token_gated_state_dict(0)
    return sd, PredictPipeline(DeepChopper.from_state_dict(sd), max_length=1000)


def _oracle_predicts(sd, reads, max_length):
    from oracle import dc_oracle as o
    from oracle import forward_ref as fr

    preds = []
    for h, s, q in reads:
        rid = h.split(b" ")[0].decode()
        rec = o.parse_fastq_record(rid, s.decode(), q.decode())
        f = o.tokenize_and_align_labels_and_quals_ids(rid, rec["seq"], rec["qual"], rec["target"], max_length=max_length)
        bt = o.collate([f])
        logits = fr.forward(sd, bt["input_ids"], bt["input_quals"]).numpy()
        margin = np.abs(logits[0, :-1, 1] - logits[0, :-1, 0]).min()
        assert margin > 0.3, "the designed gate must keep every logit pair far apart"
        preds.append(o.load_predicts_from_batch({"prediction": logits, "target": bt["labels"], "seq": bt["input_ids"],
                                                 "id": bt["id"]})[rid])
    return preds


@pytest.mark.parametrize("chop_type,ocq", [("all", False), ("terminal", False), ("internal", False), ("all", True)])
def test_fused_predict_chop_matches_oracle_and_pt_route(gated, tmp_path, chop_type, ocq):
    """synthetic mixed-length .fastq.gz -> ONE call -> chopped BGZF.  Every output record
    equals the oracle's `chop_read` on the oracle's own prediction of that read, and the file equals the
    `.pt` -> chop route byte for byte (decompressed)."""
    import gzip

    from deepchopper_amd import rs_api
    from deepchopper_amd.chop import chop, predict_chop
    from deepchopper_amd.fastq import FastqReader
    from deepchopper_amd.predict import write_pt_batch
    from oracle import dc_oracle as o

    sd, pipe = gated
    reads = _designed_reads(np.random.default_rng(3))
    fq = tmp_path / "reads.fastq.gz"
    with gzip.open(fq, "wb") as fh:
        fh.write(b"".join(b"@" + h + b"\n" + s + b"\n+\n" + q + b"\n" for h, s, q in reads))
    os.chdir(tmp_path)
    final, n_reads, n_out = predict_chop(str(fq), pipe, chop_type=chop_type, output_chopped_seqs=ocq, max_reads=7,
                                         max_tokens=4000)
    assert n_reads == len(reads)
    assert os.path.basename(final) == f"reads.fastq.{n_reads}pd.{n_out}record.chop.fq.gz"
    got = gzip.open(final, "rb").read()
    # oracle: CPU forward -> labels -> chop_read, read by read
    sp = pipe.smooth
    exp = []
    for (h, s, q), p in zip(reads, _oracle_predicts(sd, reads, 1000)):
        exp += o.chop_read(p, h.split(b" ")[0], s, q, sp.window, sp.min_interval, sp.approved, sp.max_process_intervals,
                           sp.min_read_length_after_chop, chop_type, ocq, fq_header=h)
    assert got == b"".join(b"@" + i + b"\n" + s + b"\n+\n" + q + b"\n" for i, s, q in exp)
    assert n_out == len(exp)
    ids = [i for i, _, _ in exp]
    if chop_type == "all" and not ocq:
        for want in (b"short ch=1", b"clean runid=ab ch=2", b"internal|0:300|I", b"internal|360:799|I", b"terminal3|0:740|T",
                     b"terminal5|50:799|I", b"many", b"toomany x", b"tiny", b"truncated ch=9", b"close|0:200|I",
                     b"close|330:699|I", b"isolatedN|0:420|I", b"lower|0:200|I", b"four|0:100|I"):
            assert want in ids, want
        low = [s for i, s, _ in exp if i == b"lower|0:200|I"][0]
        assert low == b"ACGTTTNN" * 25
    if chop_type == "terminal":
        assert b"terminal3|0:740|T" in ids and b"internal" in ids       # internal chops pass as the bare record
    if chop_type == "internal":
        assert b"internal|0:300|I" in ids and b"terminal3" in ids
    if ocq:
        assert b"internal|300:360" in ids and b"four|900:930" in ids
    # the reference's route: predict -> 28 B/token .pt files -> chop; same bytes
    pdir = tmp_path / f"pred_{chop_type}_{int(ocq)}" / "0"
    with FastqReader(str(fq)) as rd:
        for b, res in enumerate(pipe.predict_stream(rd.chunks(6, 5000), reference_padding=True, ids_i64=True)):
            write_pt_batch(str(pdir / f"0_{b}.pt"), res["device"], res["names"])
    via_pt = chop([str(pdir)], str(fq), sp, chop_type, output_prefix=str(tmp_path / f"pt_{chop_type}_{int(ocq)}"),
                  output_chopped_seqs=ocq)
    assert gzip.open(via_pt, "rb").read() == got
    assert len(rs_api.load_predicts_from_batch_pts(str(pdir))) == n_reads


def test_fused_predict_chop_random_weights_vs_oracle_on_gpu_labels(env, tmp_path):
    """Same flow with the ordinary synthetic weights (noisy labels, permissive policy so that reads really get
    chopped): records == oracle.chop_read applied to the GPU's own labels."""
    import gzip

    from deepchopper_amd import synth
    from deepchopper_amd.chop import predict_chop
    from deepchopper_amd.predict import PredictPipeline, SmoothParams
    from oracle import dc_oracle as o

    torch, sd, pipe0 = env
    sp = SmoothParams(window=21, min_interval=5, approved=40, max_process_intervals=40)
    pipe = PredictPipeline(pipe0.model, smooth=sp)
    seqs, quals, ids = synth.synth_reads_mixed(40, 100, 2500, seed=5)
    fq = tmp_path / "noisy.fq"
    fq.write_bytes(synth.to_fastq(seqs, quals, ids))
    os.chdir(tmp_path)
    final, n_reads, n_out = predict_chop(str(fq), pipe, max_reads=16)
    out, _ = pipe.predict_reads(ids, seqs, quals)
    exp, n_chopped = [], 0
    for r, s, q in zip(out, seqs, quals):
        p = o.Predict(np.asarray(r["prediction"], np.int8), o.id_list2seq(o.tokenize_chars(o.normalize_seq(s.upper()))[:-1]),
                      r["id"], False)
        recs = o.chop_read(p, r["id"].encode(), s, q, 21, 5, 40, 40, 20, "all")
        n_chopped += len(recs) != 1 or recs[0][0] != r["id"].encode()
        exp += recs
    assert n_chopped > 5
    assert gzip.open(final, "rb").read() == b"".join(b"@" + i + b"\n" + s + b"\n+\n" + q + b"\n" for i, s, q in exp)
    assert (n_reads, n_out) == (40, len(exp))


def test_predict_stream_attributes_range_guard_per_batch(hip_lib, tmp_path):
    """with batch k+1 launched before batch k is collected, an overflow of the fp16x3 range
    guard must be charged to the batch that raised it -- and only that batch re-run in bf16x6."""
    from deepchopper_amd.fastq import FastqReader
    from deepchopper_amd.model import DeepChopper
    from deepchopper_amd.predict import PredictPipeline
    from deepchopper_amd.weights import BB, token_gated_state_dict

    sd = token_gated_state_dict(0, backbone_gain=0.0)
    emb = sd[BB + "embeddings.word_embeddings.weight"]
    p1 = BB + "layers.1."
    sd[p1 + "norm2.weight"], sd[p1 + "norm2.bias"] = np.ones(256, np.float32), np.zeros(256, np.float32)
    w = sd[p1 + "mlp.fc1.weight"].copy()
    b = sd[p1 + "mlp.fc1.bias"].copy()
    w[0], b[0] = emb[11] * np.float32(3.0e5 / 256), 0.0   # hidden unit 0 = 3e5 on every 'N' token: gelu(3e5) > 65504
    sd[p1 + "mlp.fc1.weight"], sd[p1 + "mlp.fc1.bias"] = w, b
    pipe = PredictPipeline(DeepChopper.from_state_dict(sd))
    assert pipe.model.net.gemm_mode == "fp16x3"
    rng = np.random.default_rng(8)
    recs = []
    for i in range(40):
        s = bytearray(rng.choice(np.frombuffer(b"ACGT", np.uint8), 300).tobytes())
        if 16 <= i < 24 or i >= 36:  # batches 2 and 4 of five carry N runs
            s[100:140] = b"N" * 40
        recs.append((b"r%02d" % i, bytes(s), bytes(rng.integers(34, 74, 300).astype(np.uint8))))
    fq = tmp_path / "g.fq"
    fq.write_bytes(b"".join(b"@" + h + b"\n" + s + b"\n+\n" + q + b"\n" for h, s, q in recs))
    flags, labels = [], []
    with FastqReader(str(fq)) as rd:
        for res in pipe.predict_stream(rd.chunks(8)):
            flags.append(res["range_guard_rerun"])
            o_ = res["offsets"]
            labels += [res["pred"][o_[r]: o_[r + 1]] for r in range(len(res["names"]))]
    assert flags == [False, False, True, False, True]
    for (h, s, q), lab in zip(recs, labels):
        assert np.array_equal(lab, (np.frombuffer(s, np.uint8) == ord("N")).astype(np.int8)), h
    assert pipe.model.net.gemm_mode == "fp16x3" and pipe.model.net.overflow() == 0
    # the public forward surface consults the guard too
    import torch

    ids = np.full((2, 50), 7, np.int64)
    ids[1, 10:20] = 11
    ids[:, -1] = 1
    qq = np.full((2, 50), 0.1, np.float32)
    out = pipe.model.forward(torch.from_numpy(ids), torch.from_numpy(qq)).cpu().numpy()
    assert np.isfinite(out).all()
    assert np.array_equal(out[1, :-1, 1] > out[1, :-1, 0], ids[1, :-1] == 11)


@pytest.mark.parametrize("n_reads,bp", [(2048, 1000), (1024, 2000)])
def test_bench_shape_batch_properties(env, n_reads, bp):
    """BASELINE configs 2 / 3 at the batch shape bench.py runs (VERDICT r1 next #2c): size-independent
    properties through `run_device_batch` -- finite logits, labels == argmax of the returned logits (first max
    wins), smoothing / intervals / kept segments == the oracle on those labels for every 64th read."""
    from deepchopper_amd import ops, synth
    from oracle import dc_oracle as o

    torch, sd, pipe = env
    seq, qual, ids = synth.synth_reads_fixed(n_reads, bp, seed=42)
    offs = np.arange(n_reads + 1, dtype=np.int64) * bp
    lens = np.full(n_reads, bp, np.int64)
    res = pipe.run_device_batch(ops.to_device(seq.reshape(-1)), ops.to_device(qual.reshape(-1)), ops.to_device(offs), lens)
    torch.cuda.synchronize()
    logits = res["logits"].cpu().numpy()
    assert logits.shape == (n_reads, bp + 1, 2) and np.isfinite(logits).all()
    sm = res["smooth"]
    pred = sm["pred"].cpu().numpy().reshape(n_reads, bp)
    assert np.array_equal(pred, (logits[:, :bp, 1] > logits[:, :bp, 0]).astype(np.int8))
    smooth = sm["smooth"].cpu().numpy().reshape(n_reads, bp)
    iv, niv = sm["intervals"].cpu().numpy(), sm["n_intervals"].cpu().numpy()
    kp, nkp = sm["keep"].cpu().numpy(), sm["n_keep"].cpu().numpy()
    tok = res["tok"]["input_ids"].cpu().numpy()
    assert (tok[:, -1] == 1).all() and set(np.unique(tok[:, :-1]).tolist()) <= {7, 8, 9, 10, 11}
    for r in range(0, n_reads, 64):
        assert np.array_equal(smooth[r], o.majority_voting(pred[r], 21))
        ivs = o.smooth_label_region(pred[r], 21, 13, 20)
        assert [tuple(x) for x in iv[r, : niv[r]].tolist()] == ivs
        assert [tuple(x) for x in kp[r, : nkp[r]].tolist()] == o.generate_unmaped_intervals(ivs, bp)
    assert pipe.model.net.overflow() == 0


@pytest.mark.parametrize("launch", ["self-2", "self-3", "torchrun-2"])
def test_multi_rank_chop_output_is_in_input_order_and_byte_identical_to_one_gpu(gated, tmp_path, launch):
    """`predict -g N --chop` on a plain .fastq.gz (one inflater deals the chunks round-robin) writes
    the SAME FILE, byte for byte, as `-g 1` -- records in input order, no sorting -- whether the command starts its own
    workers or runs under torch.distributed.run (the external launcher used to leave N partial files).
    The ranks share this box's one GPU (DC_SHARE_GPU=1)."""
    import gzip

    from deepchopper_amd.weights import save_state_dict

    sd, _ = gated
    ckpt = tmp_path / "ckpt" / "model.safetensors"
    save_state_dict(sd, str(ckpt))
    reads = _designed_reads(np.random.default_rng(5)) * 3
    reads = [(b"r%03d_" % k + h, s, q) for k, (h, s, q) in enumerate(reads)]
    text = b"".join(b"@" + h + b"\n" + s + b"\n+\n" + q + b"\n" for h, s, q in reads)
    fq = tmp_path / "reads.fastq.gz"
    with gzip.open(fq, "wb") as fh:
        fh.write(text)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["predict", str(fq), "--chop", "-m", str(ckpt.parent), "--max-reads", "7", "--max-tokens", "6000", "-t", "2"]

    def run(cmd, out_prefix, extra_env=None):
        envv = dict(os.environ, DC_SHARE_GPU="1", **(extra_env or {}))
        proc = subprocess.run(cmd + ["--chop-output", str(out_prefix)], cwd=root, capture_output=True, text=True, timeout=600, env=envv)
        assert proc.returncode == 0, proc.stderr[-3000:]
        out = [l for l in proc.stdout.strip().splitlines() if l.endswith(".chop.fq.gz")]
        assert len(out) == 1, proc.stdout
        return out[0]

    one = run([sys.executable, "-m", "deepchopper_amd.cli"] + common + ["-g", "1"], tmp_path / "one")
    kind, n = launch.split("-")
    if kind == "self":
        rep = tmp_path / "report.json"
        many = run([sys.executable, "-m", "deepchopper_amd.cli"] + common + ["-g", n, "--report-json", str(rep)], tmp_path / "many")
        # CPU placement (deepchopper_amd/placement.py): every worker bound to its own slice of the GPU's NUMA node's CPUs,
        # torch's intra-op pool bounded by the slice, the parent (single inflater) on CPUs none of them uses
        import json

        from deepchopper_amd.placement import parse_cpulist

        report = json.load(open(rep))
        sets = [set(parse_cpulist(r["cpuset"])) for r in report["ranks"]]
        assert all(0 < len(c) <= 5 for c in sets), report["cpusets"]
        assert all(not (sets[i] & sets[j]) for i in range(len(sets)) for j in range(i + 1, len(sets))), report["cpusets"]
        assert all(r["torch_threads"] == len(c) for r, c in zip(report["ranks"], sets))
        assert len({r["numa_node"] for r in report["ranks"]}) == 1  # the ranks share this box's one GPU, hence its node
        if report["cpusets"]["parent"]:
            assert not (set(parse_cpulist(report["cpusets"]["parent"])) & set().union(*sets))
    else:
        many = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                    "--master-port", "29617", "-m", "deepchopper_amd.cli"] + common + ["-g", n], tmp_path / "many")
    assert os.path.basename(one).split(".", 1)[1] == os.path.basename(many).split(".", 1)[1]  # same {n}pd.{m}record counts
    a, b = open(one, "rb").read(), open(many, "rb").read()
    assert gzip.decompress(a) == gzip.decompress(b)
    assert a == b
    names = [l[1:].split(b"|")[0].split(b" ")[0] for l in gzip.decompress(b).split(b"\n")[0::4] if l]
    order = [h.split(b" ")[0] for h, _, _ in reads]
    assert [x for i, x in enumerate(names) if i == 0 or names[i - 1] != x] == [x for x in order if x in set(names)]
    left = [f for f in os.listdir(tmp_path) if f.startswith(".deepchopper_temp")]
    assert left == [], left


@pytest.mark.parametrize("launch", ["self", "torchrun"])
def test_bench_two_rank_file_legs_match_one_gpu_bytes(hip_lib, tmp_path, launch):
    """`bench.py --gpus 2` carries file-fed legs beside the resident headline -- c3 (one 2 kb FASTQ in
    byte-range shards), c5 (one .fastq.gz through the node's single inflater; the same reads as BGZF shards) -- with per-rank
    reads/s, CPU seconds per 1 000 reads and one host_bound verdict each; and the N = 2 c5 output IS the file `-g 1` writes:
    byte for byte for the dealt gzip stream, record for record for the BGZF shards.  Ranks share this box's one GPU."""
    import gzip
    import json

    from deepchopper_amd.weights import save_state_dict, token_gated_state_dict

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    keep = tmp_path / "keep"
    envv = dict(os.environ, DC_BENCH_SHARE_GPU="1", DC_BENCH_KEEP_DIR=str(keep), DC_BENCH_TIMEOUT="500")
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "256", "--strict-steps", "0", "--no-profile",
            "--c3-reads-per-rank", "3000", "--c5-reads-per-rank", "5000"]
    if launch == "self":
        cmd = [sys.executable, "bench.py"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", "29641", "bench.py"] + args
    proc = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=560, env=envv)
    assert proc.returncode == 0, proc.stderr[-3000:]
    line = json.loads([l for l in proc.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    legs = line["multi_gpu"]
    assert set(legs) == {"c3_2kb_file_sharded", "c5_gz_single_inflater", "c5_bgzf_sharded"}, legs
    for name, leg in legs.items():
        assert "error" not in leg, (name, leg)
        assert len(leg["per_rank"]) == 2 and all(r["reads"] > 0 and r["reads_per_s"] > 0 for r in leg["per_rank"])
        assert leg["host"]["cpu_budget_cores"] >= 1 and isinstance(leg["host"]["host_bound"], bool)
        assert leg["cpu_s_per_1000_reads"]["worker"] > 0
        assert all(v is True for k, v in leg["check"].items() if isinstance(v, bool)), (name, leg["check"])
    c3 = legs["c3_2kb_file_sharded"]
    assert c3["reads"] == 6000 and [r["reads"] for r in c3["per_rank"]] == [3000, 3000]
    assert c3["check"]["per_base_f1_vs_oracle"] == 1.0 and c3["check"]["intervals_identical"]
    gzl, bgl = legs["c5_gz_single_inflater"], legs["c5_bgzf_sharded"]
    assert gzl["input"].startswith("single inflater") and gzl["cpu_s_per_1000_reads"]["parent"] > 0
    assert gzl["reads"] == bgl["reads"] == 10000 and gzl["check"]["head_bytes_identical"]
    # ... and `-g 1` on the very same input
    tmp = next(keep.iterdir())
    ckpt = tmp_path / "ckpt" / "model.safetensors"
    save_state_dict(token_gated_state_dict(0), str(ckpt))
    one = subprocess.run([sys.executable, "-m", "deepchopper_amd.cli", "predict", str(tmp / "hct116_like.fastq.gz"), "--chop", "-g", "1",
                          "-m", str(ckpt.parent), "-t", "4", "--chop-output", str(tmp_path / "one")], cwd=root, capture_output=True,
                         text=True, timeout=300, env=dict(os.environ))
    assert one.returncode == 0, one.stderr[-3000:]
    one_path = [l for l in one.stdout.strip().splitlines() if l.endswith(".chop.fq.gz")][-1]
    a, b, c = open(one_path, "rb").read(), open(gzl["output"], "rb").read(), open(bgl["output"], "rb").read()
    assert os.path.basename(one_path).split(".", 1)[1] == os.path.basename(gzl["output"]).split(".", 1)[1]
    assert a == b, "the dealt single-inflater output must be the file one GPU writes, byte for byte"
    assert gzip.decompress(a) == gzip.decompress(c), "the BGZF-sharded output must hold the same records in the same order"
    assert gzl["records_out"] == bgl["records_out"]


def test_results_do_not_depend_on_a_handles_history_or_stream(env, tmp_path):
    """Two properties the round-3 "lanes" experiment (several model handles on several HIP streams) exposed:
    (1) a read's logits must not depend on what its handle processed before -- the fp16x3 scale of a channel's long filter
        is taken from the full-length filter and the FFT spectra from all taps of their size class, not from the taps
        materialised so far;
    (2) a handle driven from a NON-BLOCKING side stream must give the same result as on the default stream -- the
        workspace zero-fill used to run on the NULL stream, which such a stream does not wait for: the first forward raced
        with it (and the cached index arrays could stay zero for good, which made side streams look 20 % "faster")."""
    torch, sd, pipe = env
    from deepchopper_amd import ops, synth
    from deepchopper_amd.model import DeepChopper
    from deepchopper_amd.predict import PredictPipeline

    def batch(n, lo, hi, seed):
        seqs, quals, _ = synth.synth_reads_mixed(n, lo, hi, seed=seed)
        sb, offs = ops.pack_reads(seqs)
        qb, _ = ops.pack_reads(quals)
        return ops.to_device(sb), ops.to_device(qb), ops.to_device(offs), np.diff(offs)

    short, long_ = batch(9, 100, 900, 1), batch(3, 2600, 5000, 2)   # the long batch takes the FFT path and more filter taps

    def valid_logits(res, lens):
        lg = res["logits"].cpu().numpy()
        return np.concatenate([lg[r, lg.shape[1] - int(n) - 1:] for r, n in enumerate(lens)])

    fresh = PredictPipeline(DeepChopper.from_state_dict(sd))
    a = valid_logits(fresh.run_device_batch(*short), short[3])           # short reads on a handle that has seen nothing
    seasoned = PredictPipeline(DeepChopper.from_state_dict(sd))
    seasoned.run_device_batch(*long_)                                     # ... and after long reads grew filters / spectra
    b = valid_logits(seasoned.run_device_batch(*short), short[3])
    assert np.array_equal(a, b)
    c = valid_logits(pipe.run_device_batch(*short), short[3])             # the module's handle: many other tests before
    assert np.array_equal(a, c)
    side = torch.cuda.Stream()                                            # non-blocking, as torch creates them
    other = PredictPipeline(DeepChopper.from_state_dict(sd))
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        r1 = other.run_device_batch(*short)                              # FIRST forward of this handle, on the side stream
        r2 = other.run_device_batch(*long_)
    side.synchronize()
    assert np.array_equal(valid_logits(r1, short[3]), a)
    assert np.array_equal(valid_logits(r2, long_[3]), valid_logits(seasoned.run_device_batch(*long_), long_[3]))
    for k in ("pred", "smooth", "intervals", "n_intervals", "keep", "n_keep"):
        assert torch.equal(r1["smooth"][k], fresh.run_device_batch(*short)["smooth"][k]), k


def test_fused_reference_batches_write_the_files_of_separate_launches(env, tmp_path):
    """`deepchopper predict -b 12` (the reference's default: left-padded batches of 12, one `.pt` per batch,
    deepchopper/cli.py:70, models/callbacks.py:12-25) shares GPU launches between reference batches (`fastq.fuse_chunks`): every
    batch keeps ITS OWN left padding -- its pad tokens run through the model as the reference's collator lays them out
    (tokenizer.py:71-87) -- so the files must be those of one launch per batch: same names, shapes, token / label /
    quality / id tensors bit for bit, logits to fp32 rounding (another batch composition reorders nothing inside a row, but the
    long-convolution groups differ), and against the batched oracle within the parity tolerance."""
    torch, sd, pipe = env
    from deepchopper_amd import synth
    from deepchopper_amd.fastq import FastqReader, fuse_chunks
    from deepchopper_amd.predict import write_pt_batch

    seqs, quals, ids = synth.synth_reads_mixed(67, 60, 1400, seed=21)  # batches of 12 with very different paddings; 67 = 5 x 12 + 7
    fq = tmp_path / "reads.fq"
    fq.write_bytes(synth.to_fastq(seqs, quals, ids))

    def run(fused: bool, out):
        n = 0
        with FastqReader(str(fq)) as rd:
            chunks = rd.chunks(12, 1 << 30)
            if fused:
                chunks = fuse_chunks(chunks, max_reads=40, max_tokens=1 << 30)  # 3 + 3 batches per launch
            for res in pipe.predict_stream(chunks, reference_padding=True, ids_i64=True):
                subs = res["sub_sizes"] or [len(res["names"])]
                cols = res["sub_cols"] or [0] * len(subs)
                assert fused == (len(subs) > 1) or len(res["names"]) <= 12
                r0 = 0
                for k, c0 in zip(subs, cols):
                    write_pt_batch(str(out / f"0_{n}.pt"), res["device"], res["names"][r0: r0 + k], rows=slice(r0, r0 + k), col0=c0)
                    n += 1
                    r0 += k
        return n

    a, b = tmp_path / "separate", tmp_path / "fused"
    assert run(False, a) == run(True, b) == 6
    for i in range(6):
        x = torch.load(a / f"0_{i}.pt", weights_only=True)
        y = torch.load(b / f"0_{i}.pt", weights_only=True)
        assert x["prediction"].shape == y["prediction"].shape and x["prediction"].shape[0] == (12 if i < 5 else 7)
        for key in ("target", "seq", "qual", "id"):
            assert torch.equal(x[key], y[key]), (i, key)
        assert float((x["prediction"] - y["prediction"]).abs().max()) < 2e-5
        assert (y["seq"][:, 0] != 4).any()  # cut at the batch's own padding: its longest read starts in column 0
    # ... and the fused files against the oracle run batch by batch (the reference at -b 12)
    ref = _oracle_reads(sd, ids[:12], seqs[:12], quals[:12], batch=12)
    y = torch.load(b / "0_0.pt", weights_only=True)
    assert np.abs(y["prediction"].numpy() - ref[ids[0]][1]).max() < 2e-4


def test_pt_pack_against_numpy_on_random_geometries(env):
    """dc_pt_pack alone: random launches ([n, stride] dense arrays), random files (any rows, any first column, gaps and
    overlaps between files allowed), uint8 and int64 token ids -- every file's payload is prediction f32[B][L][2] | target
    i64[B][L] | seq i64[B][L] | qual f32[B][L] of its slice, bit for bit; bytes between payloads are left alone; bad tables
    (rows outside the launch, unaligned or overflowing offsets) are refused."""
    torch, sd, pipe = env
    from deepchopper_amd import _lib, ops

    rng = np.random.default_rng(12)
    for trial in range(8):
        n, stride = int(rng.integers(1, 40)), int(rng.integers(1, 3000))
        logits = rng.standard_normal((n, stride, 2)).astype(np.float32)
        labels = rng.integers(-100, 2, (n, stride)).astype(np.int8)
        ids = rng.integers(0, 12, (n, stride)).astype(np.int64 if trial % 2 else np.uint8)
        quals = rng.random((n, stride)).astype(np.float32)
        files, off = [], 0
        for _ in range(int(rng.integers(1, 12))):
            r0 = int(rng.integers(0, n))
            B = int(rng.integers(1, n - r0 + 1))
            c0 = int(rng.integers(0, stride))
            off = (off + 63) // 64 * 64 + 64 * int(rng.integers(0, 3))
            files.append((r0, B, c0, off))
            off += B * (stride - c0) * 28
        table = np.array(files, np.int64)
        out = torch.full((off + 64,), 0xAB, dtype=torch.uint8, device="cuda")
        ops.pt_pack(ops.to_device(logits), ops.to_device(labels), ops.to_device(ids), ops.to_device(quals), table, out)
        got = out.cpu().numpy()
        want = np.full(off + 64, 0xAB, np.uint8)
        for r0, B, c0, o in files:
            L = stride - c0
            parts = [logits[r0: r0 + B, c0:].tobytes(), labels[r0: r0 + B, c0:].astype(np.int64).tobytes(),
                     ids[r0: r0 + B, c0:].astype(np.int64).tobytes(), quals[r0: r0 + B, c0:].tobytes()]
            blob = np.frombuffer(b"".join(parts), np.uint8)
            assert blob.size == B * L * 28
            want[o: o + blob.size] = blob
        assert np.array_equal(got, want), trial
    d = [ops.to_device(logits), ops.to_device(labels), ops.to_device(ids), ops.to_device(quals)]
    out = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    for bad in ([[0, n + 1, 0, 0]], [[0, 1, stride, 0]], [[0, 1, 0, 32]], [[0, n, 0, (1 << 20) - 64]], [[-1, 1, 0, 0]]):
        with pytest.raises(_lib.DeepChopperHipError):
            ops.pt_pack(*d, np.array(bad, np.int64), out)


def test_gpu_packed_pt_files_equal_torch_save(env, tmp_path):
    """`predict.PtWriter` on the pipeline's device tensors -- dc_pt_pack lays the storages of every reference batch of a fused
    launch out on the GPU, one asynchronous copy brings them to pinned memory, libdeepchopper_io writes the files with a
    writev() each -- against `write_pt_batches` (torch.save of host slices) on the same launches: same file names; `torch.load`
    gives the same tensors bit for bit (uint8 and int64 token ids on the device alike); the pickle stream is torch.save's byte
    for byte; CRCs check out.  Long and short reads, 31 batches over several launches, several launches in flight."""
    import zipfile

    torch, sd, pipe = env
    from deepchopper_amd import synth
    from deepchopper_amd.fastq import FastqReader, fuse_chunks
    from deepchopper_amd.predict import PtWriter, write_pt_batches

    seqs, quals, ids = synth.synth_reads_mixed(367, 60, 5000, seed=5)
    fq = tmp_path / "reads.fq"
    fq.write_bytes(synth.to_fastq(seqs, quals, ids))
    for ids_i64 in (False, True):
        a, b = tmp_path / f"torch{int(ids_i64)}", tmp_path / f"packed{int(ids_i64)}"
        n = 0
        with FastqReader(str(fq)) as rd, PtWriter(threads=3, max_inflight=2) as w:
            chunks = fuse_chunks(rd.chunks(12, 1 << 30), max_reads=96, max_tokens=1 << 30)
            for res in pipe.predict_stream(chunks, reference_padding=True, ids_i64=ids_i64):
                k = w.submit(str(b), 0, n, res["device"], res["names"], ready_event=res["ready_event"], truncated=res["truncated"])
                assert write_pt_batches(str(a), 0, n, res["device"], res["names"]) == k
                n += k
        assert n == 31 and sorted(os.listdir(a)) == sorted(os.listdir(b)) and len(os.listdir(b)) == 31
        for name in sorted(os.listdir(a)):
            x = torch.load(a / name, weights_only=True)
            y = torch.load(b / name, weights_only=True)
            assert list(x) == list(y)
            for key in x:
                assert x[key].dtype == y[key].dtype and x[key].shape == y[key].shape and torch.equal(x[key], y[key]), (name, key)
            with zipfile.ZipFile(a / name) as za, zipfile.ZipFile(b / name) as zb:
                assert zb.testzip() is None
                assert za.read(name[:-3] + "/data.pkl") == zb.read(name[:-3] + "/data.pkl")


@pytest.mark.parametrize("chop_type,ocq", [("all", False), ("terminal", False), ("internal", False), ("all", True)])
def test_fused_predict_chop_with_reference_batching_equals_predict_then_chop(env, tmp_path, chop_type, ocq):
    """`predict --chop --reference-batching` = the reference's DEFAULT two commands -- `predict -b 12`
    (left-padded batches of 12 whose pad tokens run through the model, deepchopper/cli.py:66-198, tokenizer.py:71-87) then
    `chop` over the `.pt` files (src/bin/predict.rs:130-192) -- as one streaming pass without the 28 B/token files.
    RANDOM weights: the logits of a read depend on the pads in front of it (token-gated weights would not notice), so the
    native zero-padding flow gives other predictions; a permissive chop policy so that reads really are chopped.
    Reads stay below the FFT threshold: every row takes a block-Toeplitz long convolution, whose results do not depend on
    what else shares the launch -- the fused launches and the one-launch-per-batch `.pt` route agree bit for bit, so the
    files must agree byte for byte (decompressed)."""
    import gzip

    from deepchopper_amd import synth
    from deepchopper_amd.chop import chop, predict_chop
    from deepchopper_amd.fastq import FastqReader
    from deepchopper_amd.predict import PredictPipeline, SmoothParams, write_pt_batch

    torch, sd, pipe0 = env
    sp = SmoothParams(window=21, min_interval=5, approved=40, max_process_intervals=40)
    pipe = PredictPipeline(pipe0.model, smooth=sp)
    assert pipe.shared_prefix
    seqs, quals, ids = synth.synth_reads_mixed(79, 150, 2400, seed=8)  # 6 batches of 12 + one of 7
    heads = [i.encode() + b" ch=%d" % k for k, i in enumerate(ids)]
    fq = tmp_path / "reads.fq"
    fq.write_bytes(synth.fastq_text(heads, seqs, quals))
    os.chdir(tmp_path)
    tag = f"{chop_type}_{int(ocq)}"
    final, n_reads, n_out = predict_chop(str(fq), pipe, chop_type, output_prefix=str(tmp_path / f"ref_{tag}"), output_chopped_seqs=ocq,
                                         max_reads=40, max_tokens=1 << 30, reference_batch=12)
    assert n_reads == 79
    got = gzip.open(final, "rb").read()
    # the reference's route: one launch and one `.pt` per batch of 12, then chop
    pdir = tmp_path / f"pred_{tag}" / "0"
    with FastqReader(str(fq)) as rd:
        for b, res in enumerate(pipe.predict_stream(rd.chunks(12, 1 << 30), reference_padding=True, ids_i64=True)):
            write_pt_batch(str(pdir / f"0_{b}.pt"), res["device"], res["names"])
    assert b == 6
    via_pt = chop([str(pdir)], str(fq), sp, chop_type, output_prefix=str(tmp_path / f"pt_{tag}"), output_chopped_seqs=ocq)
    want = gzip.open(via_pt, "rb").read()
    assert got == want
    assert os.path.basename(final) == os.path.basename(via_pt).replace("pt_", "ref_")  # same {n}pd.{m}record counts
    # ... the same without the shared [PAD] rows (every row's pads through the model, as rounds 1-4 ran it)
    padded = PredictPipeline(pipe0.model, smooth=sp)
    padded.shared_prefix = False
    f2, _, _ = predict_chop(str(fq), padded, chop_type, output_prefix=str(tmp_path / f"pad_{tag}"), output_chopped_seqs=ocq,
                            max_reads=40, max_tokens=1 << 30, reference_batch=12)
    assert gzip.open(f2, "rb").read() == want
    # ... and the flag does something: every read at its own length predicts (and chops) differently
    f3, n3, _ = predict_chop(str(fq), pipe, chop_type, output_prefix=str(tmp_path / f"nat_{tag}"), output_chopped_seqs=ocq)
    native = gzip.open(f3, "rb").read()
    assert n3 == 79 and native != want
    if chop_type == "all" and not ocq:
        recs = want.split(b"\n")[0::4]
        assert sum(1 for r in recs if r.endswith(b"|T") or r.endswith(b"|I")) > 20  # reads really were chopped


def test_cli_predict_chop_reference_batching(env, tmp_path):
    """The command line of the above: `deepchopper predict reads.fq --chop --reference-batching -b 12` against the two
    commands `deepchopper predict reads.fq -b 12 -o pred` + `deepchopper chop pred/0 reads.fq`."""
    import gzip

    from deepchopper_amd import synth
    from deepchopper_amd.weights import save_state_dict

    torch, sd, _ = env
    seqs, quals, ids = synth.synth_reads_mixed(50, 150, 1800, seed=31)
    fq = tmp_path / "reads.fq"
    fq.write_bytes(synth.to_fastq(seqs, quals, ids))
    ck = tmp_path / "ckpt"
    save_state_dict(sd, str(ck / "model.safetensors"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    envv = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    policy = ["--min-interval-size", "5", "--approved-intervals", "40", "--max-process-intervals", "40"]

    def run(*argv):
        p = subprocess.run([sys.executable, "-m", "deepchopper_amd.cli", *argv], cwd=tmp_path, env=envv, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        return p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""

    one = run("predict", str(fq), "-g", "1", "-m", str(ck), "-b", "12", "--chop", "--reference-batching", "--chop-output", "one", *policy)
    run("predict", str(fq), "-g", "1", "-m", str(ck), "-b", "12", "-o", "pred")
    two = run("chop", "pred/0", str(fq), "-o", "two", *policy)
    a, b = gzip.open(tmp_path / one, "rb").read(), gzip.open(tmp_path / two, "rb").read()
    assert a == b and a.count(b"\n") >= 4 * 50
    nat = run("predict", str(fq), "-g", "1", "-m", str(ck), "--chop", "--chop-output", "nat", *policy)
    assert gzip.open(tmp_path / nat, "rb").read() != a
