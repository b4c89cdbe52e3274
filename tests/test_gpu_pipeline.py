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


def test_mixed_length_native_equals_per_read_oracle(env):
    torch, sd, pipe = env
    from deepchopper_amd import synth

    seqs, quals, ids = synth.synth_reads_mixed(24, 100, 1500, seed=42)
    out, res = pipe.predict_reads(ids, seqs, quals)
    ref = _oracle_reads(sd, ids, seqs, quals)
    for r in out:
        p, _ = ref[r["id"]]
        assert len(r["prediction"]) == len(p.prediction)
        assert _margin_ok(r["prediction"], p.prediction)
        if np.array_equal(r["prediction"], p.prediction):
            assert np.array_equal(r["smooth"], p.smooth_label(21))
            assert r["intervals"] == p.smooth_and_select_intervals(21, 13, 20)


def test_reference_padding_equals_batched_oracle(env):
    torch, sd, pipe = env
    from deepchopper_amd import synth

    seqs, quals, ids = synth.synth_reads_mixed(12, 150, 700, seed=7)
    out, res = pipe.predict_reads(ids, seqs, quals, reference_padding=True)
    ref = _oracle_reads(sd, ids, seqs, quals, batch=12)
    got = res["logits"].cpu().numpy()
    _, ref_logits = ref[ids[0]]
    assert np.abs(got - ref_logits).max() < 2e-4
    for r in out:
        p, _ = ref[r["id"]]
        assert _margin_ok(r["prediction"], p.prediction)


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
