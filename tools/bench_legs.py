"""The legs of bench.py beside its headline: one per BASELINE.json config, so that the driver's record carries every config
at its stated shape (N = 1 only; none of them is `value`).  Each leg times the product path and then checks it -- against
the CPU oracle on a bounded sample (the oracle is the checker here, never the thing measured), or through record counts.

    configs.c1_512bp            configs[0]  1 000 x 512 bp reads, HBM-resident step loop (the whole config per step) + a 4 096-read batch
    configs.c3_2kb_1gpu_slice   configs[2]  2 kb reads, one GPU's slice of the read-sharded run, HBM-resident step loop
    configs.c4_mixed            configs[3]  log-uniform 100 bp - 10 kb FASTQ file -> predict_stream (token-budget batches) -> intervals
    configs.c5_gz_chop          configs[4]  nanopore-like .fastq.gz -> fused predict -> chop -> BGZF (-t 4), adapters planted
    file_e2e                    configs[1]  1 000 000 x 1 kb reads, FASTQ file -> labels + intervals, first vs last 100 k reads
    configs.reference_b12       the reference's default batching (`predict -b 12`: left-padded batches of 12, pad tokens through
                                the model), reference batches fused into shared launches, FASTQ file -> logits / labels / intervals
"""
from __future__ import annotations

import gzip
import os
import shutil
import tempfile
import time

import numpy as np


# ------------------------------------------------------------------------------------------------------------
# the checker
# ------------------------------------------------------------------------------------------------------------
def oracle_reads(sd, names, seqs, quals, batch: int = 12):
    """Oracle labels / smoothed labels / intervals for reads of EQUAL length (left-padding-free batches of `batch`, the
    reference's default batch size) or, with batch = 1, for any reads at their own length."""
    from oracle import dc_oracle as o
    from oracle import forward_ref as fr

    out = {}
    feats = []
    for i, s, q in zip(names, seqs, quals):
        rec = o.parse_fastq_record(i, s.decode("latin1"), q.decode("latin1"))
        feats.append(o.tokenize_and_align_labels_and_quals_ids(rec["id"], rec["seq"], rec["qual"], rec["target"]))
    for b0 in range(0, len(feats), batch):
        b = o.collate(feats[b0: b0 + batch])
        logits = fr.forward(sd, b["input_ids"], b["input_quals"]).numpy()
        preds = o.load_predicts_from_batch({"prediction": logits, "target": b["labels"], "seq": b["input_ids"], "id": b["id"]})
        for rid, p in preds.items():
            out[rid] = (p.prediction, p.smooth_label(21), p.smooth_and_select_intervals(21, 13, 20))
    return out


def cpu_baseline_leg(sd, read_len: int, threads: int, budget_s: float) -> dict:
    """The CPU oracle (kind "port") on reads of another length, at a given thread count: one warm-up batch of 12, then batches of
    12 until the budget is spent -- a sustained sample, timed as one run."""
    import torch

    from deepchopper_amd import synth

    n = 1200
    seq, qual, ids = synth.synth_reads_fixed(n, read_len, seed=42)
    seqs, quals = [seq[i].tobytes() for i in range(n)], [qual[i].tobytes() for i in range(n)]
    before = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        oracle_reads(sd, ids[:12], seqs[:12], quals[:12])
        t0 = time.perf_counter()
        done = 12
        while time.perf_counter() - t0 < budget_s and done + 12 <= n:
            oracle_reads(sd, ids[done: done + 12], seqs[done: done + 12], quals[done: done + 12])
            done += 12
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(before)
    return {"value": round((done - 12) / dt, 3), "unit": "reads/s", "cores": threads, "kind": "port",
            "sample": f"{done - 12} synthetic {read_len}-bp reads after a 12-read warm-up, oracle tokenise + forward (torch fp32, batch 12) + smooth, "
                      f"sustained over {dt:.1f} s"}


def f1_check(expected: dict, got: dict) -> dict:
    """expected / got: {read id: (labels int8[n], intervals [(s, e), ...])} -> per-base F1 of label 1 and interval identity."""
    tp = fp = fn = 0
    same_iv = same_lab = True
    for rid, (e_lab, _, e_iv) in expected.items():
        g_lab, g_iv = got[rid]
        tp += int(np.sum((g_lab == 1) & (e_lab == 1)))
        fp += int(np.sum((g_lab == 1) & (e_lab == 0)))
        fn += int(np.sum((g_lab == 0) & (e_lab == 1)))
        same_lab &= bool(np.array_equal(g_lab, e_lab))
        same_iv &= [tuple(x) for x in g_iv] == [tuple(x) for x in e_iv]
    f1 = 1.0 if (fp + fn) == 0 else 2 * tp / max(2 * tp + fp + fn, 1)
    return {"reads": len(expected), "per_base_f1_vs_oracle": round(f1, 6), "labels_identical": bool(same_lab),
            "intervals_identical": bool(same_iv)}


# ------------------------------------------------------------------------------------------------------------
# HBM-resident step loops at other read lengths (configs[0], configs[2])
# ------------------------------------------------------------------------------------------------------------
def resident_leg(pipe, sd, B: int, L: int, steps: int, what: str, check_reads: int = 12) -> dict:
    import torch

    from deepchopper_amd import ops, synth

    seq, qual, ids = synth.synth_reads_fixed(B, L, seed=42)
    offs = np.arange(B + 1, dtype=np.int64) * L
    d = (ops.to_device(seq.reshape(-1)), ops.to_device(qual.reshape(-1)), ops.to_device(offs))
    lens = np.full(B, L, np.int64)
    for _ in range(2):
        res = pipe.run_device_batch(*d, lens, want_labels=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = pipe.run_device_batch(*d, lens, want_labels=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": what, "value": round(B * steps / dt, 1), "unit": "reads/s", "Mbases_per_s": round(B * steps * L / dt / 1e6, 2),
           "reads_per_step": B, "read_len": L, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3)}
    if check_reads:
        n = min(check_reads, B)
        exp = oracle_reads(sd, ids[:n], [seq[i].tobytes() for i in range(n)], [qual[i].tobytes() for i in range(n)])
        o = res["smooth"]["offsets"].cpu().numpy()
        pred = res["smooth"]["pred"].cpu().numpy()
        iv, niv = res["smooth"]["intervals"].cpu().numpy(), res["smooth"]["n_intervals"].cpu().numpy()
        got = {ids[r]: (pred[o[r]: o[r + 1]], iv[r, : niv[r]].tolist()) for r in range(n)}
        out["check"] = f1_check(exp, got)
    return out


# ------------------------------------------------------------------------------------------------------------
# configs[3]: mixed lengths from a file
# ------------------------------------------------------------------------------------------------------------
def mixed_leg(pipe, sd, n_reads: int, check_reads: int = 6) -> dict:
    import torch

    from deepchopper_amd import synth
    from deepchopper_amd.fastq import FastqReader

    seqs, quals, ids = synth.synth_reads_mixed(n_reads, 100, 10000, seed=42)
    bases = sum(len(s) for s in seqs)
    tmp = tempfile.mkdtemp(prefix="dc_c4_")
    path = os.path.join(tmp, "mixed.fq")
    try:
        with open(path, "wb") as fh:
            fh.write(synth.to_fastq(seqs, quals, ids))
        # warm-up on the head of the file: every FFT size class, both Toeplitz forms, pinned staging
        with FastqReader(path) as rd:
            for i, _ in enumerate(pipe.predict_stream(rd.chunks(8192, 2_000_000))):
                if i >= 3:
                    break
        torch.cuda.synchronize()
        want = {ids[i] for i in range(min(400, n_reads)) if len(seqs[i]) <= 1500}
        want = set(sorted(want)[:check_reads])
        got = {}
        t0 = time.perf_counter()
        nr = n_iv = 0
        with FastqReader(path) as rd:
            for res in pipe.predict_stream(rd.chunks(8192, 2_000_000)):
                nr += len(res["names"])
                n_iv += int(res["n_intervals"].sum())
                if want:  # (a set lookup per read of the first chunks only; nothing once the sample is complete)
                    for r, name in enumerate(res["names"]):
                        if name in want:
                            o = res["offsets"]
                            got[name] = (np.array(res["pred"][o[r]: o[r + 1]]), res["intervals"][r, : res["n_intervals"][r]].tolist())
                            want.discard(name)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {"workload": "configs[3]: log-uniform 100 bp - 10 kb reads, FASTQ file -> parse -> token-budget batches (2 M tokens) -> "
                       "tokenise/forward/smooth -> labels + intervals on the host (predict_stream)",
           "value": round(nr / dt, 1), "unit": "reads/s", "Mbases_per_s": round(bases / dt / 1e6, 2), "reads": nr,
           "mean_bp": round(bases / n_reads, 1), "seconds": round(dt, 3), "intervals": n_iv}
    if got:
        idx = {n: i for i, n in enumerate(ids)}
        names = sorted(got)
        exp = oracle_reads(sd, names, [seqs[idx[n]] for n in names], [quals[idx[n]] for n in names], batch=1)
        out["check"] = f1_check(exp, got)
        out["check"]["what"] = "every sampled read at its own length == the reference at --batch-size 1 (SURVEY 0.6)"
    assert nr == n_reads, (nr, n_reads)
    return out


# ------------------------------------------------------------------------------------------------------------
# configs[4]: .fastq.gz -> predict -> chop -> BGZF
# ------------------------------------------------------------------------------------------------------------
def _count_output(path: str) -> dict:
    from deepchopper_amd.fastq import FastqReader

    n = n_t = n_i = bases = 0
    with FastqReader(path) as rd:
        for c in rd.chunks(65536, 256 << 20):
            raw, o = c._names.tobytes(), c._name_offsets
            n += c.n
            bases += int(c.offsets[c.n])
            for k in range(c.n):
                tail = raw[o[k + 1] - 2: o[k + 1]]
                n_t += tail == b"|T"
                n_i += tail == b"|I"
    return {"records": n, "terminal_chops": int(n_t), "internal_chop_segments": int(n_i), "passed_through": int(n - n_t - n_i),
            "bases_out": bases}


def gz_chop_leg(n_reads: int, threads: int = 4, check_reads: int = 24, reference_batch: int | None = None) -> dict:
    """Nanopore-like reads (log-normal lengths, ONT headers, adapters planted as runs of the base the token-gated
    weights label 1) -> gzip file -> `predict_chop` (the body of `deepchopper predict --chop`) -> chopped BGZF.
    Checks: (1) the head of the output file equals, byte for byte, what the oracle's forward + `chop_read` emit for the
    first `check_reads` reads; (2) the whole file parses and its record count is the one the driver reported.
    reference_batch = 12: `predict --chop --reference-batching` -- the reads run in the reference's left-padded batches of 12
    (pad tokens through the model, shared per launch), the oracle of check (1) runs those batches."""
    import torch

    from deepchopper_amd import synth
    from deepchopper_amd.chop import predict_chop
    from deepchopper_amd.model import DeepChopper
    from deepchopper_amd.predict import PredictPipeline
    from deepchopper_amd.weights import token_gated_state_dict

    tmp = tempfile.mkdtemp(prefix="dc_c5_")
    cwd = os.getcwd()
    try:
        path = os.path.join(tmp, "hct116_like.fastq.gz")
        t0 = time.perf_counter()
        n_bases = 0
        head = None
        with gzip.open(path, "wb", compresslevel=1) as fh:
            for b0 in range(0, n_reads, 50_000):
                n = min(50_000, n_reads - b0)
                h, s, q = synth.synth_nanopore_like(n, seed=42 + b0, start_index=b0)
                if head is None:
                    head = (h[:check_reads], s[:check_reads], q[:check_reads])
                n_bases += sum(len(x) for x in s)
                fh.write(synth.fastq_text(h, s, q))
        gen_s = time.perf_counter() - t0
        sd = token_gated_state_dict(0)
        pipe = PredictPipeline(DeepChopper.from_state_dict(sd))
        os.chdir(tmp)
        warm = os.path.join(tmp, "warm.fastq")
        h, s, q = synth.synth_nanopore_like(20_000, seed=7)
        with open(warm, "wb") as fh:
            fh.write(synth.fastq_text(h, s, q))
        final, _, _ = predict_chop(warm, pipe, threads=threads, max_reads=8192, max_tokens=2_000_000, reference_batch=reference_batch)
        os.remove(final)
        os.remove(warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks = [(0, t0)]

        def mark(res):
            marks.append((marks[-1][0] + int(res["lengths"].shape[0]), time.perf_counter()))

        final, n_in, n_out = predict_chop(path, pipe, threads=threads, max_reads=8192, max_tokens=2_000_000, reference_batch=reference_batch,
                                          on_batch=mark)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0

        def rate_between(lo, hi):  # reads/s between the first launch at or behind `lo` reads and the last one at or before `hi`
            a = next(m for m in marks if m[0] >= lo)
            b = next(m for m in reversed(marks) if m[0] <= hi)
            return round((b[0] - a[0]) / (b[1] - a[1]), 1) if b[0] > a[0] else None
        out = {"workload": "configs[4] flow on one GPU: nanopore-like reads (log-normal lengths, ONT headers, adapters planted), plain "
                           ".fastq.gz -> parallel inflate -> predict -> chop (src/bin/predict.rs:130-192) -> BGZF, no .pt in between" +
                           ("" if reference_batch is None else f"; the reads in the reference's left-padded batches of {reference_batch} "
                            "(--reference-batching: the predictions of the reference's default `predict -b 12`), [PAD] runs shared per launch"),
               "batching": "every read at its own length" if reference_batch is None else f"reference batches of {reference_batch}",
               "value": round(n_in / dt, 1), "unit": "reads/s", "Mbases_per_s": round(n_bases / dt / 1e6, 2), "reads": n_in,
               "mean_bp": round(n_bases / n_reads, 1), "seconds": round(dt, 3), "bgzf_threads": threads,
               "input_MB": round(os.path.getsize(path) / 1e6, 1), "output_MB": round(os.path.getsize(final) / 1e6, 1),
               "generate_s": round(gen_s, 1),
               # droop check inside the driver's own run: the rate over the first and the last third of the reads
               "first_third_reads_per_s": rate_between(0, n_in // 3), "last_third_reads_per_s": rate_between(n_in - n_in // 3, n_in)}
        counts = _count_output(final)
        out.update(counts)
        check = {"records_match_driver_report": counts["records"] == n_out, "reads_in_match": n_in == n_reads,
                 "bases_out_le_bases_in": counts["bases_out"] <= n_bases}
        if check_reads:
            check.update(_chop_head_check(sd, head, final, reference_batch or 1))
        out["check"] = check
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def _chop_head_check(sd, head, final_path: str, batch: int = 1) -> dict:
    """The first reads through the oracle: tokenise -> collate in batches of `batch` (left-padded, as the reference's
    DataLoader forms them) -> forward (torch fp32) -> argmax -> chop_read, compared with the head of the chopped file
    byte for byte."""
    from oracle import dc_oracle as o
    from oracle import forward_ref as fr

    headers, seqs, quals = head
    want = []
    feats = []
    for h, s, q in zip(headers, seqs, quals):
        name = h.split(b" ", 1)[0]
        rec = o.parse_fastq_record(name.decode(), s.decode(), q.decode())
        feats.append(o.tokenize_and_align_labels_and_quals_ids(rec["id"], rec["seq"], rec["qual"], rec["target"]))
    preds = {}
    for b0 in range(0, len(feats), batch):
        b = o.collate(feats[b0: b0 + batch])
        logits = fr.forward(sd, b["input_ids"], b["input_quals"]).numpy()
        preds.update(o.load_predicts_from_batch({"prediction": logits, "target": b["labels"], "seq": b["input_ids"], "id": b["id"]}))
    for h, s, q in zip(headers, seqs, quals):
        name = h.split(b" ", 1)[0]
        p = preds[name.decode()]
        for rid, rs, rq in o.chop_read(p, name, s, q, fq_header=h):  # the chop binary's defaults (src/bin/predict.rs:39-64)
            want.append(b"@" + rid + b"\n" + rs + b"\n+\n" + rq + b"\n")
    want = b"".join(want)
    with gzip.open(final_path, "rb") as fh:
        got = fh.read(len(want))
    return {"head_reads_vs_oracle": len(headers), "head_bytes_identical": got == want}


# ------------------------------------------------------------------------------------------------------------
# configs[1] from a file, at its stated size
# ------------------------------------------------------------------------------------------------------------
def file_e2e_leg(pipe, sd, n_reads: int, read_len: int, batch: int, check_reads: int = 12, gen_threads: int = 8) -> dict:
    """FASTQ-inclusive rate ("FASTQ bytes in -> per-base labels + intervals out") over `n_reads` reads --
    configs[1] says 1 000 000 x 1 kb: about half a minute of sustained work, so clock droop shows: the rate over the first
    and over the last 100 000 reads is reported next to the whole run's."""
    import torch

    from deepchopper_amd import synth
    from deepchopper_amd.fastq import FastqReader

    tmp = tempfile.mkdtemp(prefix="dc_bench_")
    path = os.path.join(tmp, "reads.fq")
    try:
        t0 = time.perf_counter()
        nbytes = synth.write_fastq_fixed_fast(path, n_reads, read_len, seed=42, threads=gen_threads)
        gen_s = time.perf_counter() - t0
        cap = batch * (read_len + 1)
        with FastqReader(path) as rd:  # warm-up: pinned staging, copy stream
            for i, _ in enumerate(pipe.predict_stream(rd.chunks(batch, cap))):
                if i >= 2:
                    break
        torch.cuda.synchronize()
        marks = [(0, time.perf_counter())]
        n = n_iv = 0
        first = None
        t0 = marks[0][1]
        with FastqReader(path) as rd:
            for res in pipe.predict_stream(rd.chunks(batch, cap)):
                if first is None and check_reads:
                    o = res["offsets"]
                    first = {"names": res["names"][:check_reads],
                             "seq": [bytes(res["seq"][res["in_offsets"][r]: res["in_offsets"][r + 1]]) for r in range(check_reads)],
                             "qual": [bytes(res["qual"][res["in_offsets"][r]: res["in_offsets"][r + 1]]) for r in range(check_reads)],
                             "got": {res["names"][r]: (np.array(res["pred"][o[r]: o[r + 1]]),
                                                       res["intervals"][r, : res["n_intervals"][r]].tolist()) for r in range(check_reads)}}
                n += len(res["names"])
                n_iv += int(res["n_intervals"].sum())
                marks.append((n, time.perf_counter()))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def rate_between(lo: int, hi: int):
        a = next(m for m in marks if m[0] >= lo)
        b = next(m for m in reversed(marks) if m[0] <= hi)
        return round((b[0] - a[0]) / (b[1] - a[1]), 1) if b[0] > a[0] else None

    out = {"value": round(n / dt, 1), "unit": "reads/s", "reads": n, "seconds": round(dt, 3), "container": "plain FASTQ",
           "file_GB": round(nbytes / 1e9, 3), "generate_s": round(gen_s, 1), "intervals": n_iv,
           "what": "file -> parse -> H2D -> tokenise/forward/smooth -> D2H labels + intervals (predict_stream)"}
    if n >= 300_000:
        out["first_100k_reads_per_s"] = rate_between(0, 100_000)
        out["last_100k_reads_per_s"] = rate_between(n - 100_000, n)
    if first is not None:
        exp = oracle_reads(sd, first["names"], first["seq"], first["qual"])
        out["check"] = f1_check(exp, first["got"])
        out["check"]["reads_counted"] = n == n_reads
    return out


# ------------------------------------------------------------------------------------------------------------
# the reference's default batching (-b 12)
# ------------------------------------------------------------------------------------------------------------
def reference_b12_leg(pipe, sd, n_reads: int, read_len: int = 1000, mix: str = "uniform", write_pt: bool = False) -> dict:
    """FASTQ file -> batches of 12 as the reference's DataLoader forms them (deepchopper/cli.py:70) -> left-padded per batch,
    pad tokens through the model -> logits + labels + intervals on the host; reference batches share GPU
    launches (`fastq.fuse_chunks`), each with its own padding.  mix = "uniform": log-uniform lengths read_len / 2 ... 3 read_len / 2;
    "nanopore": config 5's direct-RNA-like length mix (`synth.synth_nanopore_like`: log-normal, median 900, up to 30 kb).
    `padded_over_real_tokens` = the tokens the reference's batches hold / the tokens of the reads themselves.
    write_pt: every batch also becomes its `{rank}_{batch}.pt` (CustomWriter, callbacks.py:12-25) through predict.PtWriter -- the
    work of `deepchopper predict reads.fq` without flags, in process (no interpreter start-up); the first file is read back with
    torch.load and compared with the launch's logits.
    Check: the first batch against the oracle run on that batch of 12 (logits <= 1e-3, labels, intervals)."""
    import torch

    from deepchopper_amd import synth
    from deepchopper_amd.fastq import FastqReader, fuse_chunks

    if mix == "nanopore":
        heads, seqs, quals = synth.synth_nanopore_like(n_reads, seed=77, gate=b"A")  # (synthetic weights: no token gate to plant for)
        ids = [h.split(b" ")[0].decode() for h in heads]
        text = synth.fastq_text(heads, seqs, quals)
    else:
        seqs, quals, ids = synth.synth_reads_mixed(n_reads, read_len // 2, read_len + read_len // 2, seed=77)
        text = synth.to_fastq(seqs, quals, ids)
    lens = np.array([len(x) for x in seqs], np.int64)
    bases = int(lens.sum())
    padded = sum(int(lens[i: i + 12].max() + 1) * len(lens[i: i + 12]) for i in range(0, n_reads, 12))
    tmp = tempfile.mkdtemp(prefix="dc_ref_")
    path = os.path.join(tmp, "reads.fq")
    try:
        with open(path, "wb") as fh:
            fh.write(text)
        del text

        pt_dir = os.path.join(tmp, "predictions", "0")
        pt_stats = {}

        def run(limit=None):
            from deepchopper_amd.predict import PtWriter

            writer = PtWriter(threads=4) if write_pt else None
            try:
                return _run(limit, writer)
            except BaseException:
                if writer is not None:
                    try:
                        writer.close()  # (nothing may still read the launch's pinned copies when the leg's directory goes)
                    except Exception:
                        pass
                raise

        def _run(limit, writer):
            n = n_files = 0
            first = None
            with FastqReader(path) as rd:
                for res in pipe.predict_stream(fuse_chunks(rd.chunks(12, 1 << 30), 8192, 2_000_000, padded=not pipe.shared_prefix), reference_padding=True,
                                               ids_i64=not write_pt):
                    if writer is not None:
                        n_files += writer.submit(pt_dir, 0, n_files, res["device"], res["names"], ready_event=res["ready_event"],
                                                 truncated=res["truncated"])
                    if first is None:
                        k, c0 = res["sub_sizes"][0], res["sub_cols"][0]
                        o = res["offsets"]
                        first = {"logits": res["device"]["logits"][:k, c0:].cpu().numpy(), "names": res["names"][:k],
                                 "got": {res["names"][r]: (np.array(res["pred"][o[r]: o[r + 1]]),
                                                           res["intervals"][r, : res["n_intervals"][r]].tolist()) for r in range(k)}}
                    n += len(res["names"])
                    if limit and n >= limit:
                        break
            if writer is not None:
                writer.close()
                pt_stats.update({"files": n_files, "main_thread_waits_s": {k: round(v, 3) for k, v in writer.timers.items()}})
            return n, first

        run(limit=3000)  # warm-up
        shutil.rmtree(os.path.join(tmp, "predictions"), ignore_errors=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n, first = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if write_pt:
            names = os.listdir(pt_dir)
            pt_stats["MB"] = round(sum(os.path.getsize(os.path.join(pt_dir, f)) for f in names) / 1e6, 1)
            pt_stats["MB_per_s"] = round(pt_stats["MB"] / dt, 1)
            d0 = torch.load(os.path.join(pt_dir, "0_0.pt"), weights_only=True)
            pt_stats["first_file_equals_the_launch"] = bool(
                len(names) == pt_stats["files"] and np.array_equal(d0["prediction"].numpy(), first["logits"])
                and d0["seq"].dtype == torch.int64 and d0["target"].dtype == torch.int64 and tuple(d0["id"].shape) == (len(first["names"]), 256))
            shutil.rmtree(os.path.join(tmp, "predictions"), ignore_errors=True)
        stage_ms = None
        if os.environ.get("DC_LEG_STAGES"):  # a second, untimed pass with the kernels bracketed by events: where the GPU time goes
            net = pipe.model.net
            net.set_profiling(True)
            net.profile_reset()
            try:
                run()
                torch.cuda.synchronize()
                prof = net.profile()
            finally:
                net.set_profiling(False)
            fw = prof.get("forward", (0.0, 0))[0]
            stage_ms = {k: {"ms": round(v[0], 1), "launches": int(v[1]), "share_of_forward": round(v[0] / fw, 4) if fw else None}
                        for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {"workload": "the reference's default batching: FASTQ file -> batches of 12 (left-padded per batch, pad tokens through the "
                       "model) -> logits + labels + intervals on the host; batches fused into shared launches, each with its own padding",
           "lengths": "log-uniform %d-%d bp" % (read_len // 2, read_len + read_len // 2) if mix != "nanopore" else
                      "nanopore-like (log-normal, median 900 bp, sigma 0.75, 100 bp - 30 kb)",
           "value": round(n / dt, 1), "unit": "reads/s", "Mbases_per_s": round(bases / dt / 1e6, 2), "reads": n, "seconds": round(dt, 3),
           "mean_bp": round(bases / n_reads, 1), "max_bp": int(lens.max()),
           "padded_over_real_tokens": round(padded / float(bases + n_reads), 3),
           "shared_pad_prefix": bool(getattr(pipe, "shared_prefix", False))}
    if stage_ms is not None:
        out["stages"] = stage_ms
    if write_pt:
        out["workload"] = ("`deepchopper predict reads.fq` without flags, in process: FASTQ file -> the reference's batches of 12 -> one `.pt` "
                           "per batch (CustomWriter's wire format, 28 B per padded token: storages packed on the GPU, files written by "
                           "libdeepchopper_io) + labels + intervals")
        out["pt"] = pt_stats
    from oracle import dc_oracle as o
    from oracle import forward_ref as fr

    feats = []
    for i in range(12):
        rec = o.parse_fastq_record(ids[i], seqs[i].decode(), quals[i].decode())
        feats.append(o.tokenize_and_align_labels_and_quals_ids(rec["id"], rec["seq"], rec["qual"], rec["target"]))
    b = o.collate(feats)
    ref = fr.forward(sd, b["input_ids"], b["input_quals"]).numpy()
    preds = o.load_predicts_from_batch({"prediction": ref, "target": b["labels"], "seq": b["input_ids"], "id": b["id"]})
    exp = {rid: (p.prediction, p.smooth_label(21), p.smooth_and_select_intervals(21, 13, 20)) for rid, p in preds.items()}
    out["check"] = f1_check(exp, first["got"])
    out["check"]["max_abs_dlogit_vs_batched_oracle"] = float(np.abs(first["logits"] - ref).max())
    out["check"]["reads_counted"] = n == n_reads
    return out
