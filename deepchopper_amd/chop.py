"""`deepchopper chop` driver: consumer of the predict -> chop wire format.

Reference: `deepchopper-chop` = src/bin/predict.rs:130-384 (process_chunk + main) with
output/split.rs:60-226 for the record splitting.  Per 10 000-read chunk the per-read work of the
reference's `par_iter` (majority vote -> regions -> filter -> complement) is ONE fused GPU launch
(`dc_smooth_intervals_from_labels`); the host only slices byte strings and writes records.
"""
from __future__ import annotations

import os

import numpy as np

from . import ops, rs_api
from .fastq import BgzfWriter, iter_fastq
from .predict import SmoothParams

CHOP_TYPES = ("terminal", "internal", "all")


def split_record_by_remove_intervals(seq: bytes, rid: bytes, qual: bytes, keep: list[tuple[int, int]],
                                     min_chop_read_length: int, id_annotation: bool, chop_type: str):
    """split_noodle_records_by_remove_intervals (output/split.rs:171-226) given the kept segments
    (`generate_unmaped_intervals` already evaluated on the GPU)."""
    n_before = len(keep)
    cur = "terminal" if n_before == 1 else "internal"
    parts = [(rid + b"|%d:%d" % (s, e), seq[s:e], qual[s:e]) for s, e in keep]
    parts = [p for p in parts if len(p[1]) >= min_chop_read_length]
    if ((chop_type == "terminal" and cur == "internal") or (chop_type == "internal" and cur == "terminal")
            or (parts and len(parts[0][1]) == len(seq))):
        return [(rid, seq, qual)]
    if id_annotation:
        tag = b"|T" if cur == "terminal" else b"|I"
        parts = [(i + tag, s, q) for i, s, q in parts]
    return parts


def chop_chunk(records, predicts: dict, sp: SmoothParams, chop_type: str = "all"):
    """process_chunk (src/bin/predict.rs:130-192, `--ocq` off).  records: [(name, seq, qual)]."""
    todo, out_slots = [], []
    out: list[list] = []
    for name, seq, qual in records:
        p = predicts.get(name)
        if p is None:
            continue  # records without predictions are skipped
        slot = len(out)
        out.append([(name.encode(), seq, qual)])
        if p.seq_len() < sp.min_read_len:
            continue
        todo.append((slot, p, name, qual))
        out_slots.append(slot)
    if todo:
        labels = [np.asarray(p.prediction, dtype=np.int8).tobytes() for _, p, _, _ in todo]
        lab, offs = ops.pack_reads(labels)
        res = ops.smooth_intervals_from_labels(ops.to_device(lab.view(np.int8)), ops.to_device(offs), window=sp.window,
                                               min_interval=sp.min_interval, approved=sp.approved,
                                               max_intervals=max(sp.approved, 1), want_smooth=False)
        niv = res["n_intervals"].cpu().numpy()
        keep = res["keep"].cpu().numpy()
        nkeep = res["n_keep"].cpu().numpy()
        for k, (slot, p, name, qual) in enumerate(todo):
            if niv[k] > sp.max_process_intervals or niv[k] == 0:
                continue
            if p.seq_len() != len(qual):
                continue  # truncated prediction: do not process (predict.rs:160-164)
            segs = [tuple(x) for x in keep[k, : nkeep[k]].tolist()]
            out[slot] = split_record_by_remove_intervals(p.seq.encode(), name.encode(), qual, segs,
                                                         sp.min_read_length_after_chop, True, chop_type)
    return [r for recs in out for r in recs]


def chop(predict_dirs: list[str], fq: str, sp: SmoothParams | None = None, chop_type: str = "all",
         output_prefix: str | None = None, chunk_size: int = 10000, max_batch_size: int | None = None,
         threads: int = 2) -> str:
    """main() of src/bin/predict.rs:197-384.  Returns the output path
    `{prefix or fastq stem}.{n_predictions}pd.{n_records}record.chop.fq.gz` -- a BGZF stream written by
    `threads` compression workers, as the reference's bgzf::io::MultithreadedWriter (predict.rs:266-270)."""
    if chop_type not in CHOP_TYPES:
        raise ValueError("Invalid chop type")
    sp = sp or SmoothParams()
    predicts: dict = {}
    for d in predict_dirs:
        predicts.update(rs_api.load_predicts_from_batch_pts(d, rs_api.IGNORE_LABEL, max_batch_size))
    out_dir = os.path.dirname(os.path.abspath(output_prefix or fq)) or "."
    tmp = os.path.join(out_dir, f".deepchopper_temp_{os.getpid()}.fq.gz")
    n_out = 0
    try:
        with BgzfWriter(tmp, threads=threads) as fh:
            chunk = []

            def flush():
                nonlocal n_out
                for rid, seq, qual in chop_chunk(chunk, predicts, sp, chop_type):
                    fh.write_record(rid, seq, qual)
                    n_out += 1
                chunk.clear()

            for rec in iter_fastq(fq):
                chunk.append(rec)
                if len(chunk) >= chunk_size:
                    flush()
            if chunk:
                flush()
        stem = output_prefix or os.path.splitext(os.path.basename(fq))[0]
        final = f"{stem}.{len(predicts)}pd.{n_out}record.chop.fq.gz"
        if not os.path.isabs(final) and output_prefix is None:
            final = os.path.join(os.getcwd(), final)
        os.replace(tmp, final)
    except Exception:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise
    return final
