"""`deepchopper chop` in two forms.

* `chop(...)`: consumer of the reference's predict -> chop wire format, `deepchopper-chop` =
  src/bin/predict.rs:130-384 (process_chunk + main) with output/split.rs:60-226 for the record splitting.
  Per 10 000-read chunk the per-read work of the reference's `par_iter` (majority vote -> regions -> filter ->
  complement) is ONE fused GPU launch (`dc_smooth_intervals_from_labels`).
* `predict_chop(...)`: the fused streaming flow FASTQ(.gz) -> predict -> chop -> BGZF with no `.pt` in between
  (the 28 B/token wire format is ~280 GB for 5 M nanopore reads): `PredictPipeline.predict_stream` leaves every
  batch's intervals / kept segments on the host, and `dc_chop_write_chunk` (include/deepchopper_io.h) applies the
  same gating and record splitting natively, straight from the reader's packed chunk into the BGZF writer.

Both produce the same bytes for the same predictions (tests/test_gpu_pipeline.py).
"""
from __future__ import annotations

import os

import numpy as np

from . import _iolib, ops, rs_api
from .fastq import BgzfWriter, FastqReader
from .predict import SmoothParams

CHOP_TYPES = ("terminal", "internal", "all")
LAST_WRITER_STATS: dict = {}  # host-budget accounting of the last predict_chop() of this process


def chop_params(sp: SmoothParams, chop_type: str = "all", output_chopped_seqs: bool = False) -> "_iolib.ChopParams":
    if chop_type not in CHOP_TYPES:
        raise ValueError("Invalid chop type")
    return _iolib.ChopParams(min_read_len=sp.min_read_len, max_process_intervals=sp.max_process_intervals,
                             min_read_length_after_chop=sp.min_read_length_after_chop,
                             chop_type=CHOP_TYPES.index(chop_type), output_chopped_seqs=int(output_chopped_seqs),
                             id_annotation=1)


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


def split_record_by_intervals(seq: bytes, rid: bytes, qual: bytes, intervals: list[tuple[int, int]]):
    """split_noodle_records_by_intervals (output/split.rs:138-169), the `--ocq` form: the chopped segments."""
    return [(rid + b"|%d:%d" % (s, e), seq[s:e], qual[s:e]) for s, e in intervals]


def chop_chunk(records, predicts: dict, sp: SmoothParams, chop_type: str = "all", output_chopped_seqs: bool = False):
    """process_chunk (src/bin/predict.rs:130-192).  records: [(name, header, seq, qual)], `header` = the whole
    definition line (records that pass through keep their description, as fq_record.clone() does)."""
    todo = []
    out: list[list] = []
    for name, header, seq, qual in records:
        p = predicts.get(name)
        if p is None:
            continue  # records without predictions are skipped
        slot = len(out)
        out.append([(header.encode("latin1"), seq, qual)])
        if p.seq_len() < sp.min_read_len:
            continue
        todo.append((slot, p, name, qual))
    if todo:
        labels = [np.asarray(p.prediction, dtype=np.int8).tobytes() for _, p, _, _ in todo]
        lab, offs = ops.pack_reads(labels)
        res = ops.smooth_intervals_from_labels(ops.to_device(lab.view(np.int8)), ops.to_device(offs), window=sp.window,
                                               min_interval=sp.min_interval, approved=sp.approved,
                                               max_intervals=max(sp.approved, 1), want_smooth=False)
        niv = res["n_intervals"].cpu().numpy()
        iv = res["intervals"].cpu().numpy()
        keep = res["keep"].cpu().numpy()
        nkeep = res["n_keep"].cpu().numpy()
        for k, (slot, p, name, qual) in enumerate(todo):
            if niv[k] > sp.max_process_intervals or niv[k] == 0:
                continue
            if p.seq_len() != len(qual):
                continue  # truncated prediction: do not process (predict.rs:160-164)
            if output_chopped_seqs:
                out[slot] = split_record_by_intervals(p.seq.encode(), name.encode("latin1"), qual,
                                                      [tuple(x) for x in iv[k, : niv[k]].tolist()])
            else:
                segs = [tuple(x) for x in keep[k, : nkeep[k]].tolist()]
                out[slot] = split_record_by_remove_intervals(p.seq.encode(), name.encode("latin1"), qual, segs,
                                                             sp.min_read_length_after_chop, True, chop_type)
    return [r for recs in out for r in recs]


def _final_name(fq: str, output_prefix: str | None, n_predictions: int, n_records: int) -> str:
    """`{prefix or fastq stem}.{n_predictions}pd.{n_records}record.chop.fq.gz` (predict.rs:343-356; file_stem
    strips one extension, the file lands in the working directory when no prefix is given)."""
    stem = output_prefix or os.path.splitext(os.path.basename(fq))[0]
    return f"{stem}.{n_predictions}pd.{n_records}record.chop.fq.gz"


def _temp_name(fq: str, output_prefix: str | None, tag: str = "") -> str:
    out_dir = os.path.dirname(os.path.abspath(output_prefix or fq)) or "."  # same filesystem as the output (predict.rs:252-266)
    return os.path.join(out_dir, f".deepchopper_temp_{os.getpid()}{tag}.fq.gz")


def chop(predict_dirs: list[str], fq: str, sp: SmoothParams | None = None, chop_type: str = "all",
         output_prefix: str | None = None, chunk_size: int = 10000, max_batch_size: int | None = None,
         threads: int = 2, output_chopped_seqs: bool = False) -> str:
    """main() of src/bin/predict.rs:197-384.  Returns the output path
    `{prefix or fastq stem}.{n_predictions}pd.{n_records}record.chop.fq.gz` -- a BGZF stream written by
    `threads` compression workers, as the reference's bgzf::io::MultithreadedWriter (predict.rs:266-270).

    The predictions come out of the `.pt` files natively (`rs_api.LabelStore`: mmap + argmax + mask on a pool of threads);
    per FASTQ chunk the reads' labels go through ONE fused GPU launch (majority vote -> regions -> filter -> complement,
    `dc_smooth_intervals_from_labels`) and the gating / record splitting / FASTQ text / BGZF of process_chunk
    (predict.rs:130-192) run in libdeepchopper_io (`dc_chop_write_chunk`) straight from the reader's packed chunk -- no
    per-read Python.  Chopped records take their letters from the FASTQ record through the token alphabet (A C G T, U -> T,
    anything else N), which is what `Predict.seq` holds when the predictions were made from this FASTQ.
    DC_CHOP_PYTHON=1: the per-record form (`chop_chunk`, the line-by-line restatement of process_chunk)."""
    if chop_type not in CHOP_TYPES:
        raise ValueError("Invalid chop type")
    sp = sp or SmoothParams()
    python_form = os.environ.get("DC_CHOP_PYTHON") == "1"
    if python_form:
        predicts: dict = {}
        for d in predict_dirs:
            predicts.update(rs_api.load_predicts_from_batch_pts(d, rs_api.IGNORE_LABEL, max_batch_size))
        n_predictions = len(predicts)
    else:
        store = rs_api.LabelStore.load(predict_dirs, rs_api.IGNORE_LABEL, max_batch_size, threads=max(4, threads))
        n_predictions = len(store)
        params = chop_params(sp, chop_type, output_chopped_seqs)
    tmp = _temp_name(fq, output_prefix)
    n_out = n_in = 0
    try:
        with BgzfWriter(tmp, threads=threads) as fh, FastqReader(fq, full_header=True, strict_four_line=True) as rd:
            for c in rd.chunks(chunk_size, 64 << 20):
                names = c.names()
                n_in += c.n
                if python_form:
                    hdr, ho = c._names.tobytes(), c._name_offsets
                    s, q, o = c.seq.tobytes(), c.qual.tobytes(), c.offsets
                    recs = [(names[i], hdr[ho[i]: ho[i + 1]].decode("latin1"), s[o[i]: o[i + 1]], q[o[i]: o[i + 1]])
                            for i in range(c.n)]
                    for rid, seq, qual in chop_chunk(recs, predicts, sp, chop_type, output_chopped_seqs):
                        fh.write_record(rid, seq, qual)
                        n_out += 1
                else:
                    n_out += _chop_chunk_native(fh, params, sp, c, names, store)
        if n_in == 0:
            raise ValueError(f"No valid records found in {fq}")
        final = _final_name(fq, output_prefix, n_predictions, n_out)
        os.replace(tmp, final)
    except Exception:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise
    return final


def _chop_chunk_native(fh: BgzfWriter, params, sp: SmoothParams, c, names: list[str], store) -> int:
    """One FASTQ chunk of `chop`: look the reads up, smooth their labels on the GPU, gate / split / write natively.  Records
    without a prediction are skipped (predict.rs:139-142): the chunk is handled in runs of consecutive reads that have one."""
    index = store.index
    idx = np.fromiter((index.get(nm, -1) for nm in names), np.int64, c.n)
    have = idx >= 0
    if not have.any():
        return 0
    edges = np.flatnonzero(np.diff(np.concatenate([[False], have, [False]]).astype(np.int8)))
    n_out = 0
    for r0, r1 in zip(edges[0::2].tolist(), edges[1::2].tolist()):
        ii = idx[r0:r1]
        k = r1 - r0
        so = store.offsets
        lens = so[ii + 1] - so[ii]
        slot = np.maximum(lens, 1)  # (a prediction without unmasked positions keeps a one-label slot: it passes through on its length)
        offs = np.zeros(k + 1, np.int64)
        np.cumsum(slot, out=offs[1:])
        if k == int(ii[-1] - ii[0]) + 1 and bool((np.diff(ii) == 1).all()) and bool((lens > 0).all()):
            lab = store.labels[so[ii[0]]: so[ii[-1] + 1]]  # the files follow the FASTQ's order: one slice
        else:
            lab = np.zeros(int(offs[-1]), np.int8)
            for j, i in enumerate(ii.tolist()):
                lab[offs[j]: offs[j] + lens[j]] = store.labels[so[i]: so[i + 1]]
        res = ops.smooth_intervals_from_labels(ops.to_device(np.ascontiguousarray(lab)), ops.to_device(offs), window=sp.window,
                                               min_interval=sp.min_interval, approved=sp.approved,
                                               max_intervals=max(sp.approved, 1), want_smooth=False)
        nl = c._name_len
        n_out += fh.chop_write_chunk(params, k, c._names, c._name_offsets[r0:], None if nl is None else nl[r0:], c.seq, c.qual,
                                     c.offsets[r0:], lens.astype(np.int32), res["n_intervals"].cpu().numpy(),
                                     res["intervals"].cpu().numpy(), res["n_keep"].cpu().numpy(), res["keep"].cpu().numpy())
    return n_out


def predict_chop_stream(pipe, chunks, writer: BgzfWriter, chop_type: str = "all", output_chopped_seqs: bool = False,
                        on_batch=None, reference_padding: bool = False, mark_sub_batches: bool = False) -> tuple[int, int]:
    """The fused flow on an open writer: every chunk goes FASTQ bytes -> tokenise -> forward -> smoothing ->
    intervals on the GPU, then gating + splitting + BGZF on the host library.  Chunks must come from a reader
    opened with full_header=True.  Returns (reads predicted, records written).

    reference_padding: the chunks are the reference's batches (`predict -b`, deepchopper/cli.py:70), alone or fused into
    shared launches (`fastq.fuse_chunks`, `sub_sizes`): every batch is left-padded to its longest read and the pad tokens
    count as the reference's collator and model make them count -- the predictions, and so the chopped
    records, are those of the reference's two commands `predict` -> `chop` (src/bin/predict.rs:130-192), without the
    28 B/token `.pt` files in between.  mark_sub_batches: end a BGZF block behind every original batch instead of behind
    every launch (workers of a run whose batches were dealt round-robin: the parent splices batch by batch)."""
    params = chop_params(pipe.smooth, chop_type, output_chopped_seqs)
    n_reads = n_out = 0
    for res in pipe.predict_stream(chunks, want_labels=False, borrow=True, want_names=False, reference_padding=reference_padding):
        n = int(res["lengths"].shape[0])
        subs = res.get("sub_sizes") if mark_sub_batches else None
        r0 = 0
        for k in (subs or [n]):  # (offsets are absolute into the chunk's packed buffers: a row range is a view of the per-read arrays)
            nl = res["name_len"]
            n_out += writer.chop_write_chunk(params, k, res["headers"], res["header_offsets"][r0:], None if nl is None else nl[r0:],
                                             res["seq"], res["qual"], res["in_offsets"][r0:], res["pred_len"][r0:],
                                             res["n_intervals"][r0:], res["intervals"][r0:], res["n_keep"][r0:], res["keep"][r0:])
            writer.mark()  # a chunk's records end their BGZF block: per-GPU parts can be spliced in input order
            r0 += k
        assert r0 == n
        n_reads += n
        if on_batch is not None:
            on_batch(res)
    return n_reads, n_out


def predict_chop(fq: str, pipe, chop_type: str = "all", output_prefix: str | None = None, threads: int = 2,
                 output_chopped_seqs: bool = False, max_reads: int = 8192, max_tokens: int = 2_000_000,
                 chunks=None, level: int = 6, part_path: str | None = None, reference_batch: int | None = None,
                 mark_sub_batches: bool = False, on_batch=None):
    """`deepchopper predict` + `deepchopper chop` as one streaming pass (no `.pt` files).

    reference_batch = B: the reads run in the reference's batches of B (`predict -b B`: consecutive reads, left-padded per
    batch, pad tokens through the model), fused into launches of at most `max_reads` reads / `max_tokens` tokens; `chunks`,
    when given, must then be those batches (reader chunks of B reads).  Default: every read at its own length.

    Output naming and the temp-file-then-rename protocol are the chop binary's (predict.rs:252-366).  With
    `part_path` (multi-GPU workers) the records go to that file instead and nothing is renamed: the parent
    concatenates the per-rank BGZF parts.  Returns (path, reads, records)."""
    if chop_type not in CHOP_TYPES:
        raise ValueError("Invalid chop type")
    tmp = part_path or _temp_name(fq, output_prefix)
    try:
        fh = BgzfWriter(tmp, threads=threads, level=level)
        with fh:
            ref = reference_batch is not None

            def batches(src):
                if not ref:
                    return src
                from .fastq import fuse_chunks

                return fuse_chunks(src, max_reads, max_tokens, padded=not getattr(pipe, "shared_prefix", False))

            if chunks is None:
                with FastqReader(fq, full_header=True, strict_four_line=True) as rd:
                    src = rd.chunks(reference_batch, 1 << 30) if ref else rd.chunks(max_reads, max_tokens)
                    n_reads, n_out = predict_chop_stream(pipe, batches(src), fh, chop_type, output_chopped_seqs, on_batch=on_batch,
                                                         reference_padding=ref, mark_sub_batches=mark_sub_batches)
            else:
                n_reads, n_out = predict_chop_stream(pipe, batches(chunks), fh, chop_type, output_chopped_seqs, on_batch=on_batch,
                                                     reference_padding=ref, mark_sub_batches=mark_sub_batches)
        LAST_WRITER_STATS.update({"deflate_cpu_s": fh.deflate_cpu_s, "bgzf_bytes": fh.data_bytes, "threads": threads})
        if part_path:  # the writer is closed: the chunk offsets are known
            write_part_index(part_path, fh)
            return tmp, n_reads, n_out
        if n_reads == 0:
            raise ValueError(f"No valid records found in {fq}")
        final = _final_name(fq, output_prefix, n_reads, n_out)
        os.replace(tmp, final)
    except Exception:
        for f in (tmp, tmp + ".idx.npy"):
            if os.path.exists(f):
                os.remove(f)
        raise
    return final, n_reads, n_out


BGZF_EOF = bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def write_part_index(part_path: str, writer: BgzfWriter) -> None:
    """Sidecar of a per-rank part (after the writer is closed): the file offset behind every chunk (writer.mark())
    and, last, the number of bytes in front of the EOF block."""
    np.save(part_path + ".idx.npy", np.concatenate([writer.mark_offsets, [writer.data_bytes]]).astype(np.int64))


def remove_parts(parts: list[str]) -> None:
    for p in parts:
        for f in (p, p + ".idx.npy"):
            if os.path.exists(f):
                os.remove(f)


def merge_bgzf_parts(parts: list[str], final: str, interleaved: bool = False) -> None:
    """Splice the per-rank BGZF parts of a multi-GPU `predict --chop` into ONE file in INPUT order (the reference's
    chop streams the FASTQ in order, src/bin/predict.rs:275-316; SURVEY 8e: "the host concatenates per-read results
    in input order").  Every part has a sidecar `<part>.idx.npy` = the file offset after each of its chunks (the writer
    ends a BGZF block there) + the bytes in front of its EOF block.

    interleaved=False: rank r held the r-th contiguous range of the input (byte-range / BGZF-block shards): the parts'
    data follow each other rank by rank.  interleaved=True: chunk g of the input went to rank g % W as its (g // W)-th
    chunk (one gzip stream has one entry point, so chunks are dealt round-robin): the pieces are taken chunk by chunk.
    Either way the result is whole BGZF blocks + one EOF block -- for the dealt form byte-identical to the file one
    GPU writes, because that writer ends its blocks at the same chunk boundaries."""
    idx = [np.load(p + ".idx.npy") for p in parts]
    W = len(parts)
    fhs = [open(p, "rb") for p in parts]

    def copy(out, r, a, b):
        fhs[r].seek(a)
        left = b - a
        while left > 0:
            buf = fhs[r].read(min(left, 8 << 20))
            if not buf:
                raise IOError(f"{parts[r]}: truncated part")
            out.write(buf)
            left -= len(buf)

    try:
        with open(final + ".tmp", "wb") as out:
            if not interleaved:
                for r in range(W):
                    copy(out, r, 0, int(idx[r][-1]))
            else:
                n_chunks = [len(ix) - 1 for ix in idx]
                total = sum(n_chunks)
                if any(n_chunks[r] != (total - r + W - 1) // W for r in range(W)):
                    raise ValueError(f"parts do not form a round-robin deal of {total} chunks over {W} ranks: {n_chunks}")
                for g in range(total):
                    r, j = g % W, g // W
                    copy(out, r, int(idx[r][j - 1]) if j else 0, int(idx[r][j]))
            out.write(BGZF_EOF)
    finally:
        for fh in fhs:
            fh.close()
    os.replace(final + ".tmp", final)
    remove_parts(parts)
