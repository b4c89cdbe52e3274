"""Host-side mirror of the reference's PyO3 module `deepchopper.deepchopper`
(src/python.rs:879-958) for the hot-path subset: same function names, argument meaning and error
behaviour, so the parity tests read like the reference's own tests.

Byte / label work runs on the GPU through the C ABI (no CPU fallback); the few table / region
helpers that are O(k) arithmetic on a handful of integers are plain host code.
"""
from __future__ import annotations

import gzip
import itertools
import os
from typing import Sequence

import numpy as np
import torch

from . import _lib, ops

# `default` submodule (src/python.rs:45-53, src/default.rs:1-7)


class default:  # noqa: N801 - mirrors the PyO3 submodule name
    QUAL_OFFSET = 33
    BASES = "ATCGN"
    KMER_SIZE = 3
    VECTORIZED_TARGET = False


__all__ = [
    "default", "Predict", "RecordData",
    "encode_qual", "normalize_seq", "parse_target_from_id", "vectorize_target", "vectorize_targets",
    "splite_qual_by_offsets", "seq_to_kmers", "seq_to_kmers_and_offset", "kmers_to_seq", "kmerids_to_seq",
    "generate_kmers", "generate_kmers_table", "to_kmer_target_region", "to_original_targtet_region",
    "encode_fq_path_to_tensor", "encode_fq_paths_to_tensor", "write_fq", "write_fq_parallel",
    "majority_voting", "get_label_region", "smooth_label_region", "summary_predict",
    "remove_intervals_and_keep_left", "id_list2seq", "load_predicts_from_batch_pt", "load_predicts_from_batch_pts",
    "highlight_targets",
]

MIN_READ_LEN = 150
MIN_CHOPED_SEQ_LEN = 20
IGNORE_LABEL = -100

MSG_SEQ_SHORTER = "The sequence is shorter than the k-mer size"
MSG_TARGET_INVALID = "The target region is invalid"
MSG_INVALID_KMER_ID = "The k-mer id is invalid"


def _b(x) -> bytes:
    return x.encode() if isinstance(x, str) else bytes(x)


# ------------------------------------------------------------------------------------------
# host helpers (K1, K4 and friends)
# ------------------------------------------------------------------------------------------
def generate_kmers(base: str, k: int) -> list[str]:
    """src/python.rs:264-270 -> kmer.rs:221-228 (cartesian order, first base most significant)."""
    return ["".join(p) for p in itertools.product(base, repeat=k)]


def generate_kmers_table(base: str, k: int) -> dict[str, int]:
    """src/python.rs:257-261 -> kmer.rs:213-219."""
    return {km: i for i, km in enumerate(generate_kmers(base, k))}


def seq_to_kmers(seq: str, k: int, overlap: bool) -> list[str]:
    """src/python.rs:243-249 (normalises first, then windows / chunks)."""
    s = normalize_seq(seq, False)
    if overlap:
        return [s[i : i + k] for i in range(max(len(s) - k + 1, 0))]
    return [s[i : i + k] for i in range(0, len(s), k)]


def seq_to_kmers_and_offset(seq: str, kmer_size: int, overlap: bool):
    """src/python.rs:225-235 -> kmer.rs:175-211."""
    if kmer_size == 0 or kmer_size > len(seq):
        raise Exception(MSG_SEQ_SHORTER)
    if overlap:
        return [(seq[i : i + kmer_size], (i, i + kmer_size)) for i in range(len(seq) - kmer_size + 1)]
    return [(seq[i : i + kmer_size], (i, i + kmer_size)) for i in range(0, len(seq) - kmer_size + 1, kmer_size)]


def kmers_to_seq(kmers: Sequence[str]) -> str:
    """src/python.rs:251-255 -> kmer.rs:152-172."""
    if not kmers:
        return ""
    tail = []
    for km in kmers[1:]:
        if not km:
            raise Exception("Invalid kmer")
        tail.append(km[-1])
    return kmers[0] + "".join(tail)


def kmerids_to_seq(kmer_ids: Sequence[int], id2kmer_table: dict[int, str]) -> str:
    """src/python.rs:194-205 -> kmer.rs:79-91."""
    try:
        return kmers_to_seq([id2kmer_table[int(i)] for i in kmer_ids])
    except KeyError:
        raise Exception(MSG_INVALID_KMER_ID) from None


def to_original_targtet_region(start: int, end: int, k: int) -> tuple[int, int]:
    """src/python.rs:207-211 -> kmer.rs:93-105."""
    return (start, end + k - 1) if end > start else (start, end)


def to_kmer_target_region(start: int, end: int, k: int, seq_len: int | None = None) -> tuple[int, int]:
    """src/python.rs:213-222 -> kmer.rs:107-142."""
    if start >= end or k == 0 or (seq_len is not None and end > seq_len):
        raise Exception(MSG_TARGET_INVALID)
    span = end - start
    return (start, start + span - k + 1) if span >= k else (start, end)


def vectorize_target(start: int, end: int, length: int) -> list[int]:
    """kmer.rs:33-46."""
    if start > end or end > length:
        raise Exception(MSG_TARGET_INVALID)
    return [1 if start <= i < end else 0 for i in range(length)]


def vectorize_targets(targets: Sequence[int], length: int) -> list[int]:
    """kmer.rs:51-77."""
    out = [0] * length
    if len(targets) == 0:
        return out
    assert len(targets) % 2 == 0
    for s, e in zip(targets[::2], targets[1::2]):
        s, e = int(s), int(e)
        if s > e or e > length:
            raise Exception(MSG_TARGET_INVALID)
        for i in range(s, e):
            out[i] = 1
    return out


def splite_qual_by_offsets(target: Sequence[int], offsets: Sequence[tuple[int, int]]) -> list[int]:
    """kmer.rs:14-28."""
    return [0 if s == e else sum(target[s:e]) // (e - s) for s, e in offsets]


def _parse_usize(tok: bytes) -> int:
    if not tok or not tok.isdigit() or int(tok) >= 1 << 64:
        raise ValueError(tok)
    return int(tok)


def parse_target_ranges(src: bytes | str) -> list[tuple[int, int]]:
    """fq_encode/triat.rs:14-48 as ranges."""
    s = _b(src)
    if not s:
        return []
    if b"|" not in s:
        return [(0, 0)]
    try:
        out = []
        for part in s.rsplit(b"|", 1)[1].split(b"-"):
            fields = part.split(b":")
            out.append((_parse_usize(fields[0]), _parse_usize(fields[1])))
        return out
    except (ValueError, IndexError):
        return [(0, 0)]


def parse_target_from_id(src: bytes | str) -> list[int]:
    """src/python.rs:277-286: flattened [start, end, ...]."""
    return [v for rng in parse_target_ranges(src) for v in rng]


_ID2BASE = {7: "A", 8: "C", 9: "G", 10: "T", 11: "N"}


def id_list2seq(ids: Sequence[int]) -> str:
    """src/python.rs:810-813 -> smooth/utils.rs:41-46."""
    return "".join(_ID2BASE.get(int(i), "N") for i in ids)


def generate_unmaped_intervals(intervals: Sequence[tuple[int, int]], total_length: int) -> list[tuple[int, int]]:
    """output/split.rs:260-292 (the tail segment ends at total_length-1)."""
    if not intervals:
        return [(0, total_length)]
    out, cur = [], 0
    for s, e in intervals:
        if cur < s:
            out.append((cur, s))
        cur = e
    if cur < total_length - 1:
        out.append((cur, total_length - 1))
    return out


def remove_intervals_and_keep_left(seq: str, intervals: Sequence[tuple[int, int]]):
    """src/python.rs:693-707 -> output/split.rs:295-320."""
    ivs = sorted(((int(s), int(e)) for s, e in intervals), key=lambda r: r[0])
    sel = generate_unmaped_intervals(ivs, len(seq))
    parts = []
    for s, e in sel:
        if s >= len(seq):
            raise Exception(f"The interval is invalid: {s}..{e}")
        parts.append(seq[s:e])
    return parts, sel


# ------------------------------------------------------------------------------------------
# GPU-backed byte / label functions
# ------------------------------------------------------------------------------------------
def encode_qual(qual: str, qual_offset: int = 33) -> list[int]:
    """src/python.rs:25-35."""
    q = np.frombuffer(_b(qual), dtype=np.uint8)
    if q.size == 0:
        return []
    return ops.encode_qual_bytes(ops.to_device(q), qual_offset).cpu().numpy().astype(np.int64).tolist()


def normalize_seq(seq: str, iupac: bool = False) -> str:
    """src/python.rs:272-275."""
    s = np.frombuffer(_b(seq), dtype=np.uint8)
    if s.size == 0:
        return ""
    out, n = ops.normalize_seq_batch(ops.to_device(s), ops.to_device(np.array([0, s.size], np.int64)), iupac)
    return out[: int(n[0])].cpu().numpy().tobytes().decode("latin1")


def _labels_to_device(labels) -> tuple[torch.Tensor, np.ndarray]:
    lab = np.ascontiguousarray(np.asarray(labels, dtype=np.int8))
    if lab.size and not np.all((lab == 0) | (lab == 1)):
        # the reference's HashMap vote is arbitrary on ties between >2 labels; only binary is defined
        raise Exception("majority_voting / label regions are defined for binary labels (0/1) only")
    if lab.size > ops.SM_MAX_LEN:
        raise Exception(f"reads longer than {ops.SM_MAX_LEN} bases are not supported by the smoothing kernel")
    return ops.to_device(lab), lab


def majority_voting(labels: Sequence[int], window_size: int) -> list[int]:
    """src/python.rs:815-818 -> smooth/utils.rs:48-97."""
    if len(labels) == 0:
        return []
    d, lab = _labels_to_device(labels)
    out = ops.smooth_intervals_from_labels(d, ops.to_device(np.array([0, lab.size], np.int64)), window=window_size,
                                           min_interval=0, approved=1, apply_approved_rule=False, max_intervals=1)
    return out["smooth"].cpu().numpy().tolist()


def _regions(labels, window, min_interval, approved, rule) -> list[tuple[int, int]]:
    if len(labels) == 0:
        return []
    d, lab = _labels_to_device(labels)
    offs = ops.to_device(np.array([0, lab.size], np.int64))
    cap = max(approved, 1) if rule else 64
    while True:
        out = ops.smooth_intervals_from_labels(d, offs, window=window, min_interval=min_interval, approved=approved,
                                               apply_approved_rule=rule, max_intervals=cap, want_smooth=False)
        n = int(out["n_intervals"][0].item())
        if n <= cap:
            return [tuple(x) for x in out["intervals"][0, :n].cpu().numpy().tolist()]
        cap = n


def get_label_region(labels: Sequence[int]) -> list[tuple[int, int]]:
    """src/python.rs:666-672 -> utils.rs:671-695."""
    return _regions(labels, 1, 0, 0, False)


def smooth_label_region(labels, smooth_window_size: int, min_interval_size: int, approved_interval_number: int):
    """src/python.rs:674-690 -> utils.rs:699-721."""
    return _regions(labels, smooth_window_size, min_interval_size, approved_interval_number, True)


def summary_predict(predictions, labels, ignore_label: int):
    """src/python.rs:516-523 -> utils.rs:33-55 (ragged lists)."""
    n = len(predictions)
    if n == 0:
        return [], []
    width = max(max((len(p) for p in predictions), default=0), max((len(l) for l in labels), default=0), 1)
    P = np.zeros((n, width), np.int8)
    Lb = np.full((n, width), ignore_label, np.int8)
    for r, (p, l) in enumerate(zip(predictions, labels)):
        m = min(len(p), len(l))
        P[r, :m] = np.asarray(p[:m], dtype=np.int8)
        Lb[r, :m] = np.asarray(l[:m], dtype=np.int8)
    op, ol, offs = ops.summary_predict_dense(ops.to_device(P), ops.to_device(Lb), ignore_label)
    op, ol, offs = op.cpu().numpy(), ol.cpu().numpy(), offs.cpu().numpy()
    return ([op[offs[r] : offs[r + 1]].tolist() for r in range(n)], [ol[offs[r] : offs[r + 1]].tolist() for r in range(n)])


# ------------------------------------------------------------------------------------------
# FASTQ reading (host) and the k-mer tensor encoder
# ------------------------------------------------------------------------------------------
def read_fastq(path: str) -> list[tuple[bytes, bytes, bytes]]:
    """4-line FASTQ (plain or gzip) -> [(name, seq, qual)]; name = definition up to the first
    blank, as noodles' `definition().name()` (tensor_encoder.rs:134-140)."""
    opener = gzip.open if str(path).endswith((".gz", ".bgz")) else open
    recs = []
    with opener(path, "rb") as fh:
        while True:
            head = fh.readline()
            if not head:
                break
            if not head.startswith(b"@"):
                if head.strip():
                    raise ValueError(f"malformed FASTQ record header in {path}: {head[:40]!r}")
                continue
            seq = fh.readline().rstrip(b"\r\n")
            fh.readline()
            qual = fh.readline().rstrip(b"\r\n")
            name = head[1:].rstrip(b"\r\n").split(b" ", 1)[0].split(b"\t", 1)[0]
            recs.append((name, seq, qual))
    return recs


class RecordData:
    """`RecordData(id, seq, qual)` (src/python.rs:93-160; src/fq_encode/record.rs:4-14).  The writers below also take
    plain `(id, seq, qual)` tuples, as the reference's extractor does (src/python.rs:104-116)."""

    __slots__ = ("id", "seq", "qual")

    def __init__(self, id: str, seq: str, qual: str):  # noqa: A002 - the reference's argument name
        self.id, self.seq, self.qual = id, seq, qual

    def __repr__(self) -> str:
        return f"RecordData(id={self.id!r}, seq={self.seq!r}, qual={self.qual!r})"


def _record_bytes(rec) -> tuple[bytes, bytes, bytes]:
    rid, seq, qual = (rec.id, rec.seq, rec.qual) if isinstance(rec, RecordData) else rec
    return _b(rid), _b(seq), _b(qual)


def write_fq(records_data, file_path=None) -> None:
    """src/python.rs:171-178 -> src/output/writefq.rs:17-37: plain 4-line FASTQ ("@id", seq, "+", qual; the definition
    carries no description) to `file_path`, or to stdout when it is None."""
    import sys

    sink = open(file_path, "wb") if file_path is not None else sys.stdout.buffer
    try:
        for rec in records_data:
            rid, seq, qual = _record_bytes(rec)
            sink.write(b"@" + rid + b"\n" + seq + b"\n+\n" + qual + b"\n")
    finally:
        if file_path is not None:
            sink.close()
        else:
            sink.flush()


def write_fq_parallel(records_data, file_path, threads: int) -> None:
    """src/python.rs:180-192 -> write_zip_fq_parallel (src/output/writefq.rs:222-250): the same records through the
    multithreaded BGZF writer (`threads` compression workers, at least one)."""
    from .fastq import BgzfWriter

    with BgzfWriter(str(file_path), threads=max(1, int(threads))) as w:
        for rec in records_data:
            w.write_record(*_record_bytes(rec))


def tensor_encode_records(records, k: int, bases: bytes, qual_offset: int, vectorized_target: bool,
                          max_width: int = 0, max_seq_len: int = 0):
    """TensorEncoder::encode (fq_encode/tensor_encoder.rs:127-327) over parsed records.
    Returns (input i32[N,2,W], target i32[N,T,2|W], qual i32[N,Lmax]) as numpy arrays."""
    bases = _b(bases)
    if not records:
        raise Exception(MSG_SEQ_SHORTER)
    names = [r[0] for r in records]
    seqs = [r[1] for r in records]
    quals = [r[2] for r in records]
    sb, offs = ops.pack_reads(seqs)
    d_seq, d_offs = ops.to_device(sb), ops.to_device(offs)
    # fetch_records: normalised length decides skip / mismatch / max_seq_len (raw bytes are kept)
    _, nlen = ops.normalize_seq_batch(d_seq, d_offs, False)
    nlen = nlen.cpu().numpy()
    kept = []
    for r in range(len(records)):
        if nlen[r] < k:
            continue
        if nlen[r] != len(quals[r]):
            raise Exception(f"record: id {names[r].decode(errors='replace')} seq_len != qual_len")
        max_seq_len = max(max_seq_len, int(nlen[r]))
        kept.append(r)
    if max_seq_len < k:
        raise Exception(MSG_SEQ_SHORTER)
    max_width = max(max_width, max_seq_len - k + 1)

    status = ops.kmer_validate(d_seq, d_offs, k, bases).cpu().numpy()
    rows, targets = [], []
    out_row = np.full(len(records), -1, np.int32)
    for r in kept:
        raw_len = len(seqs[r])
        if max(raw_len - k, 0) + 1 > max_width or status[r] != 0 or len(quals[r]) != raw_len:
            continue
        try:
            kt = [to_kmer_target_region(s, e, k, None) for s, e in parse_target_ranges(names[r])]
        except Exception:
            continue  # encode_target fails -> the record is dropped (tensor_encoder.rs:294-299)
        out_row[r] = len(rows)
        rows.append(r)
        targets.append(kt)
    if not rows:
        raise Exception("Failed to stack inputs")
    if len({len(t) for t in targets}) != 1:
        raise Exception("Failed to stack targets")
    qb, _ = ops.pack_reads(quals)
    inp, qout = ops.kmer_encode(d_seq, ops.to_device(qb), d_offs, ops.to_device(out_row), len(rows), k, bases,
                                qual_offset, max_width, max_seq_len)
    tr = np.asarray(targets, dtype=np.int32).reshape(len(rows), len(targets[0]), 2)
    if vectorized_target:
        tgt = ops.kmer_vectorize_targets(ops.to_device(tr), max_width).cpu().numpy()
    else:
        tgt = tr
    return inp.cpu().numpy(), tgt, qout.cpu().numpy()


def encode_fq_paths_to_tensor(fq_paths, k, bases, qual_offset, vectorized_target, parallel_for_files=False,
                              max_width=None, max_seq_len=None):
    """src/python.rs:288-343: every file is encoded with its own copy of the encoder, then the
    per-file tensors are concatenated (widths must agree, as ndarray::concatenate requires)."""
    parts = [tensor_encode_records(read_fastq(os.fspath(p)), k, _b(bases), qual_offset, vectorized_target,
                                   max_width or 0, max_seq_len or 0) for p in fq_paths]
    try:
        out = tuple(np.concatenate([p[i] for p in parts], axis=0) for i in range(3))
    except ValueError as e:
        raise Exception(f"failed to stack inputs: {e}") from None
    return (*out, generate_kmers_table(_b(bases).decode(), k))


def encode_fq_path_to_tensor(fq_path, k, bases, qual_offset, vectorized_target, max_width=None, max_seq_len=None):
    """src/python.rs:345-398."""
    inp, tgt, qual = tensor_encode_records(read_fastq(os.fspath(fq_path)), k, _b(bases), qual_offset,
                                           vectorized_target, max_width or 0, max_seq_len or 0)
    return inp, tgt, qual, generate_kmers_table(_b(bases).decode(), k)


# ------------------------------------------------------------------------------------------
# Predict + .pt loader (S0)
# ------------------------------------------------------------------------------------------
class Predict:
    """smooth/predict.rs:33-209."""

    def __init__(self, prediction, seq: str, id: str, is_truncated: bool, qual: str | None = None):  # noqa: A002
        self.prediction = list(prediction)
        self.seq = seq
        self.id = id
        self.is_truncated = is_truncated
        self.qual = qual

    def __repr__(self):
        return (f"Predict(prediction: {self.prediction}, seq: {self.seq}, id: {self.id}, "
                f"is_truncated: {self.is_truncated}, qual: {self.qual})")

    def prediction_region(self):
        return get_label_region(self.prediction)

    def smooth_prediction(self, window_size: int):
        return _regions(self.prediction, window_size, 0, 0, False)

    def smooth_label(self, window_size: int):
        return majority_voting(self.prediction, window_size)

    def smooth_and_select_intervals(self, smooth_window_size: int, min_interval_size: int, append_interval_number: int):
        return smooth_label_region(self.prediction, smooth_window_size, min_interval_size, append_interval_number)

    def seq_len(self) -> int:
        return len(self.seq)

    def qual_array(self):
        return [] if self.qual is None else [ord(c) - default.QUAL_OFFSET for c in self.qual]

    def show_info(self, smooth_interval, text_width: int | None = None) -> str:
        """smooth/predict.rs:127-146: id, the raw and the given smoothed regions, and the read with each set highlighted.
        `prediction_region()` prints as Rust's `{:?}` of a `Vec<Range<usize>>` (`[3..9, 20..31]`), `smooth_interval` as a
        `Vec<(usize, usize)>` (`[(3, 9)]`).  Host-only text formatting (no kernel behind it)."""
        regions = self.prediction_region()
        oreg = "[" + ", ".join(f"{s}..{e}" for s, e in regions) + "]"
        sreg = "[" + ", ".join(f"({s}, {e})" for s, e in smooth_interval) + "]"
        return (f"id: {self.id}\nprediction: {oreg}\nsmooth_intervals: {sreg}\n"
                f"{highlight_targets(self.seq, regions, text_width)}\n\n{highlight_targets(self.seq, list(smooth_interval), text_width)}")


def highlight_targets(sequence: str, targets, text_width: int | None = None) -> str:
    """src/vis.rs:5-23 (PyO3 export `highlight_targets`): the target ranges of `sequence` in bold magenta, wrapped to
    `text_width` (80) columns.  The reference builds it from two third-party crates that no test of its own pins: `colored`
    (`.magenta().bold()` = `ESC[1;35m` ... `ESC[0m`) and `textwrap::wrap`, which measures display width -- escape sequences
    count nothing -- and breaks a word longer than the line at the width.  A read has no blanks, so it is one such word: lines
    of exactly `text_width` visible characters, escape sequences carried inline.  Restated from those published behaviours;
    **parity unpinned** (the reference's `test_highlight` only prints)."""
    width = 80 if text_width is None else int(text_width)
    if width < 1:
        raise ValueError("text_width must be positive")
    on, off = "\x1b[1;35m", "\x1b[0m"
    marks: dict[int, str] = {}  # escape sequences in front of visible position p (they take no room on a line)
    last = 0
    for s, e in targets:
        if s < last or e < s or e > len(sequence):
            raise IndexError("targets must be ascending, disjoint ranges inside the sequence")  # (Rust: a slice panic)
        if e > s:
            marks[s] = marks.get(s, "") + on
            marks[e] = marks.get(e, "") + off
        last = e
    out, col = [], 0
    for p in range(len(sequence) + 1):
        if p in marks:          # zero width: stays on the line it is met on
            out.append(marks[p])
        if p < len(sequence):
            if col == width:    # the line is full and another visible character comes: break in front of it
                out.append("\n")
                col = 0
            out.append(sequence[p])
            col += 1
    return "".join(out)


def predicts_from_batch(batch: dict, ignore_label: int = IGNORE_LABEL) -> dict[str, Predict]:
    """Body of load_predicts_from_batch_pt (smooth/predict.rs:264-317) on an in-memory batch
    (tensors or arrays with keys prediction, target, seq, id): argmax + mask on the GPU."""
    logits = torch.as_tensor(np.asarray(batch["prediction"]), dtype=torch.float32)
    target = np.asarray(batch["target"]).astype(np.int64)
    seq = np.asarray(batch["seq"]).astype(np.int64)
    idm = np.asarray(batch["id"]).astype(np.int64)
    B, L, _ = logits.shape
    keepmask = target != ignore_label
    # general masks (not only "left pad + SEP"): argmax on the device, compaction via summary_predict
    dev_logits = ops.to_device(logits.numpy())
    first = np.zeros(B, np.int32)
    length = np.full(B, L, np.int32)
    out = ops.smooth_intervals_from_logits(dev_logits, ops.to_device(first), ops.to_device(length), window=1,
                                           min_interval=0, approved=1, max_intervals=1, want_labels=True)
    pred = out["pred"].view(B, L)
    lab8 = np.where(keepmask, 0, ignore_label).astype(np.int8)
    op, _, offs = ops.summary_predict_dense(pred.contiguous(), ops.to_device(lab8), ignore_label)
    op, offs = op.cpu().numpy(), offs.cpu().numpy()
    res = {}
    for b in range(B):
        n = int(idm[b, 0])
        rid = "".join(chr(int(c)) for c in idm[b, 2 : 2 + n])
        res[rid] = Predict(op[offs[b] : offs[b + 1]].tolist(), id_list2seq(seq[b][keepmask[b]]), rid, bool(idm[b, 1] != 0))
    return res


def load_predicts_from_batch_pt(pt_path, ignore_label: int = IGNORE_LABEL) -> dict[str, Predict]:
    """src/smooth/predict.rs:263-317."""
    d = torch.load(os.fspath(pt_path), map_location="cpu", weights_only=True)
    return predicts_from_batch({k: v.numpy() for k, v in d.items()}, ignore_label)


def load_predicts_from_batch_pts(pt_path, ignore_label: int = IGNORE_LABEL, max_predicts: int | None = None):
    """src/smooth/predict.rs:211-261."""
    files = [os.fspath(pt_path)] if os.path.isfile(pt_path) and os.fspath(pt_path).endswith(".pt") else []
    for root, _, names in os.walk(os.fspath(pt_path)):
        files += [os.path.join(root, n) for n in names if n.endswith(".pt")]
    if max_predicts is not None:
        files = files[:max_predicts]
    out: dict[str, Predict] = {}
    for f in files:
        try:
            out.update(load_predicts_from_batch_pt(f, ignore_label))
        except Exception as e:  # the reference prints and skips unreadable files
            print(f"load pt {f} fail caused by Error: {e!r}")
    return out


class LabelStore:
    """What `deepchopper chop` needs of a set of `.pt` prediction files (load_predicts_from_batch_pts,
    src/smooth/predict.rs:211-317), packed: per read its id, truncation flag and argmax labels over its unmasked positions
    (`Predict.prediction`; `Predict.seq_len()` = their count).  Read natively on a pool of threads (libdeepchopper_io:
    dc_pt_load); a file that reader does not recognise goes through `load_predicts_from_batch_pt` (torch.load) instead, with
    the reference's "load pt ... fail" line if that fails too.  `index`: id -> row, a later file winning as in the
    reference's HashMap."""

    def __init__(self, labels: np.ndarray, offsets: np.ndarray, ids: list[str], truncated: np.ndarray, fallback_files=()):
        self.labels, self.offsets, self.ids, self.truncated = labels, offsets, ids, truncated
        self.index = {rid: i for i, rid in enumerate(ids)}
        self.fallback_files = list(fallback_files)  # files the native reader did not recognise (read through torch.load, or skipped)

    def __len__(self) -> int:
        return len(self.index)

    def predict(self, rid: str) -> "Predict | None":
        """The reference's `Predict` of one read (without its letters: the chop driver takes those from the FASTQ record)."""
        i = self.index.get(rid)
        if i is None:
            return None
        return Predict(self.labels[self.offsets[i]: self.offsets[i + 1]].tolist(), "", rid, bool(self.truncated[i]))

    @staticmethod
    def load(pt_paths, ignore_label: int = IGNORE_LABEL, max_predicts: int | None = None, threads: int = 4) -> "LabelStore":
        import ctypes as C

        from . import _iolib

        files = []
        for d in ([pt_paths] if isinstance(pt_paths, (str, os.PathLike)) else pt_paths):
            found = [os.fspath(d)] if os.path.isfile(d) and os.fspath(d).endswith(".pt") else []  # (WalkDir yields a file root itself)
            for root, _, names in os.walk(os.fspath(d)):
                found += [os.path.join(root, n) for n in names if n.endswith(".pt")]
            files += found[:max_predicts] if max_predicts is not None else found
        arr = (C.c_char_p * len(files))(*[os.fsencode(f) for f in files])
        h = C.c_void_p()
        n, lb, nb, nf = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _iolib.raise_for(_iolib.lib().dc_pt_load(arr, len(files), int(max(1, threads)), int(ignore_label), C.byref(h), C.byref(n),
                                                 C.byref(lb), C.byref(nb), C.byref(nf)))
        try:
            labels = np.empty(lb.value, np.int8)
            offs = np.empty(n.value + 1, np.int64)
            names = np.empty(max(nb.value, 1), np.uint8)
            noffs = np.empty(n.value + 1, np.int64)
            trunc = np.empty(max(n.value, 1), np.uint8)
            failed = np.zeros(max(len(files), 1), np.uint8)
            _iolib.raise_for(_iolib.lib().dc_pt_export(h, labels.ctypes.data, offs.ctypes.data, names.ctypes.data, noffs.ctypes.data,
                                                       trunc.ctypes.data, None, failed.ctypes.data))
        finally:
            _iolib.lib().dc_pt_free(h)
        text = names[: nb.value].tobytes().decode("latin1")
        ids = [text[noffs[i]: noffs[i + 1]] for i in range(n.value)]
        trunc = trunc[: n.value].astype(bool)
        fallback = [files[i] for i in np.flatnonzero(failed[: len(files)])] if nf.value else []
        if fallback:  # the general path for what the native reader left alone
            extra_l, extra_ids, extra_t = [labels], list(ids), [trunc]
            for f in fallback:
                try:
                    for rid, p in load_predicts_from_batch_pt(f, ignore_label).items():
                        extra_l.append(np.asarray(p.prediction, np.int8))
                        extra_ids.append(rid)
                        extra_t.append(np.array([p.is_truncated]))
                except Exception as e:  # the reference prints and skips unreadable files
                    print(f"load pt {f} fail caused by Error: {e!r}")
            sizes = [int(offs[-1])] + [len(x) for x in extra_l[1:]]
            labels = np.concatenate(extra_l) if len(extra_l) > 1 else labels
            offs = np.concatenate([offs, offs[-1] + np.cumsum(sizes[1:], dtype=np.int64)]) if len(sizes) > 1 else offs
            ids, trunc = extra_ids, np.concatenate(extra_t)
        return LabelStore(labels, offs, ids, trunc, fallback)
