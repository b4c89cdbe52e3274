"""Streaming FASTQ reader (plain / gzip / BGZF) and multithreaded BGZF writer on the native host
library (include/deepchopper_io.h), with the validation rules of `parse_fastq_file`
(deepchopper/data/only_fq.py:21-85): every record needs a name, a sequence and a quality string of
equal length; an empty file is an error.

The reader's unit is a *chunk*: packed `seq` / `qual` bytes + `offsets` -- exactly what
`dc_tokenize_batch` takes -- so a batch goes from file bytes to the GPU without a Python object per
read.  `iter_fastq` keeps the per-record view for small inputs and the chop driver.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterator

import numpy as np

from . import _iolib


def check_compressed_type(path: str) -> str:
    """check_compressed_type (src/output/writefq.rs:84-147): one of _iolib.COMPRESSED_TYPES."""
    t = C.c_int(0)
    _iolib.raise_for(_iolib.lib().dc_io_check_compressed_type(os.fsencode(path), C.byref(t)))
    return _iolib.COMPRESSED_TYPES[t.value]


class FastqChunk:
    """n reads as packed buffers (views into the reader's reusable arrays: copy to keep).

    `_names` / `_name_offsets` hold one string per read: its name (definition line up to the first blank),
    or -- from a reader opened with `full_header=True` -- the whole definition line, with `_name_len[r]`
    giving the length of the name part."""

    __slots__ = ("n", "seq", "qual", "offsets", "_names", "_name_offsets", "_name_len", "sub_sizes")

    def __init__(self, n, seq, qual, offsets, names, name_offsets, name_len=None, sub_sizes=None):
        self.n, self.seq, self.qual, self.offsets = n, seq, qual, offsets
        self._names, self._name_offsets, self._name_len = names, name_offsets, name_len
        self.sub_sizes = sub_sizes  # fused reference batches (`fuse_chunks`): reads per original chunk, in order

    @property
    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets)

    def head(self, k: int) -> "FastqChunk":
        """The first k reads of the chunk."""
        k = max(0, min(int(k), self.n))
        nb, nn = int(self.offsets[k]), int(self._name_offsets[k])
        return FastqChunk(k, self.seq[:nb], self.qual[:nb], self.offsets[: k + 1], self._names[:nn], self._name_offsets[: k + 1],
                          None if self._name_len is None else self._name_len[:k])

    def names(self) -> list[str]:
        return decode_names(self._names, self._name_offsets, self._name_len, self.n)

    def records(self) -> Iterator[tuple[str, bytes, bytes]]:
        s, q, o = self.seq.tobytes(), self.qual.tobytes(), self.offsets
        for i, name in enumerate(self.names()):
            yield name, s[o[i]: o[i + 1]], q[o[i]: o[i + 1]]


def fuse_chunks(chunks, max_reads: int, max_tokens: int, padded: bool = False):
    """Concatenate consecutive chunks (copies) into larger ones of at most `max_reads` reads / `max_tokens` tokens, remembering
    the original sizes in `sub_sizes`.  For the reference-compatible `predict -b 12`: every reader chunk is one batch of the
    reference's DataLoader; fused, many of them share ONE GPU launch while each keeps its own left padding (and its own
    `.pt` file).  A chunk's tokens: its reads' bases + one [SEP] each -- what a launch runs through the GEMMs when the
    [PAD] runs are shared (`dc_forward_shared_prefix`) -- or, with `padded`, reads x (longest read + 1): what the
    left-padded rows hold.  A chunk that alone exceeds the limits is passed through on its own."""
    pend: list[FastqChunk] = []
    n = nt = 0

    def flush():
        nonlocal pend, n, nt
        if not pend:
            return None
        offs = np.zeros(n + 1, np.int64)
        noffs = np.zeros(n + 1, np.int64)
        r = b = hb = 0
        for c in pend:
            offs[r + 1: r + c.n + 1] = c.offsets[1: c.n + 1] + b
            noffs[r + 1: r + c.n + 1] = c._name_offsets[1: c.n + 1] + hb
            r, b, hb = r + c.n, b + int(c.offsets[c.n]), hb + int(c._name_offsets[c.n])
        seq = np.concatenate([c.seq[: int(c.offsets[c.n])] for c in pend])
        qual = np.concatenate([c.qual[: int(c.offsets[c.n])] for c in pend])
        names = np.concatenate([c._names[: int(c._name_offsets[c.n])] for c in pend])
        nlen = None if pend[0]._name_len is None else np.concatenate([c._name_len[: c.n] for c in pend])
        out = FastqChunk(n, seq, qual, offs, names, noffs, nlen, sub_sizes=[c.n for c in pend])
        pend, n, nt = [], 0, 0
        return out

    for c in chunks:
        cb = int(c.offsets[c.n])
        ct = c.n * (int(np.diff(c.offsets[: c.n + 1]).max()) + 1) if padded and c.n else cb + c.n
        if pend and (n + c.n > max_reads or nt + ct > max_tokens):
            yield flush()
        pend.append(FastqChunk(c.n, c.seq[:cb].copy(), c.qual[:cb].copy(), c.offsets[: c.n + 1].copy(),
                               c._names[: int(c._name_offsets[c.n])].copy(), c._name_offsets[: c.n + 1].copy(),
                               None if c._name_len is None else c._name_len[: c.n].copy()))
        n, nt = n + c.n, nt + ct
    last = flush()
    if last is not None:
        yield last


def decode_names(names: np.ndarray, name_offsets: np.ndarray, name_len: np.ndarray | None, n: int) -> list[str]:
    raw = names.tobytes()
    o = name_offsets
    if name_len is None:
        return [raw[o[i]: o[i + 1]].decode("latin1") for i in range(n)]
    return [raw[o[i]: o[i] + int(name_len[i])].decode("latin1") for i in range(n)]


class FastqReader:
    """StreamingFastqReader (src/output/writefq.rs:172-199) + parse_fastq_file's validation.

    `byte_range=(b0, b1)` opens a shard of an uncompressed file: the records whose '@' line starts in
    [b0, b1) -- ranks tile the file with `shard_byte_range` and never parse each other's reads."""

    def __init__(self, path: str, *, byte_range: tuple[int, int] | None = None, readahead: int = 4,
                 full_header: bool = False, strict_four_line: bool = False):
        self.path = str(path)
        self.full_header = bool(full_header)  # chunks carry whole definition lines (the chop stage needs them)
        self._h = C.c_void_p()
        L = _iolib.lib()
        if byte_range is None:
            rc = L.dc_fq_open(os.fsencode(self.path), readahead, C.byref(self._h))
        else:
            rc = L.dc_fq_open_range(os.fsencode(self.path), int(byte_range[0]), int(byte_range[1]), readahead, C.byref(self._h))
        if rc:
            self._h = C.c_void_p()
            _iolib.raise_for(rc)
        if strict_four_line:  # `chop` reads as noodles does; `predict` (the default) as kseq does
            _iolib.raise_for(L.dc_fq_set_strict(self._h, 1))
        self._cap = (0, 0, 0)
        self._bufs = None

    def _buffers(self, max_reads: int, cap_bases: int, cap_names: int):
        if self._cap != (max_reads, cap_bases, cap_names):
            self._bufs = (np.empty(cap_bases, np.uint8), np.empty(cap_bases, np.uint8), np.empty(max_reads + 1, np.int64),
                          np.empty(cap_names, np.uint8), np.empty(max_reads + 1, np.int64), np.empty(max_reads, np.int32))
            self._cap = (max_reads, cap_bases, cap_names)
        return self._bufs

    def read_chunk(self, max_reads: int = 4096, cap_bases: int = 8 << 20, cap_names: int | None = None) -> FastqChunk | None:
        """Up to max_reads records / cap_bases bases; None at end of input.  The returned arrays are
        reused by the next call."""
        if not self._h:
            raise ValueError("reader is closed")
        cap_names = cap_names or max((1024 if self.full_header else 256) * max_reads, 1 << 16)
        seq, qual, offs, names, noffs, nlen = self._buffers(max_reads, cap_bases, cap_names)
        n = C.c_int64(0)
        if self.full_header:
            rc = _iolib.lib().dc_fq_read_chunk_full(self._h, max_reads, seq.ctypes.data, qual.ctypes.data, cap_bases,
                                                    offs.ctypes.data, names.ctypes.data, cap_names, noffs.ctypes.data,
                                                    nlen.ctypes.data, C.byref(n))
        else:
            rc = _iolib.lib().dc_fq_read_chunk(self._h, max_reads, seq.ctypes.data, qual.ctypes.data, cap_bases, offs.ctypes.data,
                                               names.ctypes.data, cap_names, noffs.ctypes.data, C.byref(n))
        _iolib.raise_for(rc)
        k = n.value
        if k == 0:
            return None
        nb = int(offs[k])
        return FastqChunk(k, seq[:nb], qual[:nb], offs[: k + 1], names[: int(noffs[k])], noffs[: k + 1],
                          nlen[:k] if self.full_header else None)

    def chunks(self, max_reads: int = 4096, cap_bases: int = 8 << 20) -> Iterator[FastqChunk]:
        while True:
            c = self.read_chunk(max_reads, cap_bases)
            if c is None:
                return
            yield c

    @property
    def records_read(self) -> int:
        return int(_iolib.lib().dc_fq_records_read(self._h)) if self._h else 0

    @property
    def bytes_consumed(self) -> int:
        return int(_iolib.lib().dc_fq_bytes_consumed(self._h)) if self._h else 0

    @property
    def helper_cpu_seconds(self) -> float:
        """CPU seconds of the reader's finished helper threads (read-ahead, inflate): exact once the stream is exhausted."""
        return float(_iolib.lib().dc_fq_helper_cpu_seconds(self._h)) if self._h else 0.0

    def close(self) -> None:
        if self._h:
            _iolib.lib().dc_fq_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()


def shard_byte_range(path: str, rank: int, world: int) -> tuple[int, int]:
    """Contiguous byte slice of an uncompressed FASTQ for `rank` (contiguous read ranges
    per GPU).  Slices tile [0, size); FastqReader(byte_range=...) snaps them to record boundaries."""
    size = os.path.getsize(path)
    return size * rank // world, size * (rank + 1) // world


def iter_fastq(path: str, max_records: int | None = None) -> Iterator[tuple[str, bytes, bytes]]:
    """Yields (name, seq, qual); name = the definition line up to the first blank (what pyfastx's
    `Fastx` iterator returns as `name`)."""
    n = 0
    with FastqReader(path) as rd:
        for chunk in rd.chunks():
            for rec in chunk.records():
                yield rec
                n += 1
                if max_records is not None and n >= max_records:
                    return
    if n == 0:
        raise ValueError(f"No valid records found in {path}")


class BgzfWriter:
    """bgzf::io::MultithreadedWriter as the reference uses it for FASTQ output
    (src/output/writefq.rs:252-276, src/bin/predict.rs:266-270)."""

    def __init__(self, path: str, threads: int = 2, level: int = 6):
        self._h = C.c_void_p()
        self.mark_offsets = np.zeros(0, np.int64)  # file offset of every mark(), known after close()
        self.data_bytes = 0                         # bytes in front of the EOF block, known after close()
        self.deflate_cpu_s = 0.0                    # CPU seconds of the compression workers, known after close()
        _iolib.raise_for(_iolib.lib().dc_bgzf_open(os.fsencode(str(path)), max(1, int(threads)), level, C.byref(self._h)))

    def write(self, data: bytes) -> None:
        _iolib.raise_for(_iolib.lib().dc_bgzf_write(self._h, data, len(data)))

    def write_record(self, name: bytes, seq: bytes, qual: bytes) -> None:
        self.write(b"@" + name + b"\n" + seq + b"\n+\n" + qual + b"\n")

    def write_chunk(self, chunk: FastqChunk) -> None:
        """Packed records straight from a reader chunk (no per-record Python objects)."""
        _iolib.raise_for(_iolib.lib().dc_bgzf_write_fastq(
            self._h, chunk.n, chunk._names.ctypes.data, chunk._name_offsets.ctypes.data, chunk.seq.ctypes.data,
            chunk.qual.ctypes.data, chunk.offsets.ctypes.data))

    def chop_write_chunk(self, params: "_iolib.ChopParams", n: int, headers, header_offsets, name_len, seq, qual, offsets,
                         pred_len, n_intervals, intervals, n_keep, keep, records_per_read=None) -> int:
        """dc_chop_write_chunk: gate + split + write the records of one chunk; returns the records written.
        All arguments are C-contiguous numpy arrays (intervals int32[n, max_iv, 2], keep int32[n, max_keep, 2])."""
        def ptr(a, dt):
            if a is None:
                return None
            assert a.dtype == dt and a.flags.c_contiguous, (a.dtype, dt)
            return a.ctypes.data

        total = C.c_int64(0)
        _iolib.raise_for(_iolib.lib().dc_chop_write_chunk(
            self._h, C.byref(params), n, ptr(headers, np.uint8), ptr(header_offsets, np.int64), ptr(name_len, np.int32),
            ptr(seq, np.uint8), ptr(qual, np.uint8), ptr(offsets, np.int64), ptr(pred_len, np.int32),
            ptr(n_intervals, np.int32), ptr(intervals, np.int32), intervals.shape[1] if intervals is not None else 0,
            ptr(n_keep, np.int32), ptr(keep, np.int32), keep.shape[1] if keep is not None else 0,
            ptr(records_per_read, np.int32), C.byref(total)))
        return int(total.value)

    def mark(self) -> None:
        """End the current BGZF block here and remember the position (never waits for the compression workers);
        `mark_offsets` holds the file offsets after close()."""
        _iolib.raise_for(_iolib.lib().dc_bgzf_mark(self._h))

    def close(self) -> None:
        if self._h:
            h, self._h = self._h, C.c_void_p()
            n = int(_iolib.lib().dc_bgzf_mark_count(h))
            marks = np.zeros(max(n, 1), np.int64)
            got, data, cpu = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
            _iolib.raise_for(_iolib.lib().dc_bgzf_close2(h, marks.ctypes.data, n, C.byref(got), C.byref(data), C.byref(cpu)))
            self.mark_offsets = marks[: int(got.value)].copy()
            self.data_bytes = int(data.value)
            self.deflate_cpu_s = float(cpu.value)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, *exc):
        self.close()
