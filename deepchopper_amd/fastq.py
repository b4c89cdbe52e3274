"""Streaming FASTQ reader (plain / gzip) with the validation rules of `parse_fastq_file`
(deepchopper/data/only_fq.py:21-85): every record needs a name, a sequence and a quality string of
equal length; an empty file is an error."""
from __future__ import annotations

import gzip
from typing import Iterator


def _open(path: str):
    with open(path, "rb") as fh:
        magic = fh.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


def iter_fastq(path: str, max_records: int | None = None) -> Iterator[tuple[str, bytes, bytes]]:
    """Yields (name, seq, qual); name = the definition line up to the first blank (what pyfastx's
    `Fastx` iterator returns as `name`)."""
    n = 0
    with _open(path) as fh:
        while True:
            head = fh.readline()
            if not head:
                break
            if not head.strip():
                continue
            if not head.startswith(b"@"):
                raise RuntimeError(f"FASTQ parsing error in {path}: record {n} does not start with '@'")
            seq = fh.readline().rstrip(b"\r\n")
            plus = fh.readline()
            qual = fh.readline().rstrip(b"\r\n")
            if not plus.startswith(b"+"):
                raise RuntimeError(f"FASTQ parsing error in {path}: record {n} has no '+' separator")
            name = head[1:].rstrip(b"\r\n").split(b" ", 1)[0].split(b"\t", 1)[0].decode("latin1")
            if not name or not seq or not qual:
                raise ValueError(f"Incomplete FASTQ record at position {n} in {path}")
            if len(seq) != len(qual):
                raise ValueError(f"Sequence/quality length mismatch in record '{name}': seq={len(seq)}, qual={len(qual)}")
            yield name, seq, qual
            n += 1
            if max_records is not None and n >= max_records:
                break
    if n == 0:
        raise ValueError(f"No valid records found in {path}")
