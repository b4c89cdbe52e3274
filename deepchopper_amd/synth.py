"""Deterministic synthetic inputs (SURVEY 8d / BASELINE.md section 2).

PRNG = splitmix64 (vectorised), seed 42 for reads, 43 for label streams.  Bases iid uniform
over ACGT with 0.1 % N, qualities iid uniform Phred 1..40 (ASCII 34..73), ids `read_{i:09d}`;
half the reads carry a planted "adapter" (poly-A(20) + a fixed 60-mer) at a random interior
position.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
ADAPTER_60 = b"CTGTAGGCACCATCAATGATCGTCGGACTGTAGAACTCTGAACGTGTAGATCTCGGTGGT"
PLANT = b"A" * 20 + ADAPTER_60


def splitmix64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n outputs of splitmix64 started at `seed` (+ a stream offset so that independent draws do
    not overlap).  Vectorised: output i is the mix of state seed + (i+1) * golden."""
    with np.errstate(over="ignore"):
        golden = np.uint64(0x9E3779B97F4A7C15)
        base = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(stream) * np.uint64(0xD1342543DE82EF95)
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * golden)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    return (splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def synth_reads_fixed(n_reads: int, length: int, seed: int = 42, plant: bool = True, start_index: int = 0):
    """Fixed-length reads as dense uint8 arrays: (seq[n, L], qual[n, L], ids list)."""
    total = n_reads * length
    r = splitmix64(seed, total, stream=2 * start_index + 1)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[(r & np.uint64(3)).astype(np.int64)]
    is_n = ((r >> np.uint64(8)) % np.uint64(1000)) == 0
    bases = np.where(is_n, np.uint8(ord("N")), bases).reshape(n_reads, length)
    q = splitmix64(seed, total, stream=2 * start_index + 2)
    qual = (34 + (q % np.uint64(40)).astype(np.int64)).astype(np.uint8).reshape(n_reads, length)
    if plant and length >= len(PLANT) + 2:
        pos = splitmix64(seed, n_reads, stream=2 * start_index + 3)
        pl = np.frombuffer(PLANT, dtype=np.uint8)
        for i in range(0, n_reads, 2):  # every other read
            p = 1 + int(pos[i] % np.uint64(length - len(PLANT) - 1))
            bases[i, p : p + len(PLANT)] = pl
    ids = [f"read_{start_index + i:09d}" for i in range(n_reads)]
    return bases, qual, ids


def synth_fastq_fixed_fast(n_reads: int, length: int, seed: int = 42, start_index: int = 0, plant: bool = True) -> np.ndarray:
    """FASTQ text (a flat uint8 array) of `n_reads` fixed-length reads, for the file legs of bench.py that need 10^6 reads in
    seconds: the same distributions as `synth_reads_fixed` (bases iid uniform ACGT + 0.1 % N, Phred 1..40, ids
    `read_{i:09d}`, PLANT in every other read) drawn more cheaply -- one splitmix64 output feeds eight bases (as
    `synth_nanopore_like` does), one 16-bit piece gives a quality (`v * 40 >> 16`) and the N mask (`v & 1023 == 0`), and the
    records are assembled as one [n, record] byte matrix.  NOT the same bytes as `synth_reads_fixed` for a given seed.
    numpy releases the GIL in all of it: callers generate blocks on a thread pool."""
    total = n_reads * length
    r = splitmix64(seed, (total + 7) // 8, stream=100 + 2 * start_index).view(np.uint8)[:total]
    v = splitmix64(seed, (total + 3) // 4, stream=101 + 2 * start_index).view(np.uint16)[:total]
    rec = 1 + 14 + 1 + length + 3 + length + 1
    out = np.empty((n_reads, rec), np.uint8)
    out[:, 0] = ord("@")
    out[:, 1:6] = np.frombuffer(b"read_", np.uint8)
    idx = np.arange(start_index, start_index + n_reads, dtype=np.int64)
    for d in range(9):
        out[:, 6 + d] = 48 + (idx // 10 ** (8 - d)) % 10
    out[:, 15] = 10
    seq = out[:, 16: 16 + length]
    seq[:] = np.frombuffer(b"ACGT" * 64, np.uint8)[r].reshape(n_reads, length)
    seq[((v & np.uint16(1023)) == 0).reshape(n_reads, length)] = ord("N")  # 1 / 1024
    out[:, 16 + length: 19 + length] = np.frombuffer(b"\n+\n", np.uint8)
    out[:, 19 + length: 19 + 2 * length] = (34 + ((v.astype(np.uint32) * np.uint32(40)) >> np.uint32(16))).astype(np.uint8).reshape(
        n_reads, length)
    out[:, rec - 1] = 10
    if plant and length >= len(PLANT) + 2:
        rows = np.arange(0, n_reads, 2)
        pos = 1 + (splitmix64(seed, n_reads, stream=102 + 2 * start_index)[rows] % np.uint64(length - len(PLANT) - 1)).astype(np.int64)
        seq[rows[:, None], pos[:, None] + np.arange(len(PLANT))] = np.frombuffer(PLANT, np.uint8)
    return out.reshape(-1)


def write_fastq_fixed_fast(path: str, n_reads: int, length: int, seed: int = 42, block: int = 20_000, threads: int = 8) -> int:
    """`n_reads` reads of `synth_fastq_fixed_fast` into `path`, blocks generated on `threads` threads and written in order.
    Returns the bytes written."""
    from concurrent.futures import ThreadPoolExecutor

    starts = list(range(0, n_reads, block))
    written = 0
    with ThreadPoolExecutor(max(1, threads)) as ex, open(path, "wb") as fh:
        for w0 in range(0, len(starts), 2 * threads):  # bounded look-ahead: 2 blocks per thread in memory
            futs = [ex.submit(synth_fastq_fixed_fast, min(block, n_reads - b0), length, seed, b0) for b0 in starts[w0: w0 + 2 * threads]]
            for f in futs:
                buf = f.result()
                fh.write(memoryview(buf))
                written += buf.size
    return written


def synth_reads_mixed(n_reads: int, min_len: int = 100, max_len: int = 10000, seed: int = 42):
    """Log-uniform lengths in [min_len, max_len] (config C4).  Returns lists of bytes."""
    u = uniform01(seed, n_reads, stream=7)
    lens = np.exp(np.log(min_len) + u * (np.log(max_len) - np.log(min_len))).astype(np.int64)
    lens = np.clip(lens, min_len, max_len)
    seqs, quals, ids = [], [], []
    for i, L in enumerate(lens.tolist()):
        s, q, d = synth_reads_fixed(1, L, seed=seed + 1000 + i, plant=(i % 2 == 0))
        seqs.append(s[0].tobytes())
        quals.append(q[0].tobytes())
        ids.append(f"read_{i:09d}")
    return seqs, quals, ids


def synth_label_stream(n: int, seed: int = 43, noise: float = 0.02, run_every: int = 400, mean_run: float = 60.0,
                       stream: int = 0) -> np.ndarray:
    """Binary label stream: Bernoulli(noise) flips plus planted runs of ones with geometric length
    (mean `mean_run`), roughly one run per `run_every` positions."""
    u = uniform01(seed, n, stream=3 * stream + 11)
    lab = (u < noise).astype(np.int8)
    n_runs = max(n // run_every, 1)
    starts = (uniform01(seed, n_runs, stream=3 * stream + 12) * n).astype(np.int64)
    g = uniform01(seed, n_runs, stream=3 * stream + 13)
    lens = np.maximum(1, np.floor(np.log(1.0 - g) / np.log(1.0 - 1.0 / mean_run))).astype(np.int64)
    for s, l in zip(starts.tolist(), lens.tolist()):
        lab[s : min(n, s + l)] = 1
    # flip a little noise inside the runs too
    flip = uniform01(seed, n, stream=3 * stream + 14) < noise
    lab = np.where(flip, 1 - lab, lab).astype(np.int8)
    return lab


def to_fastq(seqs, quals, ids) -> bytes:
    out = []
    for s, q, i in zip(seqs, quals, ids):
        sb = s.tobytes() if hasattr(s, "tobytes") else bytes(s)
        qb = q.tobytes() if hasattr(q, "tobytes") else bytes(q)
        out.append(b"@" + i.encode() + b"\n" + sb + b"\n+\n" + qb + b"\n")
    return b"".join(out)


def synth_nanopore_like(n_reads: int, seed: int = 42, start_index: int = 0, median_len: float = 900.0,
                        sigma: float = 0.75, min_len: int = 100, max_len: int = 30000, gate: bytes = b"N"):
    """Config C5 stand-in: a direct-RNA-like length mix (log-normal around `median_len`, clipped) with ONT-style
    definition lines (`@id runid=... ch=... start_time=...`) and planted adapters written as runs of the `gate`
    base -- the base `weights.token_gated_state_dict` labels 1 -- so that predict -> chop really chops:
    60 % of the reads carry a 3' terminal adapter (40-80 bases at the very end), 6 % an internal one
    (chimeric read), 2 % five internal ones (more than --mpi 4: passed through).  Bases iid uniform over ACGT (no natural `gate`
    bases), qualities Phred 1..40.  Returns (headers, seqs, quals) as lists of bytes."""
    u = uniform01(seed, n_reads, stream=20 + 4 * start_index)
    z = np.sqrt(-2.0 * np.log(np.maximum(u, 1e-300))) * np.cos(2 * np.pi * uniform01(seed, n_reads, stream=21 + 4 * start_index))
    lens = np.clip(np.exp(np.log(median_len) + sigma * z), min_len, max_len).astype(np.int64)
    offs = np.zeros(n_reads + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    total = int(offs[-1])
    # one splitmix64 output feeds eight bases: byte b -> base b & 3, Phred 1 + (b >> 2) % 40
    r = splitmix64(seed, (total + 7) // 8, stream=22 + 4 * start_index).view(np.uint8)[:total]
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[r & np.uint8(3)]
    qual = (np.uint8(34) + (r >> np.uint8(2)) % np.uint8(40)).astype(np.uint8)
    kind = uniform01(seed, n_reads, stream=23 + 4 * start_index)
    pick = splitmix64(seed, n_reads, stream=24 + 4 * start_index)
    g = np.uint8(gate[0])
    headers, seqs, quals = [], [], []
    for i in range(n_reads):
        a, L = int(offs[i]), int(lens[i])
        k, p = kind[i], int(pick[i] & np.uint64(0x7FFFFFFF))
        alen = 40 + p % 41
        if k < 0.60 and L >= alen + 60:
            bases[a + L - alen: a + L] = g
        elif k < 0.66 and L >= alen + 200:
            s = 60 + (p >> 8) % (L - alen - 120)
            bases[a + s: a + s + alen] = g
        elif k < 0.68 and L >= 700:
            for j in range(5):
                s = 40 + j * (L - 80) // 5
                bases[a + s: a + s + 30] = g
        gi = start_index + i
        headers.append(b"read_%09d runid=%08x ch=%d start_time=2026-01-01T00:%02d:%02dZ" % (gi, p, 1 + p % 512, (gi // 60) % 60, gi % 60))
        seqs.append(bases[a: a + L].tobytes())
        quals.append(qual[a: a + L].tobytes())
    return headers, seqs, quals


def fastq_text(headers, seqs, quals) -> bytes:
    """FASTQ text of records whose definition lines are given whole (bytes, without the '@')."""
    return b"".join(b"@" + h + b"\n" + s + b"\n+\n" + q + b"\n" for h, s, q in zip(headers, seqs, quals))
