"""The reader's own DEFLATE / gzip decoder (csrc/fast_inflate.h) against zlib: every block type, long matches,
sub-table codes, multi-member streams, block-boundary crossings, truncation and corruption."""
import gzip
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from deepchopper_amd.fastq import BgzfWriter, FastqReader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fastq(rng, n, kind):
    recs = []
    for i in range(n):
        L = int(rng.integers(1, 3000))
        if kind == "random":
            s = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L))
            q = bytes(rng.integers(33, 127, L).astype(np.uint8))
        elif kind == "repetitive":  # long matches, distances of every size, run lengths (distance 1)
            unit = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), int(rng.integers(1, 40))))
            s = (unit * (L // len(unit) + 1))[:L]
            q = (b"I" * 7 + b"#") * (L // 8 + 1)
            q = q[:L]
        else:  # skewed alphabet: short and very long Huffman codes side by side
            s = bytes(rng.choice(np.frombuffer(b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAACGT", np.uint8), L))
            p = np.array([0.5 ** (k + 1) for k in range(60)])
            q = bytes((33 + rng.choice(60, L, p=p / p.sum())).astype(np.uint8))
        recs.append((b"r%d_%s" % (i, kind.encode()), s, q))
    return recs


def _text(recs):
    return b"".join(b"@" + n + b"\n" + s + b"\n+\n" + q + b"\n" for n, s, q in recs)


def _read_all(path):
    out = []
    with FastqReader(path, full_header=True) as rd:
        for c in rd.chunks(1000, 4 << 20):
            out.append((c._names.tobytes(), c.seq.tobytes(), c.qual.tobytes(), c.offsets.tolist()))
    return out


def _deflate(data, level, strategy, mem=8):
    co = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
    return co.compress(data) + co.flush()


@pytest.mark.parametrize("kind", ["random", "repetitive", "skewed"])
def test_reader_matches_zlib_on_every_block_type(tmp_path, kind, monkeypatch):
    rng = np.random.default_rng(hash(kind) % 1000)
    recs = _fastq(rng, 1500, kind)
    text = _text(recs)
    want_seq = b"".join(s for _, s, _ in recs)
    variants = {
        "stored": _deflate(text, 0, zlib.Z_DEFAULT_STRATEGY),
        "fixed": _deflate(text, 6, zlib.Z_FIXED),
        "dynamic1": _deflate(text, 1, zlib.Z_DEFAULT_STRATEGY),
        "dynamic9": _deflate(text, 9, zlib.Z_DEFAULT_STRATEGY),
        "huffman_only": _deflate(text, 6, zlib.Z_HUFFMAN_ONLY),
        "rle": _deflate(text, 6, zlib.Z_RLE),
        "filtered_mem1": _deflate(text, 4, zlib.Z_FILTERED, mem=1),
    }
    # multi-member (incl. an empty member and trailing zero padding)
    third = len(recs) // 3
    parts = [_text(recs[:third]), b"", _text(recs[third: 2 * third]), _text(recs[2 * third:])]
    variants["multi_member"] = b"".join(gzip.compress(p, 6) for p in parts) + b"\0" * 37
    for name, blob in variants.items():
        path = str(tmp_path / f"{kind}_{name}.fq.gz")
        open(path, "wb").write(blob)
        monkeypatch.delenv("DC_IO_INFLATE", raising=False)
        got = _read_all(path)
        assert b"".join(g[1] for g in got) == want_seq, name
        assert sum(len(g[3]) - 1 for g in got) == len(recs), name
        monkeypatch.setenv("DC_IO_INFLATE", "zlib")
        assert _read_all(path) == got, name
    monkeypatch.delenv("DC_IO_INFLATE", raising=False)


def test_large_stream_crosses_output_blocks_and_bgzf(tmp_path):
    rng = np.random.default_rng(5)
    recs = _fastq(rng, 9000, "random") + _fastq(rng, 2000, "repetitive")  # ~ 33 MB of text: several 4 MiB blocks
    text = _text(recs)
    p1 = str(tmp_path / "big.fq.gz")
    open(p1, "wb").write(gzip.compress(text, 1))
    p2 = str(tmp_path / "big.bgzf.fq.gz")
    with BgzfWriter(p2, threads=3) as w:
        w.write(text)
    want = b"".join(s for _, s, _ in recs), b"".join(q for _, _, q in recs)
    for p in (p1, p2):
        got = _read_all(p)
        assert b"".join(g[1] for g in got) == want[0] and b"".join(g[2] for g in got) == want[1]
    # BGZF byte-range shards still tile the file exactly once with the new decoder
    size = os.path.getsize(p2)
    n = 0
    for r in range(5):
        with FastqReader(p2, byte_range=(size * r // 5, size * (r + 1) // 5)) as rd:
            n += sum(c.n for c in rd.chunks())
    assert n == len(recs)


def test_corrupt_and_truncated_streams_fail_loudly(tmp_path):
    rng = np.random.default_rng(6)
    text = _text(_fastq(rng, 400, "random"))
    blob = gzip.compress(text, 6)
    cases = {"truncated": blob[: len(blob) // 2], "no_trailer": blob[:-5]}
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x55
    cases["bit_flip"] = bytes(bad)
    crc = bytearray(blob)
    crc[-6] ^= 1
    cases["bad_crc"] = bytes(crc)
    for name, b in cases.items():
        p = str(tmp_path / f"{name}.fq.gz")
        open(p, "wb").write(b)
        with pytest.raises((RuntimeError, ValueError)):
            _read_all(p)


def test_inflate_rate_report(tmp_path, capsys):
    """Not a pass/fail criterion: prints the decoder's rate next to zlib's on this host."""
    import time

    rng = np.random.default_rng(7)
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), (20000, 1000))
    qual = rng.integers(34, 74, (20000, 1000)).astype(np.uint8)
    text = b"".join(b"@read_%09d\n" % i + seq[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n" for i in range(20000))
    p = str(tmp_path / "r.fq.gz")
    open(p, "wb").write(gzip.compress(text, 1))
    rates = {}
    for mode in ("fast", "zlib"):
        env = dict(os.environ, PYTHONPATH=ROOT)
        if mode == "zlib":
            env["DC_IO_INFLATE"] = "zlib"
        code = ("import time,sys\nfrom deepchopper_amd.fastq import FastqReader\nt=time.time();n=0\n"
                f"rd=FastqReader({p!r})\nfor c in rd.chunks(4096, 8<<20): n+=c.n\nprint(n/(time.time()-t))")
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        rates[mode] = float(out.stdout.strip())
    with capsys.disabled():
        print(f"\n  gzip parse rate, 1 kb reads: in-house inflate {rates['fast']:.0f} reads/s, zlib {rates['zlib']:.0f} reads/s")
    assert rates["fast"] > 0 and rates["zlib"] > 0


# ---- parallel inflate of one large member (fastq_io.cpp ParInflate): same bytes as the serial decode ----
def _par_env(monkeypatch, threads, chunk):
    monkeypatch.delenv("DC_IO_INFLATE", raising=False)
    monkeypatch.setenv("DC_IO_INFLATE_THREADS", str(threads))
    monkeypatch.setenv("DC_IO_INFLATE_CHUNK", str(chunk))


def _sync_flushed(text, level, every, mode):
    """pigz-like: one member, the compressor flushed (empty stored block) every `every` bytes."""
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    out = []
    for b0 in range(0, len(text), every):
        out.append(co.compress(text[b0: b0 + every]))
        out.append(co.flush(mode))
    out.append(co.flush())
    return b"".join(out)


@pytest.mark.parametrize("kind", ["random", "repetitive", "skewed"])
def test_parallel_inflate_matches_serial_on_every_block_type(tmp_path, kind, monkeypatch):
    rng = np.random.default_rng(100 + len(kind))
    recs = _fastq(rng, 2500, kind)
    text = _text(recs)
    variants = {
        "stored": _deflate(text, 0, zlib.Z_DEFAULT_STRATEGY),           # no dynamic block anywhere: serial stretches only
        "fixed": _deflate(text, 6, zlib.Z_FIXED),
        "dynamic1": _deflate(text, 1, zlib.Z_DEFAULT_STRATEGY),
        "dynamic6": _deflate(text, 6, zlib.Z_DEFAULT_STRATEGY),
        "dynamic9_mem1": _deflate(text, 9, zlib.Z_DEFAULT_STRATEGY, mem=1),  # a block every ~250 symbols
        "huffman_only": _deflate(text, 6, zlib.Z_HUFFMAN_ONLY),
        "sync_flush": _sync_flushed(text, 6, 50_000, zlib.Z_SYNC_FLUSH),
        "full_flush": _sync_flushed(text, 4, 131_072, zlib.Z_FULL_FLUSH),
    }
    third = len(recs) // 3
    parts = [_text(recs[:third]), b"", _text(recs[third: 2 * third]), _text(recs[2 * third:])]
    variants["multi_member"] = b"".join(gzip.compress(p, 6) for p in parts) + b"\0" * 11
    for name, blob in variants.items():
        path = str(tmp_path / f"{kind}_{name}.fq.gz")
        open(path, "wb").write(blob)
        _par_env(monkeypatch, 1, 16384)
        want = _read_all(path)
        assert sum(len(g[3]) - 1 for g in want) == len(recs), name
        for threads, chunk in ((3, 16384), (4, 4096), (2, 100_000)):
            _par_env(monkeypatch, threads, chunk)
            assert _read_all(path) == want, (name, threads, chunk)


def test_parallel_inflate_non_text_literals_and_errors(tmp_path, monkeypatch):
    rng = np.random.default_rng(11)
    # header comments with bytes outside the text range: blocks holding them fail the block-start probe, so the
    # coordinator has to bridge them serially; the result must not change
    recs = []
    for i in range(3000):
        L = int(rng.integers(200, 1500))
        junk = bytes(rng.integers(128, 256, int(rng.integers(0, 40))).astype(np.uint8)) if i % 3 == 0 else b""
        recs.append((b"r%d " % i + junk, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L)),
                     bytes(rng.integers(33, 127, L).astype(np.uint8))))
    text = _text(recs)
    blob = gzip.compress(text, 6)
    p = str(tmp_path / "junk.fq.gz")
    open(p, "wb").write(blob)
    _par_env(monkeypatch, 1, 16384)
    want = _read_all(p)
    _par_env(monkeypatch, 4, 16384)
    assert _read_all(p) == want
    # corruption and truncation are still reported in parallel mode
    cases = {"truncated": blob[: len(blob) // 2], "no_trailer": blob[:-5]}
    for frac in (0.1, 0.5, 0.9):
        bad = bytearray(blob)
        bad[int(len(bad) * frac)] ^= 0x55
        cases[f"bit_flip_{frac}"] = bytes(bad)
    crc = bytearray(blob)
    crc[-6] ^= 1
    cases["bad_crc"] = bytes(crc)
    isz = bytearray(blob)
    isz[-2] ^= 1
    cases["bad_isize"] = bytes(isz)
    for name, b in cases.items():
        pb = str(tmp_path / f"{name}.fq.gz")
        open(pb, "wb").write(b)
        with pytest.raises((RuntimeError, ValueError)):
            _read_all(pb)


def test_parallel_inflate_huge_expansion_is_bounded(tmp_path, monkeypatch):
    """A few hundred KB that inflate to > 100 MB: workers stop at their symbol budget, the rest streams serially."""
    unit = b"@r\n" + b"ACGT" * 250 + b"\n+\n" + b"I" * 1000 + b"\n"
    n = 60_000
    p = str(tmp_path / "rep.fq.gz")
    with gzip.open(p, "wb", compresslevel=6) as fh:
        for _ in range(60):
            fh.write(unit * (n // 60))
    assert os.path.getsize(p) < (2 << 20)
    _par_env(monkeypatch, 3, 16384)
    got = 0
    with FastqReader(p) as rd:
        for c in rd.chunks(8192, 16 << 20):
            got += c.n
            assert c.seq[:8].tobytes() == b"ACGTACGT"
    assert got == n
