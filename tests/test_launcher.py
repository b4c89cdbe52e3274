"""CPU tests of the command's own per-GPU launcher (deepchopper_amd/launcher.py): worker environment, the
single-inflater chunk feed for plain gzip (parent parses once, workers receive their chunks over pipes), the
parent's merge of per-rank BGZF parts, failure propagation.  The workers here are plain CPU scripts."""
import gzip
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from deepchopper_amd import launcher, synth
from deepchopper_amd.chop import merge_bgzf_parts
from deepchopper_amd.fastq import BgzfWriter, FastqReader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, zlib
    sys.path.insert(0, %r)
    from deepchopper_amd.launcher import PipeChunks
    from deepchopper_amd.fastq import BgzfWriter
    from deepchopper_amd.chop import write_part_index
    fd, part = int(sys.argv[1]), sys.argv[2]
    n = 0
    names = []
    w = BgzfWriter(part, threads=1 + int(os.environ["RANK"]))
    with w:
        for c in PipeChunks(fd):
            n += c.n
            names += c.names()
            w.write_chunk(c)
            w.mark()  # as predict_chop_stream does after every chunk
    write_part_index(part, w)
    print("noise before the report")
    print(json.dumps({"rank": int(os.environ["RANK"]), "world": int(os.environ["WORLD_SIZE"]), "reads": n,
                      "first": names[0] if names else None, "local_rank": os.environ["LOCAL_RANK"],
                      "master": os.environ["MASTER_ADDR"]}))
""" % ROOT)


def test_chunk_frames_round_trip(tmp_path):
    seqs, quals, ids = synth.synth_reads_mixed(23, 50, 400, seed=3)
    fq = tmp_path / "a.fq"
    fq.write_bytes(synth.fastq_text([i.encode() + b" ch=%d" % k for k, i in enumerate(ids)], seqs, quals))
    for full in (False, True):
        frames = tmp_path / f"frames{int(full)}.bin"
        with FastqReader(str(fq), full_header=full) as rd, open(frames, "wb") as fh:
            want = []
            for c in rd.chunks(5, 3000):
                launcher.write_chunk(fh, c)
                want.append((c.n, c.seq.tobytes(), c.qual.tobytes(), c.offsets.tolist(), c._names.tobytes(), c.names(),
                             None if c._name_len is None else c._name_len.tolist()))
        got = []
        with open(frames, "rb") as fh:
            while True:
                c = launcher.read_chunk(fh)
                if c is None:
                    break
                got.append((c.n, c.seq.tobytes(), c.qual.tobytes(), c.offsets.tolist(), c._names.tobytes(), c.names(),
                            None if c._name_len is None else c._name_len.tolist()))
        assert got == want and sum(g[0] for g in got) == 23
        assert [n for g in got for n in g[5]] == ids


@pytest.mark.parametrize("world", [2, 3])
def test_single_inflater_feeds_every_worker_and_parts_merge(tmp_path, world):
    seqs, quals, ids = synth.synth_reads_mixed(61, 60, 900, seed=9)
    fq = tmp_path / "reads.fastq.gz"
    text = synth.to_fastq(seqs, quals, ids)
    with gzip.open(fq, "wb") as fh:
        fh.write(text)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    feeder = launcher.ChunkFeeder(str(fq), world, max_reads=4, max_bases=5000)
    port = launcher.free_port()
    procs, parts = [], []
    for r in range(world):
        parts.append(str(tmp_path / f"part{r}.gz"))
        procs.append(subprocess.Popen([sys.executable, str(script), str(feeder.read_fds[r]), parts[-1]],
                                      env=launcher.rank_env(r, world, port, share_gpu=(world == 3)),
                                      pass_fds=(feeder.read_fds[r],), stdout=subprocess.PIPE, text=True))
    feeder.start()
    outs = launcher.wait_ranks(procs, timeout=120)
    feeder.join()
    reports = [launcher.last_json_line(o) for o in outs]
    assert [r["rank"] for r in reports] == list(range(world)) and all(r["world"] == world for r in reports)
    assert sum(r["reads"] for r in reports) == 61 == feeder.n_reads
    assert all(r["reads"] > 0 for r in reports) and reports[0]["first"] == ids[0]
    assert [r["local_rank"] for r in reports] == (["0"] * 3 if world == 3 else ["0", "1"])
    assert all(r["master"] == "127.0.0.1" for r in reports)
    final = str(tmp_path / "merged.fq.gz")
    merge_bgzf_parts(parts, final, interleaved=True)
    assert not any(os.path.exists(p) or os.path.exists(p + ".idx.npy") for p in parts)
    # chunks were dealt round-robin and spliced back chunk by chunk: the merged file IS the input, in input order
    assert gzip.open(final, "rb").read() == text
    # ... and byte for byte the file ONE writer produces when it ends its blocks at the same chunk boundaries
    single = str(tmp_path / "single.fq.gz")
    with BgzfWriter(single, threads=2) as w, FastqReader(str(fq)) as rd:
        for c in rd.chunks(4, 5000):
            w.write_chunk(c)
            w.mark()
    assert open(final, "rb").read() == open(single, "rb").read()
    with FastqReader(final) as rd:
        assert sum(c.n for c in rd.chunks()) == 61


def test_contiguous_parts_merge_rank_by_rank_and_bad_deals_are_refused(tmp_path):
    from deepchopper_amd.chop import remove_parts, write_part_index

    seqs, quals, ids = synth.synth_reads_mixed(30, 60, 300, seed=4)
    fq = tmp_path / "r.fq"
    text = synth.to_fastq(seqs, quals, ids)
    fq.write_bytes(text)
    with FastqReader(str(fq)) as rd:
        chunks = [c for c in (FastqReaderChunkCopy(c) for c in rd.chunks(4, 1 << 20))]
    assert len(chunks) == 8
    parts = []
    for r, sl in enumerate((slice(0, 3), slice(3, 3), slice(3, 8))):  # rank 1 holds no read at all
        parts.append(str(tmp_path / f"p{r}"))
        w = BgzfWriter(parts[-1], threads=1)
        with w:
            for c in chunks[sl]:
                w.write_chunk(c)
                w.mark()
        write_part_index(parts[-1], w)
    with pytest.raises(ValueError, match="round-robin deal"):
        merge_bgzf_parts(parts, str(tmp_path / "bad.gz"), interleaved=True)
    assert all(os.path.exists(p) for p in parts)  # refused before anything was consumed
    final = str(tmp_path / "ok.gz")
    merge_bgzf_parts(parts, final)
    assert gzip.open(final, "rb").read() == text and not os.path.exists(parts[0])
    remove_parts(parts)  # idempotent


def FastqReaderChunkCopy(c):
    from deepchopper_amd.fastq import FastqChunk

    return FastqChunk(c.n, c.seq.copy(), c.qual.copy(), c.offsets.copy(), c._names.copy(), c._name_offsets.copy(),
                      None if c._name_len is None else c._name_len.copy())


def test_failed_worker_is_reported(tmp_path):
    ok = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(30)"], stdout=subprocess.PIPE, text=True)
    bad = subprocess.Popen([sys.executable, "-c", "import sys; print('{}'); sys.exit(3)"], stdout=subprocess.PIPE, text=True)
    with pytest.raises(RuntimeError, match="rank 1 exited with status 3"):
        launcher.wait_ranks([ok, bad], timeout=60)
    assert ok.poll() is not None  # the survivor was terminated, not left behind
    with pytest.raises(ValueError):
        launcher.last_json_line("no report here")
    assert launcher.last_json_line('x\n{"a": 1}\n') == {"a": 1}


def test_single_inflater_over_named_fifos_for_sibling_ranks(tmp_path):
    """Ranks an EXTERNAL launcher started are siblings: rank 0 runs the feeder on named FIFOs, every rank -- rank 0 included --
    opens its own (launcher.open_fifo).  Same deal as over anonymous pipes: chunk g goes to rank g % W; the feeder accounts
    the CPU of its threads and of the reader's native inflate helpers; the FIFOs are gone after join(); a rank that never
    shows up is an error after the patience, not a hang."""
    import threading

    seqs, quals, ids = synth.synth_reads_mixed(47, 60, 700, seed=11)
    fq = tmp_path / "reads.fastq.gz"
    with gzip.open(fq, "wb") as fh:
        fh.write(synth.to_fastq(seqs, quals, ids))
    world = 3
    fifos = [str(tmp_path / f"rank{r}.fifo") for r in range(world)]
    allowed = sorted(os.sched_getaffinity(0))
    feeder = launcher.ChunkFeeder(str(fq), world, max_reads=4, max_bases=5000, full_header=True, fifo_paths=fifos, cpus=allowed[:2])
    got = [[] for _ in range(world)]

    def rank(r):
        src = launcher.open_fifo(fifos[r])
        for c in src:
            got[r].append(c.names())
        src.close()

    feeder.start()
    threads = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    feeder.join()
    assert feeder.n_reads == 47 and feeder.cpu_s > 0
    assert not any(os.path.exists(f) for f in fifos)
    with FastqReader(str(fq)) as rd:
        want = [c.names() for c in rd.chunks(4, 5000)]
    for r in range(world):
        assert got[r] == want[r::world]
    assert [n for g in range(len(want)) for n in got[g % world][g // world]] == ids
    # a reader that never opens its end
    lone = str(tmp_path / "lone.fifo")
    os.mkfifo(lone)
    with pytest.raises(OSError):
        launcher.ChunkFeeder._open_fifo_for_writing(lone, patience_s=0.2)


def test_reader_accounts_its_inflate_threads(tmp_path):
    """dc_fq_helper_cpu_seconds: the CPU the reader's helper threads (read-ahead, inflate incl. the parallel member decoder)
    spent, exact once the stream is exhausted -- what the host-budget legs of bench.py charge the node's single inflater."""
    seqs, quals, ids = synth.synth_reads_mixed(4000, 800, 1200, seed=5)
    text = synth.to_fastq(seqs, quals, ids)
    fq = tmp_path / "big.fastq.gz"
    with gzip.open(fq, "wb", compresslevel=1) as fh:
        fh.write(text)
    with FastqReader(str(fq)) as rd:
        n = sum(c.n for c in rd.chunks(1000, 1 << 22))
        cpu = rd.helper_cpu_seconds
    assert n == 4000 and 0.001 < cpu < 30.0, cpu
