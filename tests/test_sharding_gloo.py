"""The N>1 path on CPU: two `gloo` ranks shard a read set, process their shards independently (no
data-path collective), and rank 0 gathers by read id.  The per-read "work" here is a byte checksum --
the GPU kernels are covered by the `-m gpu` tests; this covers sharding, barrier, max-over-ranks
timing and the host merge."""
import os
import socket
import zlib

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepchopper_amd import synth
from deepchopper_amd.sharding import dist_barrier, dist_max, gather_results, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _work(ids, seqs, quals):
    return {i: (zlib.crc32(s.tobytes()), zlib.crc32(q.tobytes())) for i, s, q in zip(ids, seqs, quals)}


def _rank_main(rank, world, port, n_reads, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seq, qual, ids = synth.synth_reads_fixed(n_reads, 64, seed=42)
    lo, hi = shard_range(n_reads, rank, world)
    dist_barrier()
    local = _work(ids[lo:hi], seq[lo:hi], qual[lo:hi])
    elapsed = dist_max(1.0 + rank)  # the slowest rank defines the step time
    assert elapsed == float(world)
    merged = gather_results(local)
    if rank == 0:
        torch.save({"merged": merged, "elapsed": elapsed}, out_path)
    dist_barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    n_reads, world = 101, 2
    out = str(tmp_path / "merged.pt")
    mp.spawn(_rank_main, args=(world, _free_port(), n_reads, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=False)
    seq, qual, ids = synth.synth_reads_fixed(n_reads, 64, seed=42)
    assert got["merged"] == _work(ids, seq, qual)
    assert got["elapsed"] == 2.0


def test_single_process_helpers_are_identity():
    assert dist_max(3.5) == 3.5
    dist_barrier()
    assert gather_results({"a": 1}) == {"a": 1}


def _merge_rank(rank, world, port, fq_path, parts, sabotage, out_dir):
    """A rank of `predict --chop` under an external launcher, after its predict_chop: parts written, counts known."""
    import argparse
    import json

    from deepchopper_amd import cli

    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    args = argparse.Namespace(data_path=fq_path, chop_output=os.path.join(out_dir, "merged"))
    outcome = "ok"
    try:
        counts = (0, 0) if sabotage else ((3, 3) if rank == 0 else (2, 2))
        cli._merge_under_external_launcher(args, rank, world, parts, counts[0], counts[1], interleaved=False)
    except Exception as e:
        outcome = f"{type(e).__name__}: {e}"
    with open(os.path.join(out_dir, f"outcome{rank}.json"), "w") as fh:
        json.dump(outcome, fh)


def _make_parts(tmp_path):
    from deepchopper_amd.chop import write_part_index
    from deepchopper_amd.fastq import BgzfWriter, FastqChunk, FastqReader

    def copy(c):  # a reader's chunks share its staging buffers
        return FastqChunk(c.n, c.seq.copy(), c.qual.copy(), c.offsets.copy(), c._names.copy(), c._name_offsets.copy(),
                          None if c._name_len is None else c._name_len.copy())

    seqs, quals, ids = synth.synth_reads_mixed(5, 60, 200, seed=4)
    text = synth.to_fastq(seqs, quals, ids)
    fq = tmp_path / "r.fq"
    fq.write_bytes(text)
    parts = []
    with FastqReader(str(fq)) as rd:
        chunks = [copy(c) for c in rd.chunks(1, 1 << 20)]
        for r, sl in enumerate((slice(0, 3), slice(3, 5))):
            parts.append(str(tmp_path / f"part.rank{r}"))
            w = BgzfWriter(parts[-1], threads=1)
            with w:
                for c in chunks[sl]:
                    w.write_chunk(c)
                    w.mark()
            write_part_index(parts[-1], w)
    return str(fq), parts, text


def test_external_launcher_merge_and_failure_reaches_every_rank(tmp_path):
    """`predict --chop` under torch.distributed.run (cli._merge_under_external_launcher): rank 0 splices the parts; when its
    merge fails, EVERY rank learns it at the rendezvous and raises -- nobody waits out a timeout, nothing is left behind
    (rank 0 used to raise before the final barrier)."""
    import glob
    import gzip
    import json

    for sabotage in (False, True):
        d = tmp_path / f"case{int(sabotage)}"
        d.mkdir()
        fq, parts, text = _make_parts(d)
        mp.spawn(_merge_rank, args=(2, _free_port(), fq, parts, sabotage, str(d)), nprocs=2, join=True)
        outcomes = [json.load(open(d / f"outcome{r}.json")) for r in range(2)]
        assert not any(os.path.exists(p) or os.path.exists(p + ".idx.npy") for p in parts)
        merged = glob.glob(str(d / "merged*"))
        if sabotage:
            assert "No valid records" in outcomes[0] and "rank 0 could not merge" in outcomes[1], outcomes
            assert merged == []
        else:
            assert outcomes == ["ok", "ok"], outcomes
            assert len(merged) == 1 and gzip.open(merged[0], "rb").read() == text
