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
