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


def _external_rank(rank, world, port, fq_path, out_dir, mode):
    """A rank of `predict --chop` under an external launcher (cli.predict_chop_as_external_rank) with the GPU stage replaced:
    `predict_chop` copies the rank's chunks into its BGZF part, one block run per chunk, as the real one writes its records."""
    import json

    from deepchopper_amd import chop as chop_mod
    from deepchopper_amd import cli
    from deepchopper_amd.fastq import BgzfWriter

    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank), "LOCAL_WORLD_SIZE": str(world), "DC_SHARE_GPU": "1"})

    def fake_predict_chop(fq, pipe, chop_type="all", output_prefix=None, threads=2, ocq=False, chunks=None, part_path=None, **kw):
        if mode == "rank1_fails" and rank == 1:
            raise RuntimeError("boom on rank 1")
        n = 0
        w = BgzfWriter(part_path, threads=1)
        with w:
            for c in chunks:
                n += c.n
                w.write_chunk(c)
                w.mark()
        chop_mod.write_part_index(part_path, w)
        return part_path, n, n

    chop_mod.predict_chop = fake_predict_chop
    outcome = {"ok": True}
    try:
        rep = cli.predict_chop_as_external_rank(fq_path, None, rank, world, chop_output=os.path.join(out_dir, "merged"), max_reads=1,
                                                max_tokens=1 << 20, quiet=True, single_inflater=mode != "every_rank_inflates")
        outcome.update({k: rep.get(k) for k in ("reads", "total_reads", "input", "final", "feeder_cpu_s", "pipe_bytes")})
    except Exception as e:
        outcome = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    with open(os.path.join(out_dir, f"outcome{rank}.json"), "w") as fh:
        json.dump(outcome, fh)


def test_external_launcher_ranks_single_inflater_merge_and_failures(tmp_path):
    """`predict --chop` under torch.distributed.run (cli.predict_chop_as_external_rank), two gloo ranks on CPU:
    * an uncompressed file: byte-range shards, parts concatenated rank by rank;
    * one gzip stream: rank 0 is the node's single inflater and deals the chunks over FIFOs to both ranks (itself included), the
      parts are spliced chunk by chunk -- and the round 2-5 form in which every rank inflates for itself gives the same file;
    * the merged file IS the input, in input order; nothing is left behind (parts, sidecars, FIFO directory);
    * a rank that fails in its predict stage, or a merge that fails on rank 0 (no records), reaches EVERY rank at the rendezvous:
      everybody raises, nobody waits out a timeout."""
    import glob
    import gzip
    import json

    seqs, quals, ids = synth.synth_reads_mixed(9, 60, 200, seed=4)
    text = synth.to_fastq(seqs, quals, ids)
    for mode, name, data in (("shards", "r.fq", text), ("single_inflater", "r.fastq.gz", gzip.compress(text)),
                             ("every_rank_inflates", "r.fastq.gz", gzip.compress(text)), ("rank1_fails", "r.fastq.gz", gzip.compress(text)),
                             ("no_records", "r.fq", b"")):
        d = tmp_path / mode
        d.mkdir()
        fq = d / name
        fq.write_bytes(data)
        mp.spawn(_external_rank, args=(2, _free_port(), str(fq), str(d), mode), nprocs=2, join=True)
        outcomes = [json.load(open(d / f"outcome{r}.json")) for r in range(2)]
        left = [f for f in os.listdir(d) if f.startswith(".deepchopper")]
        assert left == [], (mode, left)
        merged = glob.glob(str(d / "merged*"))
        if mode == "rank1_fails":
            assert [o["ok"] for o in outcomes] == [False, False] and all("boom on rank 1" in o["error"] for o in outcomes), outcomes
            assert merged == []
        elif mode == "no_records":
            assert "No valid records" in outcomes[0]["error"] and "rank 0 could not merge" in outcomes[1]["error"], outcomes
            assert merged == []
        else:
            assert [o["ok"] for o in outcomes] == [True, True], outcomes
            assert sum(o["reads"] for o in outcomes) == 9 == outcomes[0]["total_reads"] and all(o["reads"] > 0 for o in outcomes)
            assert len(merged) == 1 and merged[0] == outcomes[0]["final"] and gzip.open(merged[0], "rb").read() == text
            want = {"shards": "byte-range shards", "single_inflater": "single inflater (rank 0) + FIFOs",
                    "every_rank_inflates": "every rank inflates the stream and keeps every world-th chunk"}[mode]
            assert all(o["input"] == want for o in outcomes), outcomes
            if mode == "single_inflater":
                assert outcomes[0]["feeder_cpu_s"] > 0 and all(o["pipe_bytes"] > 0 for o in outcomes)
