"""`deepchopper predict` command line (deepchopper/cli.py:66-152) on the HIP hot path.

    python -m deepchopper_amd.cli predict reads.fastq -g 1 -o predictions -b 12 -m /path/to/checkpoint

Flags and defaults are the reference's (`--gpus/-g 0`, `--output/-o predictions`, `--batch-size/-b 12`,
`--workers/-w 0`, `--model/-m rna002`, `--limit-batches`, `--max-sample`, `--verbose/-v`).  Differences:
there is no CPU mode (`-g 0` is rejected: this build has no CPU fallback) and `-m` also accepts a local
checkpoint path because the Hub is unreachable here.  Output: `{out}/0/{rank}_{batch}.pt` in the
reference's CustomWriter format (callbacks.py:12-25), which the reference's `deepchopper chop`
consumes unchanged; `--native` additionally writes per-read labels / intervals (`{rank}.intervals.tsv`).
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
from pathlib import Path

import numpy as np

from .fastq import FastqReader, check_compressed_type, shard_byte_range


def rank_chunks(path: str, rank: int, world: int, max_reads: int, max_bases: int, max_sample: int | None = None):
    """This rank's reads as FastqChunks, streamed (never the whole file in memory).

    Uncompressed and BGZF input: one contiguous byte-range shard per rank, so a rank never
    parses -- or inflates -- another rank's reads.  Plain gzip cannot be entered mid-stream: every rank
    inflates the file and keeps every world-th chunk (chunk boundaries depend only on the file and the
    limits, so ranks agree on them).  `max_sample` (the reference's --max-sample: the first N records of
    the file) also uses the round-robin form because it needs global record numbers."""
    kind = check_compressed_type(path)
    if world > 1 and max_sample is None and kind in ("Uncompress", "Bgzip"):
        with FastqReader(path, byte_range=shard_byte_range(path, rank, world)) as rd:
            yield from rd.chunks(max_reads, max_bases)
        return
    seen = 0
    with FastqReader(path) as rd:
        for i, c in enumerate(rd.chunks(max_reads, max_bases)):
            if max_sample is not None and seen + c.n > max_sample:
                c = c.head(max_sample - seen)
            seen += c.n
            if c.n and i % world == rank:
                yield c
            if max_sample is not None and seen >= max_sample:
                return


def limit_batches(it, limit: int | None):
    for i, x in enumerate(it):
        if limit is not None and i >= limit:
            return
        yield x


def _predict(args) -> int:
    import torch

    from . import _lib
    from .model import HUB_NAMES, DeepChopper
    from .predict import PredictPipeline, write_pt_batch

    data_path = Path(args.data_path)
    if not data_path.exists():
        print(f"Error: Data path '{data_path}' does not exist.", file=sys.stderr)
        return 1
    if args.gpus < 1:
        print("Error: deepchopper_amd runs only on the GPU (use --gpus 1..8); there is no CPU fallback.", file=sys.stderr)
        return 1
    _lib.require_gpu()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("Error: launch multi-GPU predict with `python -m torch.distributed.run --nproc-per-node N -m "
              "deepchopper_amd.cli predict ...` (one process per GPU)", file=sys.stderr)
        return 1
    if args.batch_size % world != 0:  # only_fq.py:198-201
        raise RuntimeError(f"Batch size ({args.batch_size}) is not divisible by the number of devices ({world}).")
    torch.cuda.set_device(local_rank)
    model_name = HUB_NAMES.get(args.model.lower(), args.model)
    model = DeepChopper.from_pretrained(model_name)
    pipe = PredictPipeline(model)
    out_dir = Path(args.output or "predictions")
    per_dev = args.batch_size // world
    os.makedirs(out_dir / "0", exist_ok=True)
    if args.native:
        # zero-padding batches bounded by a token budget (mixed-length reads, BASELINE config 4)
        chunks = rank_chunks(str(data_path), rank, world, args.max_reads, args.max_tokens, args.max_sample)
    else:
        # reference batches: `per_dev` reads per step on each rank, left-padded to the longest
        chunks = rank_chunks(str(data_path), rank, world, per_dev, 1 << 30, args.max_sample)
    n_batches = n_reads = 0
    if args.native:
        with open(out_dir / "0" / f"{rank}.intervals.tsv", "w") as tsv:
            for res in pipe.predict_stream(limit_batches(chunks, args.limit_batches)):
                iv, niv, lens = res["intervals"], res["n_intervals"], res["lengths"]
                tsv.write("".join(
                    f"{name}\t{int(lens[r])}\t" + ",".join(f"{s}:{e}" for s, e in iv[r, : int(niv[r])].tolist()) + "\n"
                    for r, name in enumerate(res["names"])))
                if not args.no_pt:
                    write_pt_batch(str(out_dir / "0" / f"{rank}_{n_batches}.pt"), res["device"], res["names"])
                n_batches += 1
                n_reads += len(res["names"])
    else:
        for chunk in limit_batches(chunks, args.limit_batches):
            recs = list(chunk.records())
            ids = [r[0] for r in recs]
            _, res = pipe.predict_reads(ids, [r[1] for r in recs], [r[2] for r in recs], reference_padding=True)
            if not args.no_pt:
                write_pt_batch(str(out_dir / "0" / f"{rank}_{n_batches}.pt"), res, ids)
            n_batches += 1
            n_reads += len(ids)
    if n_reads == 0 and world == 1:
        raise ValueError(f"No valid records found in {data_path}")
    logging.info("rank %d of %d: %d reads in %d batches", rank, world, n_reads, n_batches)
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="deepchopper", description="DeepChopper (MI355X-native predict path)")
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("predict", help="DeepChopper: predict the given dataset")
    p.add_argument("data_path", help="Path to the dataset")
    p.add_argument("--gpus", "-g", type=int, default=0, help="Number of GPUs to use")
    p.add_argument("--output", "-o", default=None, help="Output path for predictions")
    p.add_argument("--batch-size", "-b", type=int, default=12, help="Batch size")
    p.add_argument("--workers", "-w", type=int, default=0, help="Number of workers (unused: tokenisation runs on the GPU)")
    p.add_argument("--model", "-m", default="rna002", help="Model name (rna002, rna004) or a local checkpoint path")
    p.add_argument("--limit-batches", type=int, default=None, help="Limit prediction batches")
    p.add_argument("--max-sample", type=int, default=None, help="Maximum number of samples to process")
    p.add_argument("--verbose", "-v", action="store_true", help="Enable verbose output")
    p.add_argument("--native", action="store_true", help="zero-padding batches + per-read intervals TSV")
    p.add_argument("--max-tokens", type=int, default=2_000_000, help="token budget per --native batch")
    p.add_argument("--no-pt", action="store_true", help="skip the 28 B/token .pt wire format (use with --native: "
                   "labels / intervals only)")
    p.add_argument("--max-reads", type=int, default=8192, help="read cap per --native batch (-b applies to reference batches)")
    c = sub.add_parser("chop", help="DeepChopper: chop the given predictions!")
    c.add_argument("predicts", nargs="+", help="Paths to prediction folders ({out}/0)")
    c.add_argument("fq", help="Path to FASTQ file")
    c.add_argument("--smooth-window", type=int, default=21)
    c.add_argument("--min-interval-size", type=int, default=13)
    c.add_argument("--approved-intervals", type=int, default=20)
    c.add_argument("--max-process-intervals", type=int, default=4)
    c.add_argument("--min-read-length", type=int, default=20)
    c.add_argument("--chop-type", default="all", choices=["terminal", "internal", "all"])
    c.add_argument("--output", "-o", default=None, help="Output prefix")
    c.add_argument("--max-batch", type=int, default=None)
    c.add_argument("--chunk-size", type=int, default=10000)
    c.add_argument("--threads", "-t", type=int, default=2, help="BGZF compression workers for the output (smoothing itself runs on the GPU)")
    c.add_argument("--verbose", "-v", action="store_true")
    args = ap.parse_args(argv)
    if args.verbose:
        logging.basicConfig(level=logging.INFO, format="%(message)s")
    if args.cmd == "chop":
        from . import _lib
        from .chop import chop
        from .predict import SmoothParams

        _lib.require_gpu()
        sp = SmoothParams(window=args.smooth_window, min_interval=args.min_interval_size,
                          approved=args.approved_intervals, max_process_intervals=args.max_process_intervals,
                          min_read_length_after_chop=args.min_read_length)
        path = chop(args.predicts, args.fq, sp, args.chop_type, args.output, args.chunk_size, args.max_batch, args.threads)
        print(path)
        return 0
    return _predict(args)


if __name__ == "__main__":
    raise SystemExit(main())
