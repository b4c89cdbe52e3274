"""`deepchopper predict` / `deepchopper chop` command line (deepchopper/cli.py:66-170) on the HIP hot path.

    deepchopper predict reads.fastq -g 1 -o predictions -b 12 -m /path/to/checkpoint
    deepchopper predict reads.fastq.gz -g 8 --chop -m /path/to/checkpoint     # fused predict -> chop
    deepchopper chop predictions/0 reads.fastq                                 # (= deepchopper-chop --pdt ... --fq ...)

(`pip install -e .` installs the `deepchopper` and `deepchopper-chop` entry points of pyproject.toml -- the reference's
`deepchopper = "deepchopper.cli:app"`, pyproject.toml:58; `python -m deepchopper_amd.cli ...` works without installing.)

Flags and defaults are the reference's (`--gpus/-g 0`, `--output/-o predictions`, `--batch-size/-b 12`,
`--workers/-w 0`, `--model/-m rna002`, `--limit-batches`, `--max-sample`, `--verbose/-v`).  Differences:
there is no CPU mode (`-g 0` is rejected: this build has no CPU fallback) and `-m` also accepts a local
checkpoint path because the Hub is unreachable here.  Output: `{out}/0/{rank}_{batch}.pt` in the
reference's CustomWriter format (callbacks.py:12-25), which the reference's `deepchopper chop`
consumes unchanged; `--native` additionally writes per-read labels / intervals (`{rank}.intervals.tsv`);
`--chop` runs predict and chop as one streaming pass and writes `{stem}.{n}pd.{m}record.chop.fq.gz` with no
`.pt` files in between.

`--gpus N` (N > 1) starts one worker process per GPU itself, as the reference does through Lightning
(cli.py:126-147); it also runs under `python -m torch.distributed.run --nproc-per-node N` (a worker sees the
same environment either way).
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
from pathlib import Path

from .fastq import FastqReader, check_compressed_type, shard_byte_range


def rank_chunks(path: str, rank: int, world: int, max_reads: int, max_bases: int, max_sample: int | None = None,
                full_header: bool = False):
    """This rank's reads as FastqChunks, streamed (never the whole file in memory).

    Uncompressed and BGZF input: one contiguous byte-range shard per rank, so a rank never
    parses -- or inflates -- another rank's reads.  Plain gzip cannot be entered mid-stream: under the
    command's own launcher the parent inflates the file once and deals chunks to the workers
    (launcher.ChunkFeeder); under an external launcher every rank inflates the file and keeps every
    world-th chunk (chunk boundaries depend only on the file and the limits, so ranks agree on them).
    `max_sample` (the reference's --max-sample: the first N records of the file) also uses the round-robin
    form because it needs global record numbers."""
    kind = check_compressed_type(path)
    if world > 1 and max_sample is None and kind in ("Uncompress", "Bgzip"):
        with FastqReader(path, byte_range=shard_byte_range(path, rank, world), full_header=full_header) as rd:
            yield from rd.chunks(max_reads, max_bases)
        return
    seen = 0
    with FastqReader(path, full_header=full_header) as rd:
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


def _smooth_params(args):
    from .predict import SmoothParams

    return SmoothParams(window=args.smooth_window, min_interval=args.min_interval_size, approved=args.approved_intervals,
                        max_process_intervals=args.max_process_intervals, min_read_length_after_chop=args.min_read_length)


def _chunk_limits(args, world: int) -> tuple[int, int]:
    if args.native or (args.chop and not args.reference_batching):
        return args.max_reads, args.max_tokens  # zero-padding batches bounded by a token budget (config 4)
    return args.batch_size // world, 1 << 30    # reference batches: per-device reads per step, left-padded


def _process_accounting(t_start: float) -> dict:
    """CPU seconds / wall seconds / peak RSS of this process: what the host-budget measurement of an N-rank run adds up
    (tools/host_budget.py, profiles/r03_host_budget_8rank.json)."""
    import resource
    import time

    ru = resource.getrusage(resource.RUSAGE_SELF)
    return {"cpu_user_s": round(ru.ru_utime, 3), "cpu_sys_s": round(ru.ru_stime, 3), "wall_s": round(time.time() - t_start, 3),
            "max_rss_mb": round(ru.ru_maxrss / 1024, 1)}


def _launch_workers(args, argv: list[str]) -> int:
    """Parent of a self-launched multi-GPU run: never touches the GPU."""
    from . import chop as chop_mod
    from . import launcher

    import time

    t_start = time.time()
    world = args.gpus
    path = str(args.data_path)
    share = os.environ.get("DC_SHARE_GPU") == "1"
    kind = check_compressed_type(path)
    feeder = None
    child = launcher.worker_python() + ["-m", "deepchopper_amd.cli"] + argv
    part_base = None
    if args.chop:
        part_base = chop_mod._temp_name(path, args.chop_output, tag="")
    per_rank_fds = None
    if kind == "Gzip" or args.max_sample is not None:
        max_reads, max_bases = _chunk_limits(args, world)
        feeder = launcher.ChunkFeeder(path, world, max_reads, max_bases, full_header=bool(args.chop), max_sample=args.max_sample)
        per_rank_fds = [(fd,) for fd in feeder.read_fds]
    procs = []
    port = launcher.free_port()
    import subprocess

    parts = [f"{part_base}.rank{r}" for r in range(world)] if part_base else []
    plan = launcher.cpu_plan(world, share)  # a disjoint CPU slice of its GPU's NUMA node per worker, the rest for this parent
    try:
        for r in range(world):
            env = launcher.rank_env(r, world, port, share, plan)
            cmd = list(child)
            if feeder is not None:
                cmd += ["--chunk-fd", str(feeder.read_fds[r])]
            if part_base:
                cmd += ["--part-path", parts[r]]
            procs.append(subprocess.Popen(cmd, env=env, pass_fds=per_rank_fds[r] if per_rank_fds else (), stdout=subprocess.PIPE, text=True))
        parent_cpus = launcher.bind_parent(plan)  # after the spawn (children must not inherit it), before the inflater threads
        if feeder is not None:
            feeder.start()
        outs = launcher.wait_ranks(procs)
        if feeder is not None:
            feeder.join()
        reports = [launcher.last_json_line(o) for o in outs]
        n_reads = sum(r["reads"] for r in reports)
        if n_reads == 0:
            raise ValueError(f"No valid records found in {path}")
        summary = {"gpus": world, "reads": n_reads, "batches": sum(r["batches"] for r in reports),
                   "per_rank_reads": [r["reads"] for r in reports],
                   "cpusets": {"parent": parent_cpus and parent_cpus["cpuset"], "ranks": [r.get("cpuset") for r in reports]}}
        if args.chop:
            n_out = sum(r["records"] for r in reports)
            final = chop_mod._final_name(path, args.chop_output, n_reads, n_out)
            # dealt chunks (one gzip stream / --max-sample) are spliced chunk by chunk, contiguous shards rank by rank:
            # the output is in input order either way
            chop_mod.merge_bgzf_parts(parts, final, interleaved=feeder is not None)
            summary.update({"records": n_out, "output": final})
            print(final)
    finally:
        chop_mod.remove_parts(parts)  # nothing left behind when a worker, the feeder or the merge failed
    logging.info("%s", json.dumps(summary))
    if args.report_json:
        with open(args.report_json, "w") as fh:
            json.dump({**summary, "input": "single inflater + pipes" if feeder is not None else "byte-range shards",
                       "parent": _process_accounting(t_start), "ranks": reports}, fh)
    return 0


def predict_chop_as_external_rank(data_path: str, pipe, rank: int, world: int, *, chop_type: str = "all", chop_output: str | None = None,
                                  threads: int = 4, ocq: bool = False, max_reads: int = 8192, max_tokens: int = 2_000_000,
                                  reference_batch: int | None = None, max_sample: int | None = None, limit: int | None = None,
                                  single_inflater: bool = True, quiet: bool = False) -> dict:
    """`predict --chop` as ONE RANK of a job an external launcher started (torch.distributed.run, srun, mpirun: the ranks are
    siblings, there is no parent of ours to feed them or to merge).  The ranks meet on a gloo group -- control plane only: a
    few small objects and barriers -- and

    * input.  Uncompressed / BGZF: every rank reads its own contiguous byte range (`rank_chunks`).  One plain gzip stream
      (or --max-sample, which needs global record numbers) has one entry point: rank 0 runs the node's SINGLE inflater
      (`launcher.ChunkFeeder` on named FIFOs in a directory announced over the group, bound to the CPUs the placement plan keeps
      for a parent) and deals the chunks round-robin to every rank, itself included -- as the command's own parent does for
      its children.  (Rounds 2-5: every rank inflated the whole file and kept every world-th chunk: N x the inflate CPU.
      `single_inflater=False` keeps that form: A/B and the fallback where FIFOs cannot be made.)
    * output.  Every rank writes its BGZF part; rank 0 splices them in input order (`chop.merge_bgzf_parts`) exactly as the
      self-launching parent does, names the result and prints it; every rank learns the outcome.

    Returns this rank's report: reads / records (own and total), the final path (rank 0), CPU seconds of the pipe reads and
    of the feeder.  The reference's flow being replaced: deepchopper/cli.py:126-152 (Lightning read-sharding) followed by
    src/bin/predict.rs:275-316 (the chop driver's chunked loop)."""
    import datetime
    import shutil
    import tempfile

    import torch.distributed as dist

    from . import chop as chop_mod
    from . import launcher, placement

    own = not dist.is_initialized()
    if own:  # gloo announces its connections on stdout, which carries the output path: point fd 1 at stderr meanwhile
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            # a rank that died never arrives: do not wait gloo's default 30 minutes for it under a launcher that leaves the
            # siblings alone (srun, mpirun); bench.py bounds the same barrier pattern the same way
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    kind = check_compressed_type(data_path)
    dealt = kind not in ("Uncompress", "Bgzip") or max_sample is not None
    chunk_reads, chunk_bases = (reference_batch, 1 << 30) if reference_batch else (max_reads, max_tokens)
    base = os.path.join(os.path.dirname(chop_mod._temp_name(data_path, chop_output)),
                        f".deepchopper_temp_job{os.environ.get('MASTER_PORT', '0')}.fq.gz")
    parts = [f"{base}.rank{r}" for r in range(world)]
    report = {"rank": rank, "dealt": dealt, "input": "byte-range shards"}
    feeder = pipe_src = fifo_dir = drain = None
    failure = None
    try:
        if dealt and single_inflater:
            box = [None]
            if rank == 0:
                try:
                    fifo_dir = tempfile.mkdtemp(prefix=".deepchopper_feed_", dir=os.path.dirname(base) or ".")
                    cpus = None
                    try:
                        share = os.environ.get("DC_SHARE_GPU") == "1" or os.environ.get("DC_BENCH_SHARE_GPU") == "1"
                        lws = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
                        plan = placement.plan(lws, local_devices=[0] * lws if share else None) if placement.enabled() else None
                        cpus = plan["parent"]["cpus"] if plan and plan.get("parent") else None
                    except Exception:
                        cpus = None
                    feeder = launcher.ChunkFeeder(data_path, world, chunk_reads, chunk_bases, full_header=True, max_sample=max_sample,
                                                  fifo_paths=[os.path.join(fifo_dir, f"rank{r}.fifo") for r in range(world)], cpus=cpus)
                    box[0] = fifo_dir
                    report["feeder_cpuset"] = placement.format_cpulist(cpus) if cpus else None
                except OSError:  # no FIFOs on this file system: every rank inflates for itself
                    feeder = None
            dist.broadcast_object_list(box, src=0)
            if box[0] is not None:
                if feeder is not None:
                    feeder.start()
                pipe_src = launcher.open_fifo(os.path.join(box[0], f"rank{rank}.fifo"))
                report["input"] = "single inflater (rank 0) + FIFOs"
        if pipe_src is not None:
            chunks = iter(pipe_src)
        else:
            if dealt:
                report["input"] = "every rank inflates the stream and keeps every world-th chunk"
            chunks = rank_chunks(data_path, rank, world, chunk_reads, chunk_bases, max_sample, full_header=True)
        chunks = limit_batches(chunks, limit)
        ref_kw = {"max_reads": max_reads, "max_tokens": max_tokens}
        if reference_batch:
            ref_kw.update({"reference_batch": reference_batch, "mark_sub_batches": dealt})
        n_reads = n_records = 0
        import time

        t_stream = time.perf_counter()
        try:
            _, n_reads, n_records = chop_mod.predict_chop(data_path, pipe, chop_type, chop_output, threads, ocq, chunks=chunks,
                                                          part_path=parts[rank], **ref_kw)
            report["stream_s"] = round(time.perf_counter() - t_stream, 3)  # this rank's reads: first chunk in -> BGZF part closed
            if pipe_src is not None:
                for _ in pipe_src:  # (--limit-batches: drain what the feeder still deals this rank, or its writer blocks)
                    pass
        except Exception as e:  # this rank must still reach the rendezvous below, or the others wait out the timeout
            failure = e
            if pipe_src is not None:  # ... and must not leave its share of the deal unread: the feeder's writer for this rank
                import threading      # would block on a full FIFO and, behind it, every other rank's supply.  (Not close():
                                      # a writer still waiting to open its end would wait for a reader that is gone.)
                def _discard(src=pipe_src):
                    try:
                        for _ in src:
                            pass
                    except Exception:
                        pass

                drain = threading.Thread(target=_discard, daemon=True)
                drain.start()
        if feeder is not None:
            try:
                feeder.join()
                report["feeder_cpu_s"] = round(feeder.cpu_s, 3)
            except Exception as e:
                failure = failure or e
        counts = [None] * world
        dist.all_gather_object(counts, (n_reads, n_records, None if failure is None else f"{type(failure).__name__}: {failure}"))
        bad = [(r, c[2]) for r, c in enumerate(counts) if c[2] is not None]
        final = None
        if rank == 0 and not bad:
            try:
                tot_r, tot_o = sum(c[0] for c in counts), sum(c[1] for c in counts)
                if tot_r == 0:
                    raise ValueError(f"No valid records found in {data_path}")
                final = chop_mod._final_name(data_path, chop_output, tot_r, tot_o)
                t_merge = time.perf_counter()
                chop_mod.merge_bgzf_parts(parts, final, interleaved=dealt)
                report["merge_s"] = round(time.perf_counter() - t_merge, 3)
                if not quiet:
                    print(final)
            except Exception as e:
                failure = e
        verdict = [None if failure is None else f"{type(failure).__name__}: {failure}"]
        dist.broadcast_object_list(verdict, src=0)  # doubles as the final barrier; every rank learns the outcome
        if failure is None and bad:
            failure = RuntimeError(f"rank {bad[0][0]} failed: {bad[0][1]}")
        if failure is None and verdict[0] is not None:
            failure = RuntimeError(f"rank 0 could not merge the per-rank outputs: {verdict[0]}")
        report.update({"reads": n_reads, "records": n_records, "total_reads": sum(c[0] for c in counts),
                       "total_records": sum(c[1] for c in counts), "final": final})
        if pipe_src is not None:
            report.update({"pipe_read_cpu_s": round(pipe_src.cpu_s, 3), "pipe_bytes": pipe_src.bytes})
    finally:
        if drain is not None:
            drain.join(timeout=30)  # (the feeder has closed its ends by the time the rendezvous above is through)
        if pipe_src is not None:
            try:
                pipe_src.close()
            except Exception:
                pass
        if rank == 0:
            chop_mod.remove_parts(parts)
            if fifo_dir:
                shutil.rmtree(fifo_dir, ignore_errors=True)
        if own:
            dist.destroy_process_group()
    if failure is not None:
        raise failure  # every rank exits non-zero
    return report


def _predict(args, argv: list[str]) -> int:
    data_path = Path(args.data_path)
    if not data_path.exists():
        print(f"Error: Data path '{data_path}' does not exist.", file=sys.stderr)
        return 1
    if args.gpus is None:
        # the reference's documented command is `deepchopper predict reads.fq` (default --gpus 0 = CPU, deepchopper/cli.py:66-90,
        # 126-138).  This build has no CPU mode; the omitted flag means "one GPU" instead of an error.
        args.gpus = 1
        if os.environ.get("DC_SELF_LAUNCHED") != "1":
            print("Note: no CPU mode in this build; using GPU 0 (pass --gpus N for N GPUs).", file=sys.stderr)
    if args.gpus < 1:
        print("Error: deepchopper_amd runs only on the GPU (use --gpus 1..8); there is no CPU fallback.", file=sys.stderr)
        return 1
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.batch_size % max(args.gpus, 1) != 0:  # only_fq.py:198-201
        raise RuntimeError(f"Batch size ({args.batch_size}) is not divisible by the number of devices ({args.gpus}).")
    if args.gpus > 1 and world == 1:
        return _launch_workers(args, argv)  # before anything touches the GPU

    import time

    t_start = time.time()
    from . import placement

    placed = placement.apply_for_this_rank()  # before torch is imported and before the first HIP call (pinned staging after binding)
    import torch

    from . import _lib
    from .model import HUB_NAMES, DeepChopper
    from .predict import PredictPipeline, write_pt_batches

    _lib.require_gpu()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DC_SHARE_GPU") == "1":  # rehearsing N ranks on a box with fewer GPUs (tests): ranks share devices
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    model_name = HUB_NAMES.get(args.model.lower(), args.model)
    model = DeepChopper.from_pretrained(model_name)
    pipe = PredictPipeline(model, smooth=_smooth_params(args))
    t_ready = time.time()  # interpreter, torch, the libraries, weights on the GPU: everything before the first read
    out_dir = Path(args.output or "predictions")
    max_reads, max_bases = _chunk_limits(args, world)
    pipe_src = None
    if args.chunk_fd is not None:
        from .launcher import PipeChunks

        pipe_src = PipeChunks(args.chunk_fd)  # the parent is this node's single reader / inflater
        chunks = iter(pipe_src)
    else:
        chunks = rank_chunks(str(data_path), rank, world, max_reads, max_bases, args.max_sample, full_header=bool(args.chop))
    chunks = limit_batches(chunks, args.limit_batches)
    n_batches = n_reads = n_records = 0
    t_streamed = pt_timers = None
    if args.chop:
        from .chop import predict_chop

        def count(res):
            nonlocal n_batches
            n_batches += 1

        # --reference-batching: the reader chunks are the reference's batches (-b reads per device); they share launches of
        # --max-reads / --max-tokens.  Where the chunks were dealt round-robin (one gzip stream, --max-sample) the parts are
        # spliced batch by batch, so every batch ends its own BGZF block
        ref_kw = {"max_reads": args.max_reads, "max_tokens": args.max_tokens}
        if args.reference_batching:
            dealt_in = args.chunk_fd is not None or (world > 1 and (
                check_compressed_type(str(data_path)) not in ("Uncompress", "Bgzip") or args.max_sample is not None))
            ref_kw.update({"reference_batch": max_reads, "mark_sub_batches": dealt_in})
        if args.part_path:  # worker of a multi-GPU run: the parent splices the parts and names the result
            _, n_reads, n_records = predict_chop(str(data_path), pipe, args.chop_type, args.chop_output, args.threads,
                                                 args.ocq, chunks=chunks, part_path=args.part_path, **ref_kw)
        elif world > 1:  # external launcher (torch.distributed.run): single inflater on rank 0 / byte-range shards, rank 0 merges
            ext = predict_chop_as_external_rank(str(data_path), pipe, rank, world, chop_type=args.chop_type, chop_output=args.chop_output,
                                                threads=args.threads, ocq=args.ocq, max_reads=args.max_reads, max_tokens=args.max_tokens,
                                                reference_batch=max_reads if args.reference_batching else None,
                                                max_sample=args.max_sample, limit=args.limit_batches,
                                                single_inflater=os.environ.get("DC_SINGLE_INFLATER", "1") != "0")
            n_reads, n_records = ext["reads"], ext["records"]
        else:
            final, n_reads, n_records = predict_chop(str(data_path), pipe, args.chop_type, args.chop_output, args.threads,
                                                     args.ocq, chunks=chunks, **ref_kw)
            print(final)
    else:
        os.makedirs(out_dir / "0", exist_ok=True)
        tsv = open(out_dir / "0" / f"{rank}.intervals.tsv", "w") if args.native else None
        # the 28 B/token wire format is written on worker threads (--threads; the reference writes from its one predict loop,
        # callbacks.py:12-25) by the native writer (include/deepchopper_io.h: dc_pt_*): a fused launch is ~85 files of ~1 MB,
        # and torch.save -- byte-wise CRC-32, Python per file -- topped out near 750 MB/s over all threads next to a GPU path
        # that produces 2.2 GB/s of `.pt` on a nanopore-like length mix.  DC_PT_WRITER=torch keeps torch.save (A/B)
        from concurrent.futures import ThreadPoolExecutor

        from .predict import PtWriter

        native_pt = os.environ.get("DC_PT_WRITER", "native") != "torch" and not args.no_pt
        pool = ThreadPoolExecutor(max(1, args.threads)) if not args.no_pt and not native_pt else None
        writer = PtWriter(threads=max(1, args.threads)) if native_pt else None
        pending: list = []
        try:
            # reference mode (-b): left-padded batches whose pad tokens run through the model, int64
            # ids for the .pt wire format; --native: every read at its own length
            if not args.native:
                # reference batches (-b reads each, left-padded per batch) share GPU launches: a 12-read batch fills a
                # fifth of the chip, a fused launch all of it; every batch still gets its own `.pt` file
                from .fastq import fuse_chunks

                # (--max-tokens budgets what a launch runs through the GEMMs: the reads' own tokens when the [PAD] runs are
                #  shared, the padded rows otherwise)
                chunks = fuse_chunks(chunks, max_reads=args.max_reads, max_tokens=args.max_tokens, padded=not pipe.shared_prefix)
            for res in pipe.predict_stream(chunks, reference_padding=not args.native, ids_i64=pool is not None,
                                           want_labels=not args.no_pt):
                if tsv is not None:
                    iv, niv, lens = res["intervals"], res["n_intervals"], res["lengths"]
                    tsv.write("".join(
                        f"{name}\t{int(lens[r])}\t" + ",".join(f"{s}:{e}" for s, e in iv[r, : int(niv[r])].tolist()) + "\n"
                        for r, name in enumerate(res["names"])))
                if writer is not None:  # one reference batch = one file, as CustomWriter writes them
                    n_batches += writer.submit(str(out_dir / "0"), rank, n_batches, res["device"], res["names"],
                                               ready_event=res.get("ready_event"), truncated=res["truncated"])
                elif not args.no_pt:
                    n_batches += write_pt_batches(str(out_dir / "0"), rank, n_batches, res["device"], res["names"], pool, pending)
                    while len(pending) > 4 * max(1, args.threads):  # bound the host copies waiting to be written
                        pending.pop(0).result()
                else:
                    n_batches += len(res.get("sub_sizes") or [0])
                n_reads += len(res["names"])
            t_streamed = time.time()
            for f in pending:
                f.result()  # (a failed write raises here)
            if writer is not None:
                w, writer = writer, None
                w.close()   # (... or here)
                pt_timers = {k: round(v, 3) for k, v in w.timers.items()}
        finally:
            if writer is not None:
                try:
                    writer.close()
                except Exception:  # the error that brought us here is the one to report
                    pass
            if pool is not None:
                pool.shutdown(wait=True)
            if tsv is not None:
                tsv.close()
    if n_reads == 0 and world == 1:
        raise ValueError(f"No valid records found in {data_path}")
    logging.info("rank %d of %d: %d reads in %d batches", rank, world, n_reads, n_batches)
    report = {"rank": rank, "reads": n_reads, "batches": n_batches, "records": n_records, **_process_accounting(t_start)}
    # where the wall time went: start-up (imports, weights, handles) | the stream (parse -> GPU -> results handed to the writers)
    # | what the writers still had to finish behind the last launch
    t_end = time.time()
    if pt_timers is not None:
        report["pt_writer_main_thread_s"] = pt_timers  # where the predict loop itself waited on the `.pt` path
    report["phases_s"] = {"startup": round(t_ready - t_start, 3), "stream": round((t_streamed or t_end) - t_ready, 3),
                          "drain": round(t_end - (t_streamed or t_end), 3)}
    if placed is not None:
        report.update({"cpuset": placed["cpuset"], "numa_node": placed["numa_node"], "torch_threads": torch.get_num_threads(),
                       "cpu_budget": placed.get("cpu_budget"), "capped_to_budget": placed.get("capped_to_budget")})
    if pipe_src is not None:
        report.update({"pipe_read_cpu_s": round(pipe_src.cpu_s, 3), "pipe_bytes": pipe_src.bytes})
    if args.chop:
        from .chop import LAST_WRITER_STATS

        report.update({k: (round(v, 3) if isinstance(v, float) else v) for k, v in LAST_WRITER_STATS.items()})
    if os.environ.get("DC_SELF_LAUNCHED") == "1":
        print(json.dumps(report), flush=True)
    elif args.report_json and world == 1:
        with open(args.report_json, "w") as fh:
            json.dump({"gpus": 1, "ranks": [report]}, fh)
    return 0


def _add_chop_options(p, prefix_output: bool, threads: int = 2) -> None:
    p.add_argument("--smooth-window", type=int, default=21)
    p.add_argument("--min-interval-size", type=int, default=13)
    p.add_argument("--approved-intervals", type=int, default=20)
    p.add_argument("--max-process-intervals", type=int, default=4)
    p.add_argument("--min-read-length", type=int, default=20)
    p.add_argument("--chop-type", default="all", choices=["terminal", "internal", "all"])
    p.add_argument("--output-chopped", "--ocq", dest="ocq", action="store_true",
                   help="output the chopped (adapter) segments only (deepchopper/cli.py:163 `--output-chopped`; "
                        "`--ocq` is the chop binary's spelling, src/bin/predict.rs:60-62)")
    p.add_argument("--threads", "-t", type=int, default=threads,
                   help="BGZF compression workers for the output (smoothing itself runs on the GPU).  `chop`: 2 as the reference "
                        "(cli.py:165); `predict --chop`: 4 -- one GPU emits ~30 Mbases/s of FASTQ text, which costs 2.4 cores of "
                        "zlib level 6 (profiles/r03_host_budget_8rank.json: 21.4 k reads/s with 2 workers, 25.7 k with 4)")
    if prefix_output:
        p.add_argument("--output", "-o", default=None, help="Output prefix")


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    from . import __version__

    ap = argparse.ArgumentParser(prog="deepchopper", description="DeepChopper (MI355X-native predict path)")
    ap.add_argument("--version", "-V", action="version", version=f"DeepChopper Version: {__version__}")  # cli.py:204-208
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("predict", help="DeepChopper: predict the given dataset")
    p.add_argument("data_path", help="Path to the dataset")
    p.add_argument("--gpus", "-g", type=int, default=None,
                   help="Number of GPUs to use (omitted: one GPU -- the reference's default 0 means CPU, which this build "
                        "does not have; an explicit 0 is an error)")
    p.add_argument("--output", "-o", default=None, help="Output path for predictions")
    p.add_argument("--batch-size", "-b", type=int, default=12, help="Batch size")
    p.add_argument("--workers", "-w", type=int, default=0, help="Number of workers (unused: tokenisation runs on the GPU)")
    p.add_argument("--model", "-m", default="rna002", help="Model name (rna002, rna004) or a local checkpoint path")
    p.add_argument("--limit-batches", type=int, default=None, help="Limit prediction batches")
    p.add_argument("--max-sample", type=int, default=None, help="Maximum number of samples to process")
    p.add_argument("--verbose", "-v", action="store_true", help="Enable verbose output")
    p.add_argument("--native", action="store_true", help="zero-padding batches + per-read intervals TSV")
    p.add_argument("--max-tokens", type=int, default=2_000_000, help="token budget per --native / --chop batch")
    p.add_argument("--no-pt", action="store_true", help="skip the 28 B/token .pt wire format (use with --native: "
                   "labels / intervals only)")
    p.add_argument("--max-reads", type=int, default=8192, help="read cap per --native / --chop batch (-b applies to reference batches)")
    p.add_argument("--chop", action="store_true", help="fused predict -> chop: write the chopped reads as BGZF FASTQ, no .pt files")
    p.add_argument("--chop-output", default=None, help="output prefix of --chop (default: the FASTQ's stem, as deepchopper chop)")
    p.add_argument("--reference-batching", action="store_true",
                   help="with --chop: run the reads in the reference's batches (-b reads, left-padded per batch, pad tokens through "
                        "the model) so that the chopped reads are those of the reference's `predict` + `chop`; default: every read "
                        "at its own length (= the reference at -b 1)")
    _add_chop_options(p, prefix_output=False, threads=4)
    p.add_argument("--report-json", default=None, help=argparse.SUPPRESS)          # host-budget accounting of the run
    p.add_argument("--chunk-fd", type=int, default=None, help=argparse.SUPPRESS)   # worker: chunks come from the parent
    p.add_argument("--part-path", default=None, help=argparse.SUPPRESS)            # worker: per-rank BGZF part
    c = sub.add_parser("chop", help="DeepChopper: chop the given predictions!")
    c.add_argument("predicts", nargs="+", help="Paths to prediction folders ({out}/0)")
    c.add_argument("fq", help="Path to FASTQ file")
    _add_chop_options(c, prefix_output=True)
    c.add_argument("--max-batch", type=int, default=None)
    c.add_argument("--chunk-size", type=int, default=10000)
    c.add_argument("--verbose", "-v", action="store_true")
    args = ap.parse_args(argv)
    if args.verbose:
        logging.basicConfig(level=logging.INFO, format="%(message)s")
    if args.cmd == "chop":
        from . import _lib
        from .chop import chop

        _lib.require_gpu()
        path = chop(args.predicts, args.fq, _smooth_params(args), args.chop_type, args.output, args.chunk_size, args.max_batch,
                    args.threads, args.ocq)
        print(path)
        return 0
    return _predict(args, argv)


def chop_main(argv=None) -> int:
    """`deepchopper-chop`: the chop binary's own flag names (clap, src/bin/predict.rs:19-78) -- what the reference's
    `deepchopper chop` shells out to (deepchopper/cli.py:177-195) -- on the same driver as `deepchopper chop`."""
    from . import __version__

    ap = argparse.ArgumentParser(prog="deepchopper-chop")
    ap.add_argument("--version", "-V", action="version", version=f"deepchopper-chop {__version__}")
    ap.add_argument("--pdt", action="append", required=True, metavar="predicts", help="path to the predicts")
    ap.add_argument("--fq", required=True, help="path to the integrated fq file")
    ap.add_argument("-t", "--threads", type=int, default=2)
    ap.add_argument("-m", "--max-batch-size", type=int, default=None)
    ap.add_argument("-s", "--smooth-window-size", type=int, default=21)
    ap.add_argument("--mis", type=int, default=13, help="min_interval_size")
    ap.add_argument("-a", "--approved-interval-number", type=int, default=20)
    ap.add_argument("--mpi", type=int, default=4, help="max process intervals")
    ap.add_argument("--mcr", type=int, default=20, help="min read length after chop")
    ap.add_argument("--ocq", action="store_true", help="output chopped fq file only")
    ap.add_argument("--ct", default="all", choices=["terminal", "internal", "all"], help="selected chopped type")
    ap.add_argument("-o", "--output-prefix", default=None)
    ap.add_argument("--chunk-size", type=int, default=10000)
    ap.add_argument("-d", "--debug", action="count", default=0)
    a = ap.parse_args(sys.argv[1:] if argv is None else argv)
    if a.debug:
        logging.basicConfig(level=logging.INFO if a.debug == 1 else logging.DEBUG, format="%(message)s")
    from . import _lib
    from .chop import chop
    from .predict import SmoothParams

    _lib.require_gpu()
    sp = SmoothParams(window=a.smooth_window_size, min_interval=a.mis, approved=a.approved_interval_number,
                      max_process_intervals=a.mpi, min_read_length_after_chop=a.mcr)
    print(chop(a.pdt, a.fq, sp, a.ct, a.output_prefix, a.chunk_size, a.max_batch_size, a.threads, a.ocq))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
