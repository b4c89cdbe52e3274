"""bench.py's file-fed legs for N > 1 ranks: the flows in which eight GPUs CAN fail to scale.

The resident headline cannot: every rank steps through batches that already sit in its HBM.  BASELINE configs 3-5 are the
multi-GPU configs and SURVEY 8(e) names the host -- FASTQ parse, inflate, BGZF deflate, PCIe staging -- as their limiter
(the reference's flow being replaced: Lightning's read-sharded predict, deepchopper/cli.py:126-152, and the chop driver's
chunked loop, src/bin/predict.rs:275-316).  Three legs, none of them `value`:

    c3_2kb_file_sharded      one 2 kb FASTQ file, contiguous byte-range shards (`cli.rank_chunks`): every bench rank parses and
                             `predict_stream`s its own shard -> labels + intervals on the host.  In process.
    c5_gz_single_inflater    one plain .fastq.gz -> the node's SINGLE inflater (`launcher.ChunkFeeder` on rank 0, over FIFOs) deals
                             chunks to every rank -> predict -> chop -> per-GPU BGZF parts spliced in input order
                             (`cli.predict_chop_as_external_rank`: what `torch.distributed.run ... cli predict X.gz -g N --chop`
                             runs; the self-launching `predict -g N` has a parent process in rank 0's place, tools/host_budget.py).
    c5_bgzf_sharded          the same reads as BGZF: no feeder, every rank inflates its own block range.

Each reports, per rank: reads, reads/s, CPU seconds per 1 000 reads (the worker process as a whole, its BGZF deflate threads,
its pipe reads; the parent), the rank's cpuset -- and ONE `host_bound` verdict: the CPU the flow consumed against the CPUs'
worth of time the job may use (affinity mask capped by the cgroup quota, `placement.usable_cpu_count`).  On a shared card
(DC_BENCH_SHARE_GPU=1 rehearsals) the wall times mean nothing; the CPU seconds per read carry over.
"""
from __future__ import annotations

import os
import shutil
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# reads/s of ONE MI355X on the nanopore-like flow when the GPU is the limiter (driver record BENCH_r05: configs.c5_gz_chop
# 26 783): what the host must sustain per GPU for the flow to stay GPU-bound at N GPUs
GPU_BOUND_C5_READS_PER_S = 26_000.0
GPU_BOUND_C3_READS_PER_S = 15_300.0  # configs.c3_2kb_1gpu_slice, same record


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def _barrier():
    d = _dist()
    if d is not None:
        d.barrier()


def _bcast(obj, src: int = 0):
    d = _dist()
    if d is None:
        return obj
    box = [obj]
    d.broadcast_object_list(box, src=src)
    return box[0]


def _gather(obj, world: int):
    d = _dist()
    if d is None:
        return [obj]
    out = [None] * world
    d.all_gather_object(out, obj)
    return out


def _thread_cpu() -> dict:
    """CPU seconds of every thread of this process so far: {tid: (name, seconds)} from /proc (who burns the host budget?)."""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                with open(f"/proc/self/task/{tid}/stat") as fh:
                    st = fh.read()
                name = st[st.index("(") + 1: st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                out[int(tid)] = (name, (int(f[11]) + int(f[12])) / tick)  # utime + stime
            except (OSError, ValueError):
                pass
    except OSError:
        pass
    return out


def _top_threads(before: dict, after: dict, k: int = 5) -> list:
    import threading

    py = {t.native_id: t.name for t in threading.enumerate()}  # Python threads by name; the rest are the runtime's / the library's
    d = [((f"py:{py[t]}" if t in py else f"native:{after[t][0]}"), round(after[t][1] - before.get(t, ("", 0.0))[1], 2)) for t in after]
    d.sort(key=lambda x: -x[1])
    return [f"{n}: {c}" for n, c in d[:k] if c > 0.0]


def _per_k(cpu_s: float, reads: int) -> float:
    return round(1000.0 * cpu_s / max(reads, 1), 4)


JOB_CPU_BUDGET = None  # CPUs' worth of time the WHOLE job may use; set by bench.py from the placement plan (a rank that has been
                       # bound to its slice sees only the slice in its own affinity mask)


def host_verdict(cpu_s_total: float, wall_s: float, reads: int, world: int, gpu_bound_rate: float) -> dict:
    """Was the flow at the host's limit, and would it be at N GPU-bound GPUs?  `cores_used` = CPU seconds consumed per second of
    wall (all processes of the flow); `budget` = CPUs' worth of time the job may use."""
    from deepchopper_amd import placement

    budget = JOB_CPU_BUDGET or placement.usable_cpu_count()
    used = cpu_s_total / max(wall_s, 1e-9)
    need = cpu_s_total / max(reads, 1) * gpu_bound_rate * world  # cores needed to feed `world` GPUs that are each GPU-bound
    return {"cpu_s_total": round(cpu_s_total, 2), "cores_used": round(used, 2), "cpu_budget_cores": budget,
            "cgroup_cpu_quota": placement.cpu_quota(),
            "cores_needed_at_gpu_bound_rate": round(need, 2),
            "gpu_bound_reads_per_s_per_gpu_assumed": gpu_bound_rate,
            "host_bound": bool(used >= 0.9 * budget or need > budget),
            "why": ("the flow used >= 90 % of its CPU allowance" if used >= 0.9 * budget else
                    "feeding every GPU at its GPU-bound rate needs more CPU than the job may use" if need > budget else
                    "CPU per read x the GPU-bound rate x N fits the allowance")}


# ------------------------------------------------------------------------------------------------------------
# configs[2]: one 2 kb FASTQ, contiguous byte-range shards, every rank streams its own
# ------------------------------------------------------------------------------------------------------------
def _failed(err: str | None, world: int) -> str | None:
    """Collective: every rank learns whether ANY rank failed its part of a phase (doubles as the phase's barrier), so that
    all ranks leave a leg together instead of one raising while the others wait at the next barrier."""
    errs = _gather(err, world)
    bad = [f"rank {r}: {e}" for r, e in enumerate(errs) if e]
    return "; ".join(bad) if bad else None


def _try(fn):
    try:
        return fn(), None
    except Exception as e:
        return None, f"{type(e).__name__}: {e}"


def c3_sharded_leg(pipe, sd, rank: int, world: int, reads_per_rank: int, read_len: int = 2000, batch: int = 1024,
                   cpuset: str | None = None, check_reads: int = 12) -> dict | None:
    """Returns the leg's report on rank 0 (None elsewhere).  Every rank writes its own part of the ONE file (fixed-length
    records: offsets are known), then reads back whatever byte range `rank_chunks` gives it.  world = 1: the same leg without a
    group (bench.py's `configs.c3_2kb_file` at N = 1)."""
    import torch

    from deepchopper_amd import synth
    from deepchopper_amd.cli import rank_chunks

    tmp = _bcast(tempfile.mkdtemp(prefix="dc_c3n_") if rank == 0 else None)
    path = os.path.join(tmp, "reads_2kb.fq")
    rec = 1 + 14 + 1 + read_len + 3 + read_len + 1
    total = reads_per_rank * world
    cap = batch * (read_len + 1)
    try:
        t_gen = time.perf_counter()

        def create():
            if rank == 0:
                with open(path, "wb") as fh:
                    fh.truncate(total * rec)

        def write_part():
            fd = os.open(path, os.O_WRONLY)
            try:
                for b0 in range(0, reads_per_rank, 20_000):
                    n = min(20_000, reads_per_rank - b0)
                    g0 = rank * reads_per_rank + b0
                    buf = synth.synth_fastq_fixed_fast(n, read_len, 42, g0)
                    os.pwrite(fd, memoryview(buf), g0 * rec)
            finally:
                os.close(fd)

        def warm():
            for i, _ in enumerate(pipe.predict_stream(rank_chunks(path, rank, world, batch, cap))):  # staging, copy stream
                if i >= 2:
                    break
            torch.cuda.synchronize()

        for phase in (create, write_part, warm):
            bad = _failed(_try(phase)[1], world)
            if bad:
                return {"error": f"{phase.__name__}: {bad}"} if rank == 0 else None
        gen_s = time.perf_counter() - t_gen
        th0 = _thread_cpu()
        c0, t0 = time.process_time(), time.perf_counter()
        marks = [(0, t0)]
        state = {"n": 0, "n_iv": 0, "first": None}

        def stream():
            for res in pipe.predict_stream(rank_chunks(path, rank, world, batch, cap)):
                if state["first"] is None and rank == 0 and check_reads:
                    o = res["offsets"]
                    k = min(check_reads, len(res["names"]))
                    state["first"] = {
                        "names": res["names"][:k],
                        "seq": [bytes(res["seq"][res["in_offsets"][r]: res["in_offsets"][r + 1]]) for r in range(k)],
                        "qual": [bytes(res["qual"][res["in_offsets"][r]: res["in_offsets"][r + 1]]) for r in range(k)],
                        "got": {res["names"][r]: (np.array(res["pred"][o[r]: o[r + 1]]),
                                                  res["intervals"][r, : res["n_intervals"][r]].tolist()) for r in range(k)}}
                state["n"] += len(res["names"])
                state["n_iv"] += int(res["n_intervals"].sum())
                marks.append((state["n"], time.perf_counter()))
            torch.cuda.synchronize()

        _, err = _try(stream)
        own_s = time.perf_counter() - t0
        cpu_s = time.process_time() - c0
        bad = _failed(err, world)  # (the barrier that ends the timed region)
        wall = time.perf_counter() - t0
        if bad:
            return {"error": f"stream: {bad}"} if rank == 0 else None
        n, first = state["n"], state["first"]

        def rate_between(lo, hi):
            a = next(m for m in marks if m[0] >= lo)
            b = next(m for m in reversed(marks) if m[0] <= hi)
            return round((b[0] - a[0]) / (b[1] - a[1]), 1) if b[0] > a[0] else None

        tenth = max(n // 10, 1)
        mine = {"rank": rank, "reads": n, "seconds": round(own_s, 3), "reads_per_s": round(n / max(own_s, 1e-9), 1),
                "cpu_s": round(cpu_s, 3), "cpu_s_per_1000_reads": _per_k(cpu_s, n), "intervals": state["n_iv"], "cpuset": cpuset,
                "cpu_s_top_threads": _top_threads(th0, _thread_cpu()),
                "first_tenth_reads_per_s": rate_between(0, tenth), "last_tenth_reads_per_s": rate_between(n - tenth, n)}
        ranks = _gather(mine, world)
        walls = _gather(wall, world)
    finally:
        _barrier()
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    if rank != 0:
        return None
    n_all = sum(r["reads"] for r in ranks)
    wall = max(walls)
    out = {"workload": f"configs[2]: ONE FASTQ file of {total} synthetic {read_len}-bp reads" +
                       (f", contiguous byte-range shards (cli.rank_chunks), every rank" if world > 1 else "") +
                       f": parse -> H2D -> tokenise/forward/smooth -> D2H labels + intervals (predict_stream), {batch} reads per step",
           "value": round(n_all / wall, 1), "unit": "reads/s", "Mbases_per_s": round(n_all * read_len / wall / 1e6, 2), "reads": n_all,
           "seconds": round(wall, 3), "file_GB": round(total * rec / 1e9, 3), "generate_s": round(gen_s, 1),
           "per_rank": ranks,
           "cpu_s_per_1000_reads": {"worker": _per_k(sum(r["cpu_s"] for r in ranks), n_all)},
           "host": host_verdict(sum(r["cpu_s"] for r in ranks), wall, n_all, world, GPU_BOUND_C3_READS_PER_S)}
    check = {"reads_counted": n_all == total, "every_rank_got_reads": all(r["reads"] > 0 for r in ranks)}
    if first is not None:
        from tools import bench_legs

        exp = bench_legs.oracle_reads(sd, first["names"], first["seq"], first["qual"])
        check.update(bench_legs.f1_check(exp, first["got"]))
    out["check"] = check
    return out


# ------------------------------------------------------------------------------------------------------------
# configs[4]: .fastq.gz (single inflater -> N workers) and BGZF (sharded) -> chopped BGZF, through the command itself
# ------------------------------------------------------------------------------------------------------------
def _write_c5_parts(tmp: str, rank: int, reads_per_rank: int):
    """This rank's share of the nanopore-like reads as (a) one gzip member, (b) a run of BGZF blocks.  Concatenated in rank
    order they are one valid .fastq.gz of `world` large members (each decoded on several threads by the single inflater,
    csrc/fast_inflate.h) and one valid BGZF file.  Returns (bases, head records of rank 0 for the oracle check)."""
    from deepchopper_amd import synth
    from deepchopper_amd.fastq import BgzfWriter

    bases = 0
    head = None
    gz = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(os.path.join(tmp, f"part{rank:03d}.gz"), "wb") as fg, BgzfWriter(os.path.join(tmp, f"part{rank:03d}.bgzf"), threads=2) as fb:
        for b0 in range(0, reads_per_rank, 25_000):
            n = min(25_000, reads_per_rank - b0)
            g0 = rank * reads_per_rank + b0
            h, s, q = synth.synth_nanopore_like(n, seed=42 + g0, start_index=g0)
            if head is None:
                head = (h[:24], s[:24], q[:24])
            bases += sum(len(x) for x in s)
            text = synth.fastq_text(h, s, q)
            fg.write(gz.compress(text))
            fb.write(text)
        fg.write(gz.flush())
    return bases, head


def c5_legs(rank: int, world: int, reads_per_rank: int, threads: int = 4, cpuset: str | None = None,
            keep_dir: str | None = None) -> dict | None:
    """Both config-5 legs, run BY the bench ranks through the product's own entry for ranks an external launcher started
    (`cli.predict_chop_as_external_rank`: what `torch.distributed.run ... -m deepchopper_amd.cli predict X -g N --chop` runs).
    Every rank generates its share of the input; rank 0 assembles the two files.  Returns
    {"c5_gz_single_inflater": ..., "c5_bgzf_sharded": ...} on rank 0, None elsewhere.  `keep_dir`: inputs and outputs are left
    there (the GPU test compares the outputs with `-g 1`'s)."""
    import torch

    from deepchopper_amd import chop as chop_mod
    from deepchopper_amd import synth
    from deepchopper_amd.cli import predict_chop_as_external_rank
    from deepchopper_amd.model import DeepChopper
    from deepchopper_amd.predict import PredictPipeline
    from deepchopper_amd.weights import token_gated_state_dict

    if rank == 0 and keep_dir:
        os.makedirs(keep_dir, exist_ok=True)
    tmp = _bcast(tempfile.mkdtemp(prefix="dc_c5n_", dir=keep_dir) if rank == 0 else None)
    legs = {}
    total = reads_per_rank * world
    try:
        t0 = time.perf_counter()
        made, err = _try(lambda: _write_c5_parts(tmp, rank, reads_per_rank))
        bad = _failed(err, world)
        if bad:
            return {"c5_gz_single_inflater": {"error": f"generating the input: {bad}"}} if rank == 0 else None
        bases, head = made
        all_bases = sum(_gather(bases, world))
        gz, bg = os.path.join(tmp, "hct116_like.fastq.gz"), os.path.join(tmp, "hct116_like.bgzf.fastq.gz")

        def assemble():
            if rank != 0:
                return
            for dst, ext in ((gz, "gz"), (bg, "bgzf")):
                with open(dst, "wb") as out:
                    for r in range(world):
                        part = os.path.join(tmp, f"part{r:03d}.{ext}")
                        with open(part, "rb") as src:
                            shutil.copyfileobj(src, out, 8 << 20)
                        os.remove(part)

        sd = token_gated_state_dict(0)
        box = {}

        def warm_up():
            # this leg's own pipeline (token-gated weights, so that predict -> chop really chops), warmed on a private file:
            # the handle, the FFT size classes, pinned staging, the BGZF pool
            box["pipe"] = PredictPipeline(DeepChopper.from_state_dict(sd))
            warm = os.path.join(tmp, f"warm{rank}.fastq")
            h, s, q = synth.synth_nanopore_like(12_000, seed=7 + rank)
            with open(warm, "wb") as fh:
                fh.write(synth.fastq_text(h, s, q))
            final, _, _ = chop_mod.predict_chop(warm, box["pipe"], threads=threads, output_prefix=os.path.join(tmp, f"warm{rank}"))
            os.remove(final)
            os.remove(warm)
            torch.cuda.synchronize()

        for phase in (assemble, warm_up):
            bad = _failed(_try(phase)[1], world)
            if bad:
                return {"c5_gz_single_inflater": {"error": f"{phase.__name__}: {bad}"}} if rank == 0 else None
        gen_s = time.perf_counter() - t0
        pipe = box["pipe"]
        for name, path, what in (
                ("c5_gz_single_inflater", gz,
                 f"configs[4] flow at {world} ranks: ONE plain .fastq.gz ({world} members, nanopore-like reads, adapters planted) -> rank 0 "
                 "inflates it once (launcher.ChunkFeeder) and deals chunks to every rank over FIFOs -> predict -> chop -> per-GPU BGZF "
                 "parts -> spliced in input order (cli.predict_chop_as_external_rank)"),
                ("c5_bgzf_sharded", bg,
                 "the same reads as BGZF: no feeder, every rank inflates its own contiguous block range; parts concatenated rank by rank")):
            try:
                _barrier()
                th0 = _thread_cpu()
                c0, t1 = time.process_time(), time.perf_counter()
                rep = predict_chop_as_external_rank(path, pipe, rank, world, threads=threads, chop_output=os.path.join(tmp, name), quiet=True)
                torch.cuda.synchronize()
                wall = time.perf_counter() - t1
                cpu = time.process_time() - c0
                feeder_cpu = float(rep.get("feeder_cpu_s") or 0.0)
                mine = {"rank": rank, "reads": rep["reads"], "records": rep["records"], "stream_s": rep.get("stream_s"),
                        "reads_per_s": round(rep["reads"] / max(rep.get("stream_s") or wall, 1e-9), 1),
                        "cpu_s": {"worker": round(cpu - feeder_cpu, 3), "bgzf_deflate": round(float(chop_mod.LAST_WRITER_STATS.get("deflate_cpu_s", 0.0)), 3),
                                  "pipe_read": rep.get("pipe_read_cpu_s", 0.0), "feeder": feeder_cpu},
                        "cpu_s_top_threads": _top_threads(th0, _thread_cpu()),
                        "cpuset": cpuset, "feeder_cpuset": rep.get("feeder_cpuset"), "input": rep["input"], "merge_s": rep.get("merge_s"),
                        "final": rep.get("final")}
                ranks = _gather(mine, world)
                walls = _gather(wall, world)
                if rank != 0:
                    continue
                wall = max(walls)
                tot = {k: sum(r["cpu_s"][k] for r in ranks) for k in ("worker", "bgzf_deflate", "pipe_read", "feeder")}
                cs = tot.__getitem__
                for r in ranks:
                    r["cpu_s_per_1000_reads"] = {k: _per_k(v, r["reads"]) for k, v in r.pop("cpu_s").items() if k != "feeder"}
                final = ranks[0].pop("final")
                for r in ranks[1:]:
                    r.pop("final", None)
                leg = {"workload": what, "input": ranks[0]["input"], "value": round(total / wall, 1), "unit": "reads/s",
                       "Mbases_per_s": round(all_bases / wall / 1e6, 2), "reads": rep["total_reads"], "records_out": rep["total_records"],
                       "seconds": round(wall, 3), "merge_s": ranks[0].get("merge_s"), "mean_bp": round(all_bases / total, 1),
                       "input_MB": round(os.path.getsize(path) / 1e6, 1), "output_MB": round(os.path.getsize(final) / 1e6, 1),
                       "bgzf_threads_per_rank": threads, "generate_s": round(gen_s, 1), "per_rank": ranks,
                       "cpu_s_per_1000_reads": {"worker": _per_k(cs("worker"), total), "bgzf_deflate": _per_k(cs("bgzf_deflate"), total),
                                                "pipe_read": _per_k(cs("pipe_read"), total), "parent": _per_k(cs("feeder"), total),
                                                "note": "worker = the rank's whole process (parse or pipe read, staging, launches, chop, BGZF deflate) "
                                                        "without the feeder's threads; parent = the single inflater (rank 0's feeder threads + the "
                                                        "reader's native inflate helpers)"},
                       "host": host_verdict(cs("worker") + cs("feeder"), wall, total, world, GPU_BOUND_C5_READS_PER_S),
                       "output": final if keep_dir else None}
                check = {"reads_in_match": rep["total_reads"] == total, "every_rank_got_reads": all(r["reads"] > 0 for r in ranks)}
                if name == "c5_gz_single_inflater":
                    from tools import bench_legs

                    check.update(bench_legs._chop_head_check(sd, head, final, 1))
                leg["check"] = check
                legs[name] = leg
                if not keep_dir:
                    os.remove(final)
            except Exception as e:  # a leg never takes the line down (every rank raises together: the rendezvous is inside)
                if rank == 0:
                    legs[name] = {"error": f"{type(e).__name__}: {e}"}
    finally:
        _barrier()
        if rank == 0 and not keep_dir:
            shutil.rmtree(tmp, ignore_errors=True)
    return legs if rank == 0 else None
