#!/usr/bin/env python3
"""Headline benchmark: reads/s through `predict` (tokenise -> forward -> smooth/intervals) on
synthetic fixed-length reads, inputs resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N bench.py ...`
(one rank per GPU).  Reads shard across ranks with no data-path collective; the
only cross-rank traffic is the barrier and the max-over-ranks of the elapsed time.

A "step" is one pass of the hot path over one batch of `--batch` synthetic reads of `--read-len`
bases (BASELINE.json configs[1]: 1 M synthetic 1 kb reads on one MI355X, processed batch by batch).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# forward GEMM flops per token
FLOPS_PER_TOKEN = {
    "gemm_inproj": 2 * 256 * 768,
    "gemm_outproj": 2 * 256 * 256,
    "gemm_fc1": 2 * 256 * 1024,
    "gemm_fc2": 2 * 1024 * 256,
    "mlp_fused": 2 * 256 * 1024 + 2 * 1024 * 256,
    "mlp_inproj_fused": 2 * 256 * 1024 + 2 * 1024 * 256 + 2 * 256 * 768,   # + the next block's in_proj
    "mlp_head1_fused": 2 * 256 * 1024 + 2 * 1024 * 256 + 2 * 256 * 1024,   # + the head's linear1
    "gemm_head1": 2 * 256 * 1024,
    "gemm_head2": 2 * 1024 * 1024,
}
FLOPS_PER_TOKEN_TOTAL = 8_916_992
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA dense peak (no sparsity)
# Kernels that can be the step's dominant one (largest share of the forward time), with the HBM-side traffic per
# padded token measured by the PMC passes committed under profiles/ ((2 * FETCH_SIZE + WRITE_SIZE) KiB, gfx950
# correction applied) and the algorithmic bytes per token next to it.
ROOFLINE_KERNELS = {
    "mlp_inproj_fused": {
        "fp16x3": {"kernel": "mlp_fused_kernel<true,EPI_BIAS> (fc1 -> GELU -> fc2 -> +residual -> LayerNorm -> next block's in_proj, fp16x3)",
                   "traffic_bpt": (2 * 9603326.1 + 8241858.0) * 1024 / 2056448, "source": "profiles/r06_pmc_hbm_traffic.json"},
        "bf16x6": {"kernel": "mlp_fused6_kernel<true,EPI_BIAS> (the same chain in the error-free 3 x bf16 split: six MFMAs per fp32-equivalent FMA)"},
        "algorithmic_bpt": 2 * 1024 + 3 * 1024 + 16,  # residual stream in and out + in_proj output (fp32) + statistics
    },
    "mlp_head1_fused": {
        "fp16x3": {"kernel": "mlp_fused_kernel<true,EPI_BIAS_RELU_Q> (last block's MLP + head linear1, fp16x3)",
                   "traffic_bpt": (2 * 9682701.4 + 10298306.0) * 1024 / 2056448, "source": "profiles/r06_pmc_hbm_traffic.json"},
        "bf16x6": {"kernel": "mlp_fused6_kernel<true,EPI_BIAS_RELU_Q> (last block's MLP + head linear1, 3 x bf16 split)"},
        "algorithmic_bpt": 2 * 1024 + 4 * 1024 + 20,
    },
    "mlp_fused": {
        "fp16x3": {"kernel": "mlp_fused_kernel<true> (fc1 -> GELU -> fc2 -> +residual -> LN statistics, fp16x3)",
                   "traffic_bpt": 10692.42, "source": "profiles/r01_pmc_hbm_traffic_mlpfused.json"},
        "algorithmic_bpt": 2 * 1024 + 16,  # residual stream in and out (fp32) + statistics
    },
    "gemm_head2": {
        "fp16x3": {"kernel": "gemm6_kernel<EPI_HEAD2_FUSED,false,1> (head linear2+3, K=N=1024, fp16x3)",
                   "traffic_bpt": (2 * 13669277.2 + 128528.0) * 1024 / 2056448, "source": "profiles/r06_pmc_hbm_traffic.json"},
        "bf16x6": {"kernel": "gemm6_kernel<EPI_HEAD2_FUSED,false,0> (head linear2+3, K=N=1024, bf16x6)",
                   "traffic_bpt": 14848064128.0 / 1025024, "source": "profiles/r01_pmc_hbm_traffic.json"},
        "f32": {"kernel": "gemm_kernel<EPI_BIAS_RES_RELU,false> (head linear2, K=N=1024, fp32 MFMA)"},
        "algorithmic_bpt": 2 * 1024 * 4 + 16,  # B operand + residual, fp32
    },
}

def cpu_baseline(seqs, quals, ids, sd, budget_s: float):
    """Time the CPU oracle (torch fp32 + numpy restatement of the same path, reference batch size
    12) on a bounded sample of the same workload.  kind = "port": the reference's own `predict`
    cannot run here."""
    import torch

    from oracle import dc_oracle as o
    from oracle import forward_ref as fr

    def run(idx):
        feats = []
        for i in idx:
            rec = o.parse_fastq_record(ids[i], seqs[i].decode("latin1"), quals[i].decode("latin1"))
            feats.append(o.tokenize_and_align_labels_and_quals_ids(rec["id"], rec["seq"], rec["qual"], rec["target"]))
        out = {}
        for b0 in range(0, len(feats), 12):  # cli.py:70 default batch size
            fb = feats[b0: b0 + 12]
            batch = o.collate(fb)
            logits = fr.forward(sd, batch["input_ids"], batch["input_quals"]).numpy()
            preds = o.load_predicts_from_batch({"prediction": logits, "target": batch["labels"],
                                                "seq": batch["input_ids"], "id": batch["id"]})
            for rid, p in preds.items():
                out[rid] = (p.prediction, p.smooth_label(21), p.smooth_and_select_intervals(21, 13, 20))
        return out

    # The thread count is part of the baseline: torch's default (every host core, 128 on the GPU box) is NOT the fastest
    # setting for batches of 12 x 1 001 tokens (round 2 reported 7.3 reads/s on 128 threads, slower than the reference's
    # own prose figure).  Probe 4 / 8 / 16 / ... on one reference batch each and measure with the best.
    ncpu = os.cpu_count() or 1
    t_w = time.perf_counter()
    res = run(range(min(12, len(seqs))))  # warm-up: first-touch allocations, filter materialisation in the oracle
    spent = time.perf_counter() - t_w
    probe = {}
    for t in [x for x in (4, 8, 16, 32, 64, 128) if x <= ncpu] or [ncpu]:
        if probe and spent > budget_s:
            break  # a small --cpu-seconds must not be overrun by the probe itself
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        run(range(min(12, len(seqs))))
        dt = time.perf_counter() - t0
        spent += dt
        probe[t] = round(min(12, len(seqs)) / dt, 3)
        if probe[t] < 0.7 * max(probe.values()):
            break  # past the optimum: more threads only add synchronisation (all 256 host threads: 0.5 reads/s)
    best = max(probe, key=probe.get)
    torch.set_num_threads(best)
    per12 = 12 / probe[best]
    n_more = int(max(0.0, (budget_s - spent) / max(per12, 1e-3)) * 12) // 12 * 12
    n_more = max(0, min(n_more, (len(seqs) - 12) // 12 * 12))
    if n_more == 0:  # fewer than 24 reads sampled, or the budget went into warm-up + probe: the probe's best IS the measurement
        return probe[best], min(12, len(seqs)), best, res, probe
    t1 = time.perf_counter()
    res.update(run(range(12, 12 + n_more)))
    total = time.perf_counter() - t1
    return n_more / total, 12 + n_more, best, res, probe


def _cpu_all_cores_worker(idx: int, n_proc: int, threads: int, cpus, read_len: int, budget_s: float, conn) -> None:
    """One of the read-sharded oracle processes of the all-core CPU baseline (never touches the GPU)."""
    if cpus and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import torch

    torch.set_num_threads(threads)
    sys.path.insert(0, ROOT)
    from deepchopper_amd import synth
    from deepchopper_amd.weights import synthetic_state_dict
    from tools import bench_legs

    sd = synthetic_state_dict(0)
    seq, qual, ids = synth.synth_reads_fixed(600, read_len, seed=42, start_index=10_000 * (idx + 1))  # this process's shard
    seqs, quals = [seq[i].tobytes() for i in range(600)], [qual[i].tobytes() for i in range(600)]
    bench_legs.oracle_reads(sd, ids[:12], seqs[:12], quals[:12])  # warm-up
    conn.send("ready")
    conn.recv()  # all processes start their timed region together
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < budget_s and done + 12 <= 600:
        bench_legs.oracle_reads(sd, ids[done: done + 12], seqs[done: done + 12], quals[done: done + 12])
        done += 12
    conn.send((done, time.perf_counter() - t0))
    conn.close()


def cpu_baseline_all_cores(best_threads: int, read_len: int, budget_s: float):
    """`usable CPUs // best_threads` oracle processes, read-sharded (each on its own slice of the CPUs), timed together: the
    figure that uses the host the way `north_star` words it ("the same box's host cores, core count stated").  Usable =
    the affinity mask capped by the cgroup CPU quota (a GPU box hands a job a share of the host's 256 hardware threads)."""
    import multiprocessing as mp

    from deepchopper_amd import placement

    usable = placement.usable_cpu_count()
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    n_proc = max(1, usable // best_threads)
    layout = {"processes": n_proc, "threads_per_process": best_threads, "usable_cpus": usable, "affinity_cpus": len(allowed),
              "cgroup_cpu_quota": placement.cpu_quota()}
    if n_proc == 1:
        q = placement.cpu_quota()
        layout["reason"] = (f"cgroup quota {q:g} of {os.cpu_count()} CPUs: one {best_threads}-thread process already uses the job's whole allowance"
                            if q is not None and q < len(allowed) else
                            f"{usable} usable CPUs hold one {best_threads}-thread process")
        return None, layout
    ctx = mp.get_context("spawn")  # fresh interpreters: nothing that has initialised the GPU is forked
    procs, conns = [], []
    pin = len(allowed) >= n_proc * best_threads and placement.cpu_quota() is None
    for i in range(n_proc):
        a, b = ctx.Pipe()
        cpus = allowed[i * best_threads: (i + 1) * best_threads] if pin else None
        p = ctx.Process(target=_cpu_all_cores_worker, args=(i, n_proc, best_threads, cpus, read_len, budget_s, b), daemon=True)
        p.start()
        procs.append(p)
        conns.append(a)
    try:
        for c in conns:
            if not c.poll(180):
                raise TimeoutError("an oracle process did not get ready")
            c.recv()
        for c in conns:
            c.send("go")
        results = []
        for c in conns:
            if not c.poll(budget_s * 4 + 120):
                raise TimeoutError("an oracle process did not finish")
            results.append(c.recv())
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
    layout["pinned_slices"] = bool(pin)
    return sum(r[0] for r in results) / max(r[1] for r in results), layout


def self_launch(args) -> None:
    """`python bench.py --gpus N` without a launcher: start N workers (one per GPU, the environment
    torch.distributed.run would give them), let them run the normal N-rank path (barrier, max over ranks) and
    print rank 0's line.  DC_BENCH_SHARE_GPU=1 rehearses this on a one-GPU box: every worker uses device 0 and
    the process group runs on gloo (the default anyway)."""
    from deepchopper_amd import launcher

    share = os.environ.get("DC_BENCH_SHARE_GPU") == "1"
    procs = launcher.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, share_gpu=share)
    outs = launcher.wait_ranks(procs, timeout=float(os.environ.get("DC_BENCH_TIMEOUT", "1800")))  # a hung rank ends the job
    line = launcher.last_json_line(outs[0])
    line["launcher"] = "self (one subprocess per GPU)"
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2048, help="reads per step per GPU")
    ap.add_argument("--read-len", type=int, default=1000)
    ap.add_argument("--resident-batches", type=int, default=2, help="distinct synthetic batches kept in HBM")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--gemm-mode", choices=["f32", "bf16x6", "fp16x3"], default=os.environ.get("DC_GEMM_MODE", "fp16x3"))
    ap.add_argument("--strict-steps", type=int, default=5, help="extra untimed-in-`value` leg in the strict fp32-equivalent "
                    "mode (bf16x6) -> `value_bf16x6` (0 = skip)")
    ap.add_argument("--e2e-reads", type=int, default=1_000_000, help="FASTQ-inclusive leg through predict_stream -> `file_e2e`: "
                    "configs[1]'s stated size, ~31 s sustained (N = 1 only; 0 = skip)")
    ap.add_argument("--legs", default="c1,c3,c4,c5,ref", help="per-config legs -> `configs.*` (N = 1 only; '' = none): c1 = 512 bp, "
                    "c3 = 2 kb one-GPU slice, c4 = mixed 100 bp - 10 kb file, c5 = .fastq.gz -> chopped BGZF, ref = the reference's "
                    "default batching (-b 12, left-padded batches)")
    ap.add_argument("--c4-reads", type=int, default=40_000)
    ap.add_argument("--c5-reads", type=int, default=300_000)
    ap.add_argument("--c3-file-reads", type=int, default=200_000, help="N = 1: reads of the file-fed 2 kb leg `configs.c3_2kb_file` (0 = skip)")
    ap.add_argument("--multi-legs", default="c3,c5", help="N > 1: file-fed legs run by all ranks -> `multi_gpu.*` ('' = none): c3 = one 2 kb "
                    "FASTQ in byte-range shards, c5 = one .fastq.gz through the node's single inflater + the same reads as BGZF shards, "
                    "both -> chopped BGZF spliced in input order (tools/bench_multi.py)")
    ap.add_argument("--c3-reads-per-rank", type=int, default=100_000)
    ap.add_argument("--c5-reads-per-rank", type=int, default=150_000)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)  # parent: starts one worker per GPU before anything here touches the GPU

    from deepchopper_amd import placement

    # N > 1: a disjoint slice of its GPU's NUMA node's CPUs per rank -- before torch is imported and the GPU is touched, so
    # that helper threads inherit it and pinned staging is allocated after binding (deepchopper_amd/placement.py)
    JOB_USABLE_CPUS = placement.usable_cpu_count()  # before this rank binds itself to its slice
    placed = placement.apply_for_this_rank()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # One rank per GPU.  The process group is control plane only -- a barrier and the max of one host scalar -- and
    # BASELINE.json's north_star says "no RCCL": gloo is the default.  DC_BENCH_BACKEND=nccl runs the same two calls
    # over RCCL.  The backend is fixed BEFORE init from the environment every rank shares, so ranks cannot disagree
    # (a per-rank fallback after a failed init would leave some ranks on another group and the first barrier hanging).
    backend = os.environ.get("DC_BENCH_BACKEND", "gloo")
    if backend not in ("gloo", "nccl"):
        raise SystemExit(f"DC_BENCH_BACKEND must be gloo or nccl, not {backend!r}")
    n_dev = max(torch.cuda.device_count(), 1)
    # fewer GPUs than ranks only when rehearsing the N > 1 code path on a one-GPU box (DC_BENCH_SHARE_GPU=1)
    dev_index = local_rank % n_dev if os.environ.get("DC_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        import datetime

        import torch.distributed as dist

        kw = {"device_id": torch.device("cuda", dev_index)} if backend == "nccl" else {}
        # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."): stdout carries exactly ONE JSON
        # line, so file descriptor 1 points at stderr while the group forms (init + the first collective)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=10), **kw)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from deepchopper_amd import ops, synth
    from deepchopper_amd.model import DeepChopper
    from deepchopper_amd.predict import PredictPipeline
    from deepchopper_amd.weights import synthetic_state_dict

    sd = synthetic_state_dict(0)
    model = DeepChopper.from_state_dict(sd)
    model.net.set_gemm_mode(args.gemm_mode)
    pipe = PredictPipeline(model)
    B, L = args.batch, args.read_len

    # rank-local shard of the synthetic read set, resident in HBM before the timed region
    batches = []
    host_first = None
    for j in range(args.resident_batches):
        start = (rank * args.resident_batches + j) * B
        seq, qual, ids = synth.synth_reads_fixed(B, L, seed=42, start_index=start)
        if host_first is None:
            host_first = (seq, qual, ids)
        offs = np.arange(B + 1, dtype=np.int64) * L
        batches.append((ops.to_device(seq.reshape(-1)), ops.to_device(qual.reshape(-1)), ops.to_device(offs)))
    lens = np.full(B, L, np.int64)

    def step(j):
        s, q, o = batches[j % len(batches)]
        return pipe.run_device_batch(s, q, o, lens, want_labels=True)

    from deepchopper_amd.sharding import dist_barrier as barrier, dist_max

    for j in range(args.warmup):
        step(j)
    torch.cuda.synchronize()
    if not args.no_profile:
        model.net.set_profiling(True)
        model.net.profile_reset()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for j in range(args.steps):
        last = step(j)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = dist_max(elapsed, device="cuda" if backend == "nccl" else "cpu")  # the slowest rank defines the job's time

    fp16_overflow = model.net.overflow() if args.gemm_mode == "fp16x3" else 0  # range guard of the fp16 split
    prof = model.net.profile() if not args.no_profile else {}
    model.net.set_profiling(False)
    per_rank_s = [elapsed]
    cpusets = [placed and placed["cpuset"]]
    if world > 1:
        own = time.perf_counter() - t0  # not comparable across ranks to better than the barrier; informational
        gathered = [None] * world
        dist.all_gather_object(gathered, (round(own, 6), placed and placed["cpuset"]))
        per_rank_s = [g[0] for g in gathered]
        cpusets = [g[1] for g in gathered]

    # strict fp32-equivalent arithmetic (error-free 3-way bf16 split), same config, a few steps: so that the
    # driver's record also carries the throughput of the mode that needs no range guard
    strict = None
    if args.strict_steps > 0 and args.gemm_mode != "bf16x6":
        model.net.set_gemm_mode("bf16x6")
        step(0)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for j in range(args.strict_steps):
            step(j)
        torch.cuda.synchronize()
        barrier()
        dt = dist_max(time.perf_counter() - t1, device="cuda" if backend == "nccl" else "cpu")
        strict = {"value": round(world * B * args.strict_steps / dt, 2), "steps": args.strict_steps,
                  "ms_per_step": round(dt / args.strict_steps * 1e3, 3)}
        model.net.set_gemm_mode(args.gemm_mode)

    # N > 1: the flows in which eight GPUs can fail to scale (the host is the limiter) -- file-fed, run by ALL ranks,
    # reported beside the resident headline, never `value` (tools/bench_multi.py)
    multi = None
    if world > 1 and args.multi_legs:
        from tools import bench_multi

        my_cpus = placed and placed["cpuset"]
        bench_multi.JOB_CPU_BUDGET = (placed and placed.get("cpu_budget")) or JOB_USABLE_CPUS
        multi = {}
        wanted = [x for x in args.multi_legs.split(",") if x]
        if "c3" in wanted:
            leg = bench_multi.c3_sharded_leg(pipe, sd, rank, world, args.c3_reads_per_rank, cpuset=my_cpus)
            if rank == 0:
                multi["c3_2kb_file_sharded"] = leg
        if "c5" in wanted:
            # the c5 legs bring their own pipeline (token-gated weights): release this one's handle first -- its workspace is
            # tens of GB, and ranks that share a card in a rehearsal would not fit two per rank (six ranks: out of memory)
            batches.clear()
            model.net.close()
            torch.cuda.empty_cache()
            legs5 = bench_multi.c5_legs(rank, world, args.c5_reads_per_rank, threads=4, cpuset=my_cpus,
                                        keep_dir=os.environ.get("DC_BENCH_KEEP_DIR") or None)
            if rank == 0:
                multi.update(legs5 or {})

    if rank == 0:
        tokens_per_launch = B * (L + 1)
        stages = {}
        for name, (ms, cnt) in sorted(prof.items()):
            if cnt == 0:
                continue
            avg_ms = ms / cnt
            ent = {"launches": cnt, "avg_ms": round(avg_ms, 4), "share": round(ms / max(prof["forward"][0], 1e-9), 4)}
            if name in FLOPS_PER_TOKEN:
                ent["tflops"] = round(FLOPS_PER_TOKEN[name] * tokens_per_launch / (avg_ms * 1e-3) / 1e12, 2)
            stages[name] = ent
        roof = None
        cands = [k for k in ROOFLINE_KERNELS if k in stages]
        if cands:
            # the dominant kernel = the one with the largest share of the forward time (all its launches of a step)
            dom = max(cands, key=lambda k: stages[k]["share"])
            info = ROOFLINE_KERNELS[dom]
            mode = info.get(args.gemm_mode, {})
            ach = FLOPS_PER_TOKEN[dom] * tokens_per_launch / (stages[dom]["avg_ms"] * 1e-3) / 1e12
            # f32: fp32 MFMA peak.  bf16x6 / fp16x3: the 16-bit MFMA dense peak; `executed_frac` counts the 6x / 3x
            # MFMA work actually issued per algorithmic multiply-add
            peak = PEAK_F32_MFMA_TFLOPS if args.gemm_mode == "f32" else PEAK_BF16_MFMA_TFLOPS
            roof = {"bound": "mfma", "kernel": mode.get("kernel", dom), "stage": dom, "achieved": round(ach, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                    "avg_launch_ms": stages[dom]["avg_ms"], "launches_per_step": stages[dom]["launches"] // max(args.steps, 1),
                    "share_of_forward": stages[dom]["share"],
                    "algorithmic_flops_per_launch": FLOPS_PER_TOKEN[dom] * tokens_per_launch}
            if "traffic_bpt" in mode:
                padded = -(-(B * (-(-(L + 1) // 4) * 4) + 32) // 256) * 256  # a read owns ceil4(tokens) slots; 32 slots of slack; 256-token tiles
                roof["traffic"] = int(mode["traffic_bpt"] * padded)
                roof["traffic_kind"] = "constant from the committed PMC profile (bytes per padded token x this launch's tokens), not measured in this run"
                roof["traffic_source"] = mode["source"] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
                roof["algorithmic_hbm_bytes_per_launch"] = int(padded * info["algorithmic_bpt"])
            if args.gemm_mode in ("bf16x6", "fp16x3"):
                mult = 6 if args.gemm_mode == "bf16x6" else 3
                roof["mfma_per_algorithmic_fma"] = mult
                roof["executed_frac"] = round(mult * ach / peak, 4)
                roof["vs_fp32_mfma_peak"] = round(ach / PEAK_F32_MFMA_TFLOPS, 4)
        if "forward" in stages:
            stages["forward"]["tflops_all_gemm_flops"] = round(
                FLOPS_PER_TOKEN_TOTAL * tokens_per_launch / (stages["forward"]["avg_ms"] * 1e-3) / 1e12, 2)

        cpu = None
        f1 = None
        if world == 1 and args.cpu_seconds > 0:
            seq, qual, ids = host_first
            n_s = min(B, 600)
            seqs = [seq[i].tobytes() for i in range(n_s)]
            quals = [qual[i].tobytes() for i in range(n_s)]
            rate, n_done, cores, ref, probe = cpu_baseline(seqs, quals, ids[:n_s], sd, args.cpu_seconds)
            try:
                all_rate, layout = cpu_baseline_all_cores(cores, L, args.cpu_seconds)
            except Exception as e:  # a baseline leg must never take the bench line down with it
                all_rate, layout = None, {"error": f"{type(e).__name__}: {e}"}
            cpu = {"value": round(rate, 3), "unit": "reads/s", "cores": cores, "kind": "port",
                   "sample": f"{max(n_done - 12, n_done if n_done <= 12 else 0)} of the same synthetic {L}-bp reads (after a 12-read "
                             f"warm-up and the thread probe), oracle tokenise+forward(torch fp32, batch 12)+smooth",
                   # the same oracle, read-sharded over every CPU this job may use (processes x threads below); None when
                   # that is one process, i.e. `value` already uses them all.  A baseline, never credit.
                   "value_all_cores": None if all_rate is None else round(all_rate, 3), "all_cores_layout": layout,
                   # thread-count SELECTION only: one 12-read batch per setting, right after the warm-up -- not a sustained rate
                   # (r05: 45.95 on the probe, 36.25 sustained at the same 16 threads); `value` is the sustained sample
                   "thread_probe_one_batch_reads_per_s": {str(k): v for k, v in probe.items()},
                   "value_is": "sustained: every sampled read after the warm-up and the probe, timed as one run" if n_done > 12 else
                               "the probe's best (the budget allowed no sustained sample)",
                   "host_cores": os.cpu_count(),
                   "reference_prose_cpu_reads_per_s": [46, 139],  # docs/faq.md:78-80 (2-6 h per 1 M reads), context only
                   "note": "CPU restatement (oracle/), not the reference binary: its backbone cannot be loaded here (SURVEY 8c)"}
            # per-base F1 of the GPU labels against the oracle's on the sampled reads (expected 1.0)
            res = step(0)
            torch.cuda.synchronize()
            o = res["smooth"]["offsets"].cpu().numpy()
            pred = res["smooth"]["pred"].cpu().numpy()
            tp = fp = fn = 0
            same_iv = True
            iv = res["smooth"]["intervals"].cpu().numpy()
            niv = res["smooth"]["n_intervals"].cpu().numpy()
            for r in range(n_done):
                g = pred[o[r]: o[r + 1]]
                e, _, eiv = ref[ids[r]]
                tp += int(np.sum((g == 1) & (e == 1)))
                fp += int(np.sum((g == 1) & (e == 0)))
                fn += int(np.sum((g == 0) & (e == 1)))
                same_iv &= [tuple(x) for x in iv[r, : niv[r]].tolist()] == eiv
            f1 = {"per_base_f1_vs_oracle": round(2 * tp / max(2 * tp + fp + fn, 1), 6), "reads": n_done,
                  "intervals_identical": bool(same_iv)}

        e2e = None
        configs = {}
        if world == 1:
            from tools import bench_legs

            def leg(fn, *a, **kw):
                """A leg reports beside the headline; it must never take the line down with it (a full /tmp, a missing gzip ...)."""
                try:
                    return fn(*a, **kw)
                except Exception as e:
                    return {"error": f"{type(e).__name__}: {e}"}

            legs = [x for x in args.legs.split(",") if x]
            if args.e2e_reads > 0:
                e2e = leg(bench_legs.file_e2e_leg, pipe, sd, args.e2e_reads, L, B)
            if "c1" in legs:
                configs["c1_512bp"] = leg(bench_legs.resident_leg, pipe, sd, 1000, 512, 40,
                                          "configs[0]: 1 000 synthetic 512-bp reads = one step (the whole config), HBM-resident")
                configs["c1_512bp"]["batch_4096"] = leg(bench_legs.resident_leg, pipe, sd, 4096, 512, 10,
                                                        "512-bp reads in batches of 4 096 (the shape's best batch)", check_reads=0)
                if cpu is not None:  # the CPU restatement on C1 as well, same thread layout as the C2 figure
                    configs["c1_512bp"]["cpu_baseline"] = leg(bench_legs.cpu_baseline_leg, sd, 512, cpu["cores"], min(8.0, args.cpu_seconds))
            if "c3" in legs and args.c3_file_reads > 0:
                from tools import bench_multi

                configs["c3_2kb_file"] = leg(bench_multi.c3_sharded_leg, pipe, sd, 0, 1, args.c3_file_reads)
            if "c3" in legs:
                configs["c3_2kb_1gpu_slice"] = leg(
                    bench_legs.resident_leg, pipe, sd, 1024, 2000, 12,
                    "configs[2]: synthetic 2 kb reads, ONE GPU's slice of the read-sharded run (1 024 reads per step; 8 ranks run "
                    "8 of these with no exchange), HBM-resident")
            if "c4" in legs:
                configs["c4_mixed"] = leg(bench_legs.mixed_leg, pipe, sd, args.c4_reads)
            if "c5" in legs:
                configs["c5_gz_chop"] = leg(bench_legs.gz_chop_leg, args.c5_reads, threads=4)
                configs["c5_gz_chop_refbatch"] = leg(bench_legs.gz_chop_leg, args.c5_reads // 2, threads=4, reference_batch=12)
            if "ref" in legs:
                configs["reference_b12"] = leg(bench_legs.reference_b12_leg, pipe, sd, 60_000)
                configs["reference_b12_nanopore"] = leg(bench_legs.reference_b12_leg, pipe, sd, 60_000, mix="nanopore")
                configs["reference_b12_nanopore_pt_files"] = leg(bench_legs.reference_b12_leg, pipe, sd, 60_000, mix="nanopore", write_pt=True)
        value = world * B * args.steps / elapsed
        print(json.dumps({
            "metric": "reads/sec through predict, 1 kb reads",
            "value": round(value, 2),
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x6": "f32 (GEMMs as error-free 3xbf16 split on bf16 MFMA, fp32 accumulate)",
                      "fp16x3": "f32 (GEMMs as 2xfp16 split on fp16 MFMA, 22-bit products, fp32 accumulate, range-guarded)"}[args.gemm_mode],
            "data": "synthetic",
            "config": {"workload": f"configs[1]: synthetic {L}-bp reads, {world}xMI355X, batches of {B} reads per GPU "
                                   f"(tokenise -> HyenaDNA forward -> argmax/smooth/intervals), synthetic weights",
                       "reads_per_step_per_gpu": B, "read_len": L, "tokens_per_read": L + 1,
                       "parallelism": f"read-sharded x{world}, no collectives",
                       "control_plane": (backend if world > 1 else "none") + " (barrier + max of one host scalar)"},
            "roofline": roof,
            "value_bf16x6": strict["value"] if strict else None,
            "bf16x6_leg": strict,
            "file_e2e": e2e,
            "configs": configs or None,
            "multi_gpu": multi,
            "per_rank_seconds": per_rank_s,
            "per_rank_cpuset": cpusets,
            "cpu_budget": placed and {"cpus_worth_of_time": placed.get("cpu_budget"), "slices_cut_to_it": placed.get("capped_to_budget")},
            "cpu_baseline": cpu,
            "parity": f1,
            "fp16_overflow_events": fp16_overflow,
            "stages": stages,
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
