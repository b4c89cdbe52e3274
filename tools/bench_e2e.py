"""End-to-end rates either side of the hot path, on files.

* default: file-to-intervals throughput of the streaming predict path (FASTQ on disk -> native reader -> H2D ->
  tokenise/forward/smooth -> D2H labels + intervals) on fixed-length reads, next to the parse-only rate of the host
  reader -- the PCIe- and parse-inclusive figure DESIGN.md quotes beside bench.py's HBM-resident `value`.

      python tools/bench_e2e.py --reads 100000 --bp 1000 --container plain|gzip|bgzf

* `--chop`: BASELINE config 5's flow -- a nanopore-like `.fastq.gz` (log-normal lengths, ONT-style headers, adapters
  planted as N runs, `weights.token_gated_state_dict` so that predict really finds them) through the fused
  `predict -> chop` (no `.pt` files) into a chopped BGZF FASTQ; reports reads/s, bases/s, peak host RSS and what
  became of the reads.  `--gpus N` runs the command's own launcher (one worker per GPU; DC_SHARE_GPU=1 rehearses
  it on one GPU), where plain gzip goes through the parent's single inflater.

      python tools/bench_e2e.py --chop --reads 1000000 --container gzip [--gpus 2]
"""
import argparse
import gzip
import json
import os
import resource
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np


def _opener(container, path):
    from deepchopper_amd.fastq import BgzfWriter

    return {"plain": lambda: open(path, "wb"), "gzip": lambda: gzip.open(path, "wb", compresslevel=1),
            "bgzf": lambda: BgzfWriter(path, threads=8, level=1)}[container]()


def _count_output(path):
    """Records of the chopped file by kind (terminal / internal chop, passed through)."""
    from deepchopper_amd.fastq import FastqReader

    n = n_t = n_i = bases = 0
    with FastqReader(path) as rd:
        for c in rd.chunks(65536, 256 << 20):
            raw, o = c._names.tobytes(), c._name_offsets
            n += c.n
            bases += int(c.offsets[c.n])
            for k in range(c.n):
                tail = raw[o[k + 1] - 2: o[k + 1]]
                n_t += tail == b"|T"
                n_i += tail == b"|I"
    return {"records": n, "terminal_chops": int(n_t), "internal_chop_segments": int(n_i), "passed_through": int(n - n_t - n_i),
            "bases_out": bases}


def chop_bench(args):
    from deepchopper_amd import synth
    from deepchopper_amd.weights import save_state_dict, token_gated_state_dict

    tmp = tempfile.mkdtemp(prefix="dc_c5_")
    path = os.path.join(tmp, "hct116_like.fastq.gz" if args.container != "plain" else "hct116_like.fastq")
    t0 = time.time()
    n_bases = 0
    with _opener(args.container, path) as fh:
        for b0 in range(0, args.reads, 50_000):
            n = min(50_000, args.reads - b0)
            h, s, q = synth.synth_nanopore_like(n, seed=42 + b0, start_index=b0)
            n_bases += sum(len(x) for x in s)
            fh.write(synth.fastq_text(h, s, q))
    gen_s = time.time() - t0
    ckpt = os.path.join(tmp, "ckpt")
    save_state_dict(token_gated_state_dict(0), os.path.join(ckpt, "model.safetensors"))
    out = {"flow": "fused predict -> chop (no .pt)", "container": args.container, "reads": args.reads, "bases": n_bases,
           "mean_len": round(n_bases / args.reads, 1), "file_MB": round(os.path.getsize(path) / 1e6, 1), "gen_s": round(gen_s, 1),
           "gpus": args.gpus, "bgzf_threads": args.threads}
    os.chdir(tmp)
    if args.gpus == 1 and not args.cli:
        import torch

        from deepchopper_amd.chop import predict_chop
        from deepchopper_amd.model import DeepChopper
        from deepchopper_amd.predict import PredictPipeline

        torch.cuda.set_device(0)
        pipe = PredictPipeline(DeepChopper.from_pretrained(ckpt))
        warm = os.path.join(tmp, "warm.fastq")
        h, s, q = synth.synth_nanopore_like(20_000, seed=7)
        with open(warm, "wb") as fh:
            fh.write(synth.fastq_text(h, s, q))
        final, _, _ = predict_chop(warm, pipe, threads=args.threads, max_reads=args.max_reads, max_tokens=args.max_tokens)
        os.remove(final)
        os.remove(warm)
        torch.cuda.synchronize()
        if os.environ.get("DC_PROFILE_STAGES"):  # GPU time per stage next to the wall time: is the flow GPU-bound?
            pipe.model.net.set_profiling(True)
            pipe.model.net.profile_reset()
        t0 = time.time()
        final, n_reads, n_out = predict_chop(path, pipe, threads=args.threads, max_reads=args.max_reads, max_tokens=args.max_tokens)
        torch.cuda.synchronize()
        dt = time.time() - t0
        if os.environ.get("DC_PROFILE_STAGES"):
            out["stages_ms"] = {k: round(v[0], 1) for k, v in pipe.model.net.profile().items()}
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    else:
        cmd = [sys.executable, "-m", "deepchopper_amd.cli", "predict", path, "-g", str(args.gpus), "--chop", "-m", ckpt,
               "-t", str(args.threads), "--max-reads", str(args.max_reads), "--max-tokens", str(args.max_tokens), "-b", str(12 * args.gpus)]
        env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        t0 = time.time()
        proc = subprocess.run(cmd, capture_output=True, text=True, env=env)
        dt = time.time() - t0
        if proc.returncode != 0:
            raise SystemExit(proc.stderr[-3000:])
        final = proc.stdout.strip().splitlines()[-1]
        import re

        n_reads = int(re.search(r"\.(\d+)pd\.", os.path.basename(final)).group(1))
        rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
        out["note"] = "wall time of the whole command (interpreter + model load + first-batch warm-up included)"
    out.update({"seconds": round(dt, 2), "reads_per_s": round(n_reads / dt, 1), "Mbases_per_s": round(n_bases / dt / 1e6, 2),
                "peak_host_rss_MB": round(rss / 1024, 1), "output": os.path.basename(final),
                "output_MB": round(os.path.getsize(final) / 1e6, 1)})
    if not args.no_count:
        out.update(_count_output(final))
    print(json.dumps(out))
    os.remove(final)
    os.remove(path)


def feeder_bench(args):
    """How fast can ONE process parse (and inflate) a FASTQ and deal its chunks to N per-GPU workers?  The ceiling of
    `deepchopper predict --gpus N` on plain gzip input (the parent is the node's single inflater)."""
    import threading

    from deepchopper_amd import launcher, synth

    tmp = tempfile.mkdtemp(prefix="dc_feed_")
    path = os.path.join(tmp, "reads.fq")
    with _opener(args.container, path) as fh:
        for b0 in range(0, args.reads, 10_000):
            n = min(10_000, args.reads - b0)
            seq, qual, ids = synth.synth_reads_fixed(n, args.bp, seed=42 + b0, start_index=b0)
            fh.write(synth.to_fastq(seq, qual, ids))
    feeder = launcher.ChunkFeeder(path, args.gpus, args.max_reads, args.max_reads * (args.bp + 1))
    got = [0] * args.gpus

    def consume(i, fd):
        for c in launcher.PipeChunks(fd):
            got[i] += c.n

    fds = [os.dup(fd) for fd in feeder.read_fds]  # start() closes the parent's copies of the read ends
    ths = [threading.Thread(target=consume, args=(i, fd)) for i, fd in enumerate(fds)]
    t0 = time.time()
    feeder.start()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    feeder.join()
    dt = time.time() - t0
    print(json.dumps({"what": "single reader/inflater feeding N consumers over pipes (no GPU)", "container": args.container,
                      "consumers": args.gpus, "reads": sum(got), "bp": args.bp, "seconds": round(dt, 2),
                      "aggregate_reads_per_s": round(sum(got) / dt), "MB_per_s_uncompressed": round(sum(got) * (2 * args.bp + 20) / 1e6 / dt),
                      "host_cpus": os.cpu_count()}))
    os.remove(path)
    os.rmdir(tmp)


def ref_batch_bench(args):
    """`deepchopper predict -b 12` shape: tiny left-padded batches, where per-batch host waits dominate.  Same
    stream twice: pipelined (non-blocking pinned copies, results on a copy stream) and serialised (every batch
    waited for before the next launch, i.e. what blocking per-batch copies cost)."""
    import torch

    from deepchopper_amd import synth
    from deepchopper_amd.fastq import FastqReader
    from deepchopper_amd.model import DeepChopper
    from deepchopper_amd.predict import PredictPipeline

    tmp = tempfile.mkdtemp(prefix="dc_ref_")
    path = os.path.join(tmp, "reads.fq")
    with open(path, "wb") as fh:
        for b0 in range(0, args.reads, 10_000):
            n = min(10_000, args.reads - b0)
            seq, qual, ids = synth.synth_reads_fixed(n, args.bp, seed=42 + b0, start_index=b0)
            fh.write(synth.to_fastq(seq, qual, ids))
    torch.cuda.set_device(0)
    pipe = PredictPipeline(DeepChopper.synthetic(seed=42))
    out = {"mode": f"reference batches of {args.ref_batch} reads (left-padded, int64 ids, labels + intervals to the host)",
           "reads": args.reads, "bp": args.bp}
    from deepchopper_amd.fastq import fuse_chunks

    for name, ser in (("warmup", False), ("pipelined", False), ("serialised", True), ("fused", False)):
        t0 = time.time()
        n = 0
        with FastqReader(path) as rd:
            chunks = rd.chunks(args.ref_batch, 1 << 26)
            if name == "fused":  # what `deepchopper predict -b 12` does since round 4: reference batches share launches
                chunks = fuse_chunks(chunks, max_reads=8192, max_tokens=2_000_000, padded=not pipe.shared_prefix)
            for res in pipe.predict_stream(chunks, reference_padding=True, ids_i64=True, serialize=ser):
                n += len(res["names"])
                if name == "warmup" and n >= 50 * args.ref_batch:
                    break
        torch.cuda.synchronize()
        if name != "warmup":
            out[name + "_reads_per_s"] = round(n / (time.time() - t0), 1)
    print(json.dumps(out))
    os.remove(path)
    os.rmdir(tmp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--bp", type=int, default=1000)
    ap.add_argument("--container", default="plain", choices=["plain", "gzip", "bgzf"])
    ap.add_argument("--max-reads", type=int, default=2048)
    ap.add_argument("--max-tokens", type=int, default=2_000_000)
    ap.add_argument("--parse-only", action="store_true")
    ap.add_argument("--chop", action="store_true", help="config-5 flow: nanopore-like reads -> fused predict -> chop -> BGZF")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--cli", action="store_true", help="--chop through the command line even on one GPU")
    ap.add_argument("--threads", type=int, default=4, help="BGZF compression workers of the chop output")
    ap.add_argument("--no-count", action="store_true")
    ap.add_argument("--feeder-only", action="store_true", help="host side of a multi-GPU plain-gzip run: the parent's single "
                    "reader / inflater dealing chunks to --gpus consumers that discard them (no GPU work): aggregate reads/s")
    ap.add_argument("--ref-batch", type=int, default=0, help="reference-compatible mode: batches of this many reads "
                    "(the reference's -b 12), left-padded, int64 ids; reports reads/s pipelined and serialised")
    args = ap.parse_args()
    if args.ref_batch:
        return ref_batch_bench(args)
    if args.feeder_only:
        return feeder_bench(args)
    if args.chop:
        if args.max_reads == 2048:
            args.max_reads = 8192
        return chop_bench(args)
    from deepchopper_amd import synth
    from deepchopper_amd.fastq import FastqReader

    tmp = tempfile.mkdtemp(prefix="dc_e2e_")
    path = os.path.join(tmp, "reads.fq")
    t0 = time.time()
    block = 10_000
    with _opener(args.container, path) as fh:
        for b0 in range(0, args.reads, block):
            n = min(block, args.reads - b0)
            seq, qual, ids = synth.synth_reads_fixed(n, args.bp, seed=42 + b0, start_index=b0)
            fh.write(synth.to_fastq(seq, qual, ids))
    size = os.path.getsize(path)
    gen_s = time.time() - t0
    cap = args.max_reads * (args.bp + 1)
    t0 = time.time()
    n = 0
    with FastqReader(path) as rd:
        for c in rd.chunks(args.max_reads, cap):
            n += c.n
    parse_s = time.time() - t0
    out = {"container": args.container, "reads": n, "bp": args.bp, "file_MB": round(size / 1e6, 1), "gen_s": round(gen_s, 1),
           "parse_reads_per_s": round(n / parse_s), "parse_MB_per_s_uncompressed": round(n * (2 * args.bp + 20) / 1e6 / parse_s)}
    if not args.parse_only:
        import torch

        from deepchopper_amd.model import DeepChopper
        from deepchopper_amd.predict import PredictPipeline

        torch.cuda.set_device(0)
        pipe = PredictPipeline(DeepChopper.synthetic(seed=42))
        with FastqReader(path) as rd:  # warm-up: kernels, workspaces, filter planes
            for i, res in enumerate(pipe.predict_stream(rd.chunks(args.max_reads, cap))):
                if i >= 2:
                    break
        torch.cuda.synchronize()
        t0 = time.time()
        n2 = n_iv = 0
        with FastqReader(path) as rd:
            for res in pipe.predict_stream(rd.chunks(args.max_reads, cap)):
                n2 += len(res["names"])
                n_iv += int(res["n_intervals"].sum())
        torch.cuda.synchronize()
        e2e_s = time.time() - t0
        out.update({"e2e_reads_per_s": round(n2 / e2e_s), "e2e_s": round(e2e_s, 2), "intervals": n_iv, "batch_reads": args.max_reads})
    print(json.dumps(out))
    os.remove(path)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
