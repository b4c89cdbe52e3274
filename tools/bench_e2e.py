"""File-to-intervals throughput of the streaming predict path (FASTQ on disk -> native reader -> H2D ->
tokenise/forward/smooth -> D2H labels + intervals), next to the parse-only rate of the host reader.
This is the PCIe- and parse-inclusive figure DESIGN.md quotes beside bench.py's HBM-resident `value`.

    python tools/bench_e2e.py --reads 100000 --bp 1000 --container plain|gzip|bgzf
"""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--bp", type=int, default=1000)
    ap.add_argument("--container", default="plain", choices=["plain", "gzip", "bgzf"])
    ap.add_argument("--max-reads", type=int, default=2048)
    ap.add_argument("--parse-only", action="store_true")
    args = ap.parse_args()
    from deepchopper_amd import synth
    from deepchopper_amd.fastq import BgzfWriter, FastqReader

    tmp = tempfile.mkdtemp(prefix="dc_e2e_")
    path = os.path.join(tmp, "reads.fq")
    t0 = time.time()
    block = 10_000
    opener = {"plain": lambda: open(path, "wb"), "gzip": lambda: gzip.open(path, "wb", compresslevel=1),
              "bgzf": lambda: BgzfWriter(path, threads=8, level=1)}[args.container]
    with opener() as fh:
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
