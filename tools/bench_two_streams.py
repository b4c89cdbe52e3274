"""Experiment: does running two half-batches concurrently on two HIP streams (two model handles = two workspaces)
beat one batch on one stream?  The step is a strict chain of kernels that are either matrix-pipe bound and power
limited (fused MLP, head linear2: ~75 % of the time, 1.6-1.7 GHz) or HBM bound (out_proj, long convolution); on one
stream they never overlap.  With two independent halves in flight the scheduler can run workgroups of both kinds
side by side on different CUs.

    python tools/bench_two_streams.py [--reads 2048] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deepchopper_amd import ops, synth
from deepchopper_amd.model import DeepChopper
from deepchopper_amd.predict import PredictPipeline
from deepchopper_amd.weights import synthetic_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2048, help="reads per step (split over the lanes)")
    ap.add_argument("--bp", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--lanes", type=int, nargs="+", default=[1, 2, 3])
    args = ap.parse_args()
    torch.cuda.set_device(0)
    sd = synthetic_state_dict(0)
    out = {"reads_per_step": args.reads, "bp": args.bp, "steps": args.steps}
    for lanes in args.lanes:
        B = args.reads // lanes
        pipes = [PredictPipeline(DeepChopper.from_state_dict(sd)) for _ in range(lanes)]
        streams = [torch.cuda.Stream() for _ in range(lanes)]
        batches = []
        for j in range(lanes):
            seq, qual, ids = synth.synth_reads_fixed(B, args.bp, seed=42, start_index=j * B)
            offs = np.arange(B + 1, dtype=np.int64) * args.bp
            batches.append((ops.to_device(seq.reshape(-1)), ops.to_device(qual.reshape(-1)), ops.to_device(offs)))
        lens = np.full(B, args.bp, np.int64)
        torch.cuda.synchronize()

        def step():
            for j in range(lanes):
                with torch.cuda.stream(streams[j]):
                    s, q, o = batches[j]
                    pipes[j].run_device_batch(s, q, o, lens, want_labels=True)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"lanes_{lanes}"] = {"reads_per_lane": B, "ms_per_step": round(dt / args.steps * 1e3, 3),
                                 "reads_per_s": round(B * lanes * args.steps / dt, 1)}
        del pipes
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
