"""Forward time of very long reads with the FFT long convolution on and off: reads of 16 385 ..
32 769 tokens took the O(L^2) block-Toeplitz MFMA kernel until round 3; they now take the four-step 2^15-point FFT
(16 x 2048, csrc/hyena_fft.hip).  Prints one JSON line per (reads, tokens) case with both times and the stage split.

    python tools/bench_long_reads.py [--cases 1x32769 4x32769 8x20000 4x16384]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deepchopper_amd import ops
from deepchopper_amd.weights import HyenaDNAConfig, synthetic_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", nargs="+", default=["1x32769", "4x32769", "8x20000", "4x16384"])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    model = ops.HipModel(synthetic_state_dict(0), HyenaDNAConfig())
    rng = np.random.default_rng(0)
    for case in args.cases:
        B, L = (int(x) for x in case.split("x"))
        ids = rng.integers(7, 12, (B, L)).astype(np.uint8)
        ids[:, -1] = 1
        q = rng.random((B, L)).astype(np.float32)
        d_ids, d_q = ops.to_device(ids), ops.to_device(q)
        row = {"reads": B, "tokens": L, "gemm_mode": model.gemm_mode}
        outs = {}
        for name, thr in (("toeplitz_mfma", 1 << 20), ("fft", 2560)):
            model.set_fft_threshold(thr)
            outs[name] = model.forward(d_ids, d_q)  # warm-up: filters, spectra, workspace
            torch.cuda.synchronize()
            model.set_profiling(True)
            model.profile_reset()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                model.forward(d_ids, d_q)
            b.record()
            torch.cuda.synchronize()
            st = model.profile()
            model.set_profiling(False)
            row[name] = {"forward_ms": round(a.elapsed_time(b) / args.reps, 3),
                         "long_conv_ms_per_layer": {k: round(v[0] / max(v[1], 1), 3) for k, v in st.items() if k.startswith("hyena")}}
        row["max_abs_dlogit_fft_vs_toeplitz"] = float((outs["fft"] - outs["toeplitz_mfma"]).abs().max())
        row["speedup_forward"] = round(row["toeplitz_mfma"]["forward_ms"] / row["fft"]["forward_ms"], 2)
        print(json.dumps(row), flush=True)
    model.set_fft_threshold(2560)


if __name__ == "__main__":
    main()
