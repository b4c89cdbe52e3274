"""Integer / byte kernels of the path at BASELINE scale (64 k reads x 1 kb per launch): achieved HBM GB/s of
`tokenize_batch_kernel` (T1-T5), `kmer_encode_kernel` (K2-K3) and `smooth_kernel` (S0-S5) against their
algorithmic bytes (7 / 10 (+4 for the full-phred row) / 9 B per base).  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations committed under profiles/; the numbers
printed here come from HIP events on the launch stream.

    python tools/bench_intkernels.py [reads] [bp]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deepchopper_amd import ops, synth

n, bp = (int(sys.argv[1]) if len(sys.argv) > 1 else 65536), (int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
torch.cuda.set_device(0)
seq, qual, ids = synth.synth_reads_fixed(n, bp, seed=42, plant=False)
seq[seq == ord("N")] = ord("A")  # the k-mer table of the benchmark is ACGT + N; keep every read valid
d_seq, d_qual = ops.to_device(seq.reshape(-1)), ops.to_device(qual.reshape(-1))
offs = ops.to_device(np.arange(n + 1, dtype=np.int64) * bp)
rng = np.random.default_rng(43)
logits = torch.from_numpy(rng.standard_normal((n, bp + 1, 2)).astype(np.float32)).cuda()
first = torch.zeros(n, dtype=torch.int32, device="cuda")
length = torch.full((n,), bp, dtype=torch.int32, device="cuda")
out_off = ops.to_device(np.arange(n + 1, dtype=np.int64) * bp)
rows = torch.arange(n, dtype=torch.int32, device="cuda")
k, bases = 3, b"ACGTN"


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


res = {}
ms = timed(lambda: ops.tokenize_batch(d_seq, d_qual, offs, max_length=32769, row_stride=bp + 1, ids_i64=False, want_labels=True))
res["tokenize_batch_kernel"] = {"ms": round(ms, 4), "algorithmic_bytes": n * bp * 8, "note": "2 B in + 1 B token + 4 B quality + 1 B label per base"}
ms = timed(lambda: ops.kmer_encode(d_seq, d_qual, offs, rows, n, k, bases, 33, bp - k + 1, bp))
res["kmer_encode_kernel"] = {"ms": round(ms, 4), "algorithmic_bytes": n * bp * 14, "note": "2 B in + 2 x 4 B k-mer rows + 4 B full-phred row per base"}
ms = timed(lambda: ops.smooth_intervals_from_logits(logits, first, length, want_labels=True, out_offsets=out_off, total=n * bp))
res["smooth_kernel<true>"] = {"ms": round(ms, 4), "algorithmic_bytes": n * bp * 10,
                              "note": "8 B logits in + 1 B label + 1 B smoothed label out per base; logits are white noise: half the labels "
                                      "are ones and a read has dozens of runs (the worst case for the run / interval phases)"}
# SURVEY 8d's label stream (Bernoulli(0.02) noise + planted runs of geometric length, mean 60; seed 43) as logits with a margin
lab = synth.synth_label_stream(n * (bp + 1), seed=43).reshape(n, bp + 1)
logits[..., 1] = torch.from_numpy(np.where(lab == 1, 1.0, -1.0).astype(np.float32)).cuda() + 0.25 * logits[..., 1].clamp(-2, 2)
logits[..., 0] = 0.25 * logits[..., 0].clamp(-2, 2)
del lab
ms = timed(lambda: ops.smooth_intervals_from_logits(logits, first, length, want_labels=True, out_offsets=out_off, total=n * bp))
res["smooth_kernel<true>, SURVEY 8d label stream"] = {"ms": round(ms, 4), "algorithmic_bytes": n * bp * 10,
                                                      "note": "same traffic; labels = Bernoulli(0.02) noise + planted runs (mean 60), seed 43"}
for v in res.values():
    v["GB_per_s"] = round(v["algorithmic_bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
    v["frac_of_8TBps"] = round(v["GB_per_s"] / 8000, 4)
print(json.dumps({"reads": n, "bp": bp, "kernels": res}))
