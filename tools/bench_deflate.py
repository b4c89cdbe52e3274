"""CPU cost of the BGZF writer's deflate on nanopore-like FASTQ text: this library's encoder (csrc/fast_deflate.h, what dc_bgzf_*
uses at the default level) against zlib level 6, in BGZF-sized blocks, one thread.  python tools/bench_deflate.py [reads] [fastq ...]
(with FASTQ files: those instead of the synthetic reads)."""
import ctypes as C
import json
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepchopper_amd import _iolib, synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
if len(sys.argv) > 2:
    text = b"".join(open(f, "rb").read() for f in sys.argv[2:])
    n_reads = text.count(b"\n") // 4
    what = "files: " + ", ".join(os.path.basename(f) for f in sys.argv[2:])
else:
    heads, seqs, quals = synth.synth_nanopore_like(n_reads, seed=7, gate=b"A")
    text = synth.fastq_text(heads, seqs, quals)
    what = "synthetic nanopore-like reads (log-normal lengths, random bases / qualities)"
lib = _iolib.lib()
out = C.create_string_buffer(70000)
n = C.c_int64(0)
blocks = [text[o: o + 65280] for o in range(0, len(text), 65280)]
bufs = [(C.c_ubyte * len(b)).from_buffer_copy(b) for b in blocks]
res = {"input": what, "reads": n_reads, "MB": round(len(text) / 1e6, 2)}
best = 1e9
for _ in range(5):
    t = time.perf_counter()
    tot = 0
    for b, buf in zip(blocks, bufs):
        lib.dc_deflate_block(buf, len(b), out, 70000, C.byref(n))
        tot += n.value
    best = min(best, time.perf_counter() - t)
res["own_encoder"] = {"ratio": round(tot / len(text), 4), "MB_per_s": round(len(text) / best / 1e6, 1), "cpu_s_per_1000_reads": round(best / n_reads * 1000, 4)}
best = 1e9
for _ in range(3):
    t = time.perf_counter()
    tot = 0
    for b in blocks:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        tot += len(c.compress(b) + c.flush())
    best = min(best, time.perf_counter() - t)
res["zlib_level_6"] = {"ratio": round(tot / len(text), 4), "MB_per_s": round(len(text) / best / 1e6, 1), "cpu_s_per_1000_reads": round(best / n_reads * 1000, 4)}
res["speedup"] = round(res["own_encoder"]["MB_per_s"] / res["zlib_level_6"]["MB_per_s"], 2)
print(json.dumps(res))
