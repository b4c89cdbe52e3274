"""Does the ROCm runtime itself keep a thread busy while a stream has work?  Pure torch: a loop of matmuls on the GPU, the host
sleeping between event queries.  Variants: copies to pinned host memory on a second stream (as the pipeline does) or not."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.bench_multi import _thread_cpu, _top_threads

mode = sys.argv[1] if len(sys.argv) > 1 else "compute"
torch.cuda.set_device(0)
a = torch.randn(8192, 8192, device="cuda")
out = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
src = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
copy_stream = torch.cuda.Stream()
torch.cuda.synchronize()
th0, t0 = _thread_cpu(), time.perf_counter()
for it in range(40):
    b = a @ a
    b = b @ a
    ev = torch.cuda.Event()
    ev.record()
    if mode in ("copy", "copy_h2d", "xstream_kernel"):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ev)
            if mode == "copy":
                out.copy_(src, non_blocking=True)
            elif mode == "copy_h2d":
                src.copy_(out, non_blocking=True)
            else:
                src.add_(1)  # a kernel on the second stream behind the first stream's event: no copy engine involved
            ev = torch.cuda.Event()
            ev.record(copy_stream)
    elif mode == "same_stream_copy":
        out.copy_(src, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    elif mode == "record_only":
        pass
    while not ev.query():
        time.sleep(0.0005)
dt = time.perf_counter() - t0
print(json.dumps({"mode": mode, "wall_s": round(dt, 2), "top_threads": _top_threads(th0, _thread_cpu(), 6)}))
