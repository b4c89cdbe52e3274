"""Which threads of a predict_stream process burn CPU while the GPU is the limiter?  (Round 6: two threads per rank ran at
100 % whatever the throughput.)  python tools/spin_probe.py [reads]  -- env as usual (e.g. GPU_MAX_HW_QUEUES, HIP flags)."""
import json
import os
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepchopper_amd import synth
from deepchopper_amd.fastq import FastqReader
from deepchopper_amd.model import DeepChopper
from deepchopper_amd.predict import PredictPipeline
from tools.bench_multi import _thread_cpu, _top_threads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
tmp = tempfile.mkdtemp(prefix="dc_spin_")
path = os.path.join(tmp, "r.fq")
synth.write_fastq_fixed_fast(path, n, 2000, seed=1)
torch.cuda.set_device(0)
pipe = PredictPipeline(DeepChopper.synthetic(seed=42))
with FastqReader(path) as rd:
    for i, _ in enumerate(pipe.predict_stream(rd.chunks(1024, 1024 * 2001))):
        if i >= 2:
            break
torch.cuda.synchronize()
for k in pipe.stream_timers:
    pipe.stream_timers[k] = 0.0
th0, c0, t0 = _thread_cpu(), time.process_time(), time.perf_counter()
nr = 0
with FastqReader(path) as rd:
    for res in pipe.predict_stream(rd.chunks(1024, 1024 * 2001)):
        nr += len(res["names"])
torch.cuda.synchronize()
dt, cpu = time.perf_counter() - t0, time.process_time() - c0
print(json.dumps({"reads_per_s": round(nr / dt, 1), "wall_s": round(dt, 2), "cpu_s": round(cpu, 2), "cores": round(cpu / dt, 2),
                  "cpu_s_per_1000_reads": round(1000 * cpu / nr, 4), "top_threads": _top_threads(th0, _thread_cpu(), 8),
                  "main_thread": {k: round(v, 2) for k, v in pipe.stream_timers.items()},
                  "env": {k: os.environ.get(k) for k in ("AMD_DIRECT_DISPATCH", "ROC_ACTIVE_WAIT_TIMEOUT", "HSA_ENABLE_INTERRUPT", "ROC_CPU_WAIT_FOR_SIGNAL", "DC_SPIN_NOTE")}}))
os.remove(path)
os.rmdir(tmp)
