"""Config-4 style throughput: mixed-length reads (log-uniform 100 bp - 10 kb) through predict_stream from a FASTQ
file, token-budget batches.  python tools/bench_mixed.py [reads]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepchopper_amd import synth
from deepchopper_amd.fastq import FastqReader
from deepchopper_amd.model import DeepChopper
from deepchopper_amd.predict import PredictPipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seqs, quals, ids = synth.synth_reads_mixed(n, 100, 10000, seed=42)
tmp = tempfile.mkdtemp(prefix="dc_mixed_")
path = os.path.join(tmp, "mixed.fq")
with open(path, "wb") as fh:
    fh.write(synth.to_fastq(seqs, quals, ids))
bases = sum(len(s) for s in seqs)
torch.cuda.set_device(0)
pipe = PredictPipeline(DeepChopper.synthetic(seed=42))
if os.environ.get("DC_PROFILE_STAGES"):
    pipe.model.net.set_profiling(True)
for rep in range(2):
    pipe.model.net.profile_reset()
    torch.cuda.synchronize()
    t0 = time.time()
    nr = 0
    with FastqReader(path) as rd:
        for res in pipe.predict_stream(rd.chunks(8192, 2_000_000)):
            nr += len(res["names"])
    torch.cuda.synchronize()
    dt = time.time() - t0
print(json.dumps({"workload": "config 4: log-uniform 100 bp - 10 kb reads, FASTQ file -> intervals", "reads": nr,
                  "mean_bp": round(bases / n, 1), "reads_per_s": round(nr / dt), "Mbases_per_s": round(bases / dt / 1e6, 2),
                  "fft_min_tokens": os.environ.get("DC_FFT_MIN", "2560 (default)"),
                  "stages_ms": {k: round(v[0], 1) for k, v in pipe.model.net.profile().items()} if os.environ.get("DC_PROFILE_STAGES") else None}))
os.remove(path)
os.rmdir(tmp)
