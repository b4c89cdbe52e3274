"""`deepchopper predict reads.fq -b 12` end to end (the reference's default invocation: left-padded batches of 12, one 28 B/token
`.pt` file per batch): wall time of the whole command, reads/s, MB of `.pt` written.   python tools/bench_cli_ref.py [reads]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepchopper_amd import synth
from deepchopper_amd.weights import save_state_dict, synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
tmp = tempfile.mkdtemp(prefix="dc_cli_")
fq = os.path.join(tmp, "reads.fq")
synth.write_fastq_fixed_fast(fq, n, 1000)
ck = os.path.join(tmp, "ckpt")
save_state_dict(synthetic_state_dict(0), os.path.join(ck, "model.safetensors"))
out = os.path.join(tmp, "predictions")
t0 = time.time()
p = subprocess.run([sys.executable, "-m", "deepchopper_amd.cli", "predict", fq, "-o", out, "-b", "12", "-m", ck], cwd=ROOT,
                   capture_output=True, text=True)
dt = time.time() - t0
assert p.returncode == 0, p.stderr[-2000:]
files = os.listdir(os.path.join(out, "0"))
mb = sum(os.path.getsize(os.path.join(out, "0", f)) for f in files) / 1e6
print(json.dumps({"command": "deepchopper predict reads.fq -b 12 (reference-compatible: .pt per batch of 12)", "reads": n, "bp": 1000,
                  "seconds_whole_command": round(dt, 2), "reads_per_s": round(n / dt, 1), "pt_files": len(files), "pt_MB": round(mb, 1),
                  "pt_MB_per_s": round(mb / dt, 1)}))
subprocess.run(["rm", "-rf", tmp])
