"""`deepchopper predict reads.fq -b 12` end to end (the reference's default invocation: left-padded batches of 12, one 28 B/token
`.pt` file per batch): wall time of the whole command, reads/s, MB of `.pt` written -- on a nanopore-like length mix (what the
command meets in practice: the pad tokens of a batch of 12 are 2.8x its real tokens) and, labelled as such, on fixed-length
1 kb reads (no padding at all: the case in which the reference's batching costs nothing).

    python tools/bench_cli_ref.py [reads]           # DC_SHARED_PREFIX=0 / DC_PREFIX_MODE=exact in the environment: the other forms"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from deepchopper_amd import synth
from deepchopper_amd.weights import save_state_dict, synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
tmp = tempfile.mkdtemp(prefix="dc_cli_")
ck = os.path.join(tmp, "ckpt")
save_state_dict(synthetic_state_dict(0), os.path.join(ck, "model.safetensors"))


def run(fq: str, what: str, lens) -> dict:
    out = os.path.join(tmp, "predictions_" + what.split()[0])
    t0 = time.time()
    rep = os.path.join(tmp, "report.json")
    p = subprocess.run([sys.executable, "-m", "deepchopper_amd.cli", "predict", fq, "-o", out, "-b", "12", "-m", ck, "--report-json", rep],
                       cwd=ROOT, capture_output=True, text=True)
    dt = time.time() - t0
    assert p.returncode == 0, p.stderr[-2000:]
    with open(rep) as fh:
        rank0 = json.load(fh)["ranks"][0]
    phases = rank0.get("phases_s", {})
    busy = phases.get("stream", 0) + phases.get("drain", 0)
    files = os.listdir(os.path.join(out, "0"))
    mb = sum(os.path.getsize(os.path.join(out, "0", f)) for f in files) / 1e6
    padded = sum(int(lens[i: i + 12].max() + 1) * len(lens[i: i + 12]) for i in range(0, len(lens), 12))
    # the reference's second command over those files: `deepchopper chop predictions/0 reads.fq` (src/bin/predict.rs:197-384)
    chop_info = {}
    if os.environ.get("DC_BENCH_CHOP", "1") != "0":
        for form in (("native", "python") if os.environ.get("DC_BENCH_CHOP_PYTHON") == "1" else ("native",)):
            t1 = time.time()
            c = subprocess.run([sys.executable, "-m", "deepchopper_amd.cli", "chop", os.path.join(out, "0"), fq, "-o", os.path.join(tmp, "chopped_" + form)],
                               cwd=ROOT, capture_output=True, text=True, env={**os.environ, "DC_CHOP_PYTHON": "1" if form == "python" else "0"})
            dtc = time.time() - t1
            assert c.returncode == 0, c.stderr[-2000:]
            final = c.stdout.strip().splitlines()[-1]
            chop_info[form] = {"seconds_whole_command": round(dtc, 2), "reads_per_s": round(len(lens) / dtc, 1),
                               "pt_MB_per_s": round(mb / dtc, 1), "output": os.path.basename(final)}
            os.remove(final)
    subprocess.run(["rm", "-rf", out])
    return {"chop_command": chop_info or None, "two_commands_reads_per_s": round(len(lens) / (dt + chop_info["native"]["seconds_whole_command"]), 1) if chop_info else None,
            "command": "deepchopper predict reads.fq -b 12 (reference-compatible: .pt per batch of 12)", "input": what, "reads": len(lens),
            "mean_bp": round(float(lens.mean()), 1), "padded_over_real_tokens": round(padded / float(lens.sum() + len(lens)), 3),
            "seconds_whole_command": round(dt, 2), "reads_per_s": round(len(lens) / dt, 1), "Mbases_per_s": round(float(lens.sum()) / dt / 1e6, 2),
            "pt_files": len(files), "pt_MB": round(mb, 1), "pt_MB_per_s": round(mb / dt, 1),
            "phases_s": phases, "pt_writer_main_thread_s": rank0.get("pt_writer_main_thread_s"), "reads_per_s_after_startup": round(len(lens) / busy, 1) if busy else None,
            "pt_MB_per_s_after_startup": round(mb / busy, 1) if busy else None, "cpu_user_s": rank0.get("cpu_user_s"),
            "cpu_sys_s": rank0.get("cpu_sys_s"), "pt_writer": os.environ.get("DC_PT_WRITER", "native"),
            "shared_prefix": os.environ.get("DC_SHARED_PREFIX", "1") != "0", "prefix_mode": os.environ.get("DC_PREFIX_MODE", "linear")}


fq = os.path.join(tmp, "nanopore_like.fq")
heads, seqs, quals = synth.synth_nanopore_like(n, seed=77, gate=b"A")
with open(fq, "wb") as fh:
    fh.write(synth.fastq_text(heads, seqs, quals))
print(json.dumps(run(fq, "nanopore-like lengths (log-normal, median 900 bp, 100 bp - 30 kb)", np.array([len(s) for s in seqs]))), flush=True)
os.remove(fq)
fq = os.path.join(tmp, "fixed_1kb.fq")
synth.write_fastq_fixed_fast(fq, n, 1000)
print(json.dumps(run(fq, "fixed-length 1 kb reads (no batch has any padding)", np.full(n, 1000))), flush=True)
subprocess.run(["rm", "-rf", tmp])
