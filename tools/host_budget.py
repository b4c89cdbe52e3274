"""Host-side budget of an N-rank `deepchopper predict --chop` on a plain .fastq.gz: how many CPU
seconds per 1 000 reads does each worker process spend (and on what), and what does the parent -- the node's single
inflater -- cost?  From that: do 8 ranks x (32 k reads/s at 1 kb, ~24 k reads/s on the nanopore-like mix) fit the host?

The ranks share this box's one GPU (DC_SHARE_GPU=1), so the WALL time of a run says nothing about 8 GPUs; the CPU
seconds per read do.  The pool allows at most 6 processes on the card, so N = 8 itself cannot be rehearsed here: the
per-read costs are measured at N = 1, 2, 4 (they must not depend on N) and added up for 8.

    python tools/host_budget.py --reads 200000 [--gpus 1 2 4] [--threads 2 4]
"""
import argparse
import gzip
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200_000)
    ap.add_argument("--gpus", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--threads", type=int, nargs="+", default=[2, 4], help="BGZF compression workers per rank (-t)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    from deepchopper_amd import synth
    from deepchopper_amd.weights import save_state_dict, token_gated_state_dict

    tmp = tempfile.mkdtemp(prefix="dc_budget_")
    path = os.path.join(tmp, "hct116_like.fastq.gz")
    n_bases = 0
    with gzip.open(path, "wb", compresslevel=1) as fh:
        for b0 in range(0, args.reads, 50_000):
            n = min(50_000, args.reads - b0)
            h, s, q = synth.synth_nanopore_like(n, seed=42 + b0, start_index=b0)
            n_bases += sum(len(x) for x in s)
            fh.write(synth.fastq_text(h, s, q))
    ckpt = os.path.join(tmp, "ckpt")
    save_state_dict(token_gated_state_dict(0), os.path.join(ckpt, "model.safetensors"))
    runs = []
    for g in args.gpus:
        for t in (args.threads if g == args.gpus[0] or g == max(args.gpus) else args.threads[:1]):
            rep = os.path.join(tmp, f"report_g{g}_t{t}.json")
            cmd = [sys.executable, "-m", "deepchopper_amd.cli", "predict", path, "-g", str(g), "--chop", "-m", ckpt, "-t", str(t),
                   "-b", str(12 * g), "--report-json", rep, "--chop-output", os.path.join(tmp, f"out_g{g}_t{t}")]
            env = dict(os.environ, PYTHONPATH=ROOT, DC_SHARE_GPU="1")
            t0 = time.time()
            proc = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=tmp)
            wall = time.time() - t0
            if proc.returncode != 0:
                raise SystemExit(proc.stderr[-3000:])
            final = proc.stdout.strip().splitlines()[-1]
            with open(rep) as fh:
                r = json.load(fh)
            ranks = r["ranks"]
            per_k = lambda x, n: round(1000.0 * x / max(n, 1), 4)  # noqa: E731
            row = {"gpus": g, "bgzf_threads": t, "wall_s": round(wall, 2), "reads_per_s_whole_command": round(args.reads / wall, 1),
                   "output_MB": round(os.path.getsize(final) / 1e6, 1), "input": r.get("input", "one process: reader thread + GPU pipeline"),
                   "worker_cpu_s_per_1000_reads": {
                       "total": per_k(sum(x["cpu_user_s"] + x["cpu_sys_s"] for x in ranks), args.reads),
                       "bgzf_deflate": per_k(sum(x.get("deflate_cpu_s", 0.0) for x in ranks), args.reads),
                       "pipe_read": per_k(sum(x.get("pipe_read_cpu_s", 0.0) for x in ranks), args.reads),
                   },
                   "worker_max_rss_mb": max(x["max_rss_mb"] for x in ranks),
                   "per_rank_reads": [x["reads"] for x in ranks]}
            w = row["worker_cpu_s_per_1000_reads"]
            w["staging_launch_python_rest"] = round(w["total"] - w["bgzf_deflate"] - w["pipe_read"], 4)
            if "parent" in r:
                p = r["parent"]
                row["parent_cpu_s_per_1000_reads"] = per_k(p["cpu_user_s"] + p["cpu_sys_s"], args.reads)
                row["parent_max_rss_mb"] = p["max_rss_mb"]
            runs.append(row)
            os.remove(final)
            print(json.dumps(row), flush=True)
    out = {"what": "host CPU budget of `predict -g N --chop` on a nanopore-like plain .fastq.gz; ranks share ONE GPU, so only the CPU seconds per read carry over to 8 GPUs",
           "reads": args.reads, "bases": n_bases, "mean_len": round(n_bases / args.reads, 1), "file_MB": round(os.path.getsize(path) / 1e6, 1),
           "host_cpus": os.cpu_count(), "runs": runs}
    text = json.dumps(out, indent=1)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text + "\n")
    print(text)
    os.remove(path)


if __name__ == "__main__":
    main()
