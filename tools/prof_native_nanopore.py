"""rocprofv3 target: config 5's flow on 60 000 nanopore-like reads with every read at its own length (native mode) -- the kernel
stats to hold against tools/bench_ref_legs.py nanopore (the same mix in reference batches: hyena4<...,2> against <...,0>)."""
import json, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tools import bench_legs
r = bench_legs.gz_chop_leg(60000, threads=4)
print(json.dumps({k: r[k] for k in ("value", "reads", "seconds")}))
