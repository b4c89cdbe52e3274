"""The two reference-batching legs of bench.py on their own (`configs.reference_b12`, `configs.reference_b12_nanopore`): FASTQ file ->
the reference's left-padded batches of 12 -> logits + labels + intervals on the host.  For rocprofv3 runs of that flow
(`rocprofv3 --kernel-trace --stats -- python3 tools/bench_ref_legs.py`) and A/Bs of the prefix forms:

    DC_SHARED_PREFIX=0            every row's pads through the model (rounds 1-4)
    DC_PREFIX_MODE=exact          shared [PAD] rows, long convolution over the padded rows (bit-identical on the Toeplitz forms)
    (default)                     linear form

    python tools/bench_ref_legs.py [reads] [uniform|nanopore|both]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepchopper_amd.model import DeepChopper
from deepchopper_amd.predict import PredictPipeline
from deepchopper_amd.weights import synthetic_state_dict
from tools import bench_legs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
which = sys.argv[2] if len(sys.argv) > 2 else "both"
torch.cuda.set_device(0)
sd = synthetic_state_dict(42)
pipe = PredictPipeline(DeepChopper.from_state_dict(sd))
out = {"shared_prefix": pipe.shared_prefix, "prefix_mode": getattr(pipe.model.net, "prefix_mode", None)}
if which in ("uniform", "both"):
    out["reference_b12"] = bench_legs.reference_b12_leg(pipe, sd, n)
if which in ("nanopore", "both"):
    out["reference_b12_nanopore"] = bench_legs.reference_b12_leg(pipe, sd, n, mix="nanopore")
if which in ("pt", "both"):
    out["reference_b12_nanopore_pt_files"] = bench_legs.reference_b12_leg(pipe, sd, n, mix="nanopore", write_pt=True)
print(json.dumps(out))
