"""Convert the reference's committed prediction fixtures (data, not code) into npz.

Run once in the build container (reads /root/reference/tests/data, which does not exist on the
GPU box):   python tests/golden/make_pt_fixtures.py

Inputs : tests/data/eval/chunk{0,1}/{0,1,2}.pt  -- real `CustomWriter` outputs
         (deepchopper/models/callbacks.py:17-25): prediction f32[12,L,2], target, seq, qual, id.
         tests/data/input_ids.pt / input_quals.pt -- model-input samples [12,3223].
Outputs: tests/golden/eval_chunk{c}_{i}.npz, tests/golden/model_inputs.npz (lossless).
"""
import os
import numpy as np
import torch

REF = "/root/reference/tests/data"
OUT = os.path.dirname(os.path.abspath(__file__))

for c in (0, 1):
    for i in (0, 1, 2):
        d = torch.load(f"{REF}/eval/chunk{c}/{i}.pt", weights_only=True)
        np.savez_compressed(
            f"{OUT}/eval_chunk{c}_{i}.npz",
            prediction=d["prediction"].numpy(),
            target=d["target"].numpy().astype(np.int8),
            seq=d["seq"].numpy().astype(np.uint8),
            qual=d["qual"].numpy(),
            id=d["id"].numpy().astype(np.int16),
        )
ids = torch.load(f"{REF}/input_ids.pt", weights_only=True).numpy().astype(np.uint8)
quals = torch.load(f"{REF}/input_quals.pt", weights_only=True).numpy()
np.savez_compressed(f"{OUT}/model_inputs.npz", input_ids=ids, input_quals=quals)
print("ok")
