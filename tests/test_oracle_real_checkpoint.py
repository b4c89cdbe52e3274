"""The F2 pin harness (CPU oracle side), skipped unless $DEEPCHOPPER_MODEL_DIR holds the real weights: the torch-fp32
restatement of the forward (oracle/forward_ref.py) replays the reference's own stored-logit reads
(tests/golden/eval_chunk*.npz = /root/reference/tests/data/eval/chunk*/?.pt) and must reproduce the stored
`prediction` within 1e-3 -- the moment a checkpoint is available, this turns "parity unpinned" into pinned."""
import os

import numpy as np
import pytest

import real_ckpt


@pytest.mark.skipif(real_ckpt.find_real_checkpoint() is None, reason=real_ckpt.SKIP_REASON)
@pytest.mark.parametrize("name", real_ckpt.fixture_names())
def test_oracle_with_real_checkpoint_reproduces_reference_logits(golden_dir, name):
    from deepchopper_amd.weights import load_checkpoint
    from oracle import forward_ref as fr

    sd, cfg = load_checkpoint(real_ckpt.find_real_checkpoint())
    d = np.load(os.path.join(golden_dir, name))
    want = d["prediction"]
    ocfg = fr.HyenaDNAConfig(**{k: v for k, v in cfg.to_dict().items() if k in fr.HyenaDNAConfig.__dataclass_fields__})
    got = fr.forward(sd, d["seq"].astype(np.int64)[:4], d["qual"].astype(np.float32)[:4], ocfg).numpy()  # 4 rows: minutes on CPU
    err = float(np.abs(got - want[:4]).max())
    assert err <= real_ckpt.TOL_LOGIT, f"{name}: oracle max |dlogit| = {err:.3e}"
