"""The F2 pin harness (CPU oracle side), skipped unless BOTH are given: $DEEPCHOPPER_MODEL_DIR (the real weights) and
$DEEPCHOPPER_REF_PT_DIR (a directory of `{rank}_{batch}.pt` files written by the reference's own `deepchopper predict` with
those weights, deepchopper/models/callbacks.py:12-25).  The torch-fp32 restatement of the forward (oracle/forward_ref.py)
must reproduce the stored `prediction` within 1e-3 -- that, and only that, turns "parity unpinned" for F2 into pinned.

The reference's committed tests/data/eval/*.pt cannot serve: they are `BenchmarkCNN` outputs
(tests/test_reference_fixture_model.py); a file that shows look-ahead is refused here with that explanation."""
import os

import numpy as np
import pytest

import real_ckpt


@pytest.mark.skipif(not real_ckpt.find_reference_pt_files() or real_ckpt.find_real_checkpoint() is None,
                    reason=real_ckpt.SKIP_NO_REF_OUTPUT)
def test_oracle_with_real_checkpoint_reproduces_reference_output():
    from deepchopper_amd.weights import load_checkpoint
    from oracle import forward_ref as fr

    sd, cfg = load_checkpoint(real_ckpt.find_real_checkpoint())
    ocfg = fr.HyenaDNAConfig(**{k: v for k, v in cfg.to_dict().items() if k in fr.HyenaDNAConfig.__dataclass_fields__})
    budget = 40_000  # tokens: a few minutes of torch-fp32 on the CPU
    for path in real_ckpt.find_reference_pt_files():
        d = real_ckpt.load_reference_pt(path)
        real_ckpt.require_causal(d, os.path.basename(path))
        rows = max(1, min(d["seq"].shape[0], budget // max(1, d["seq"].shape[1])))
        got = fr.forward(sd, d["seq"].astype(np.int64)[:rows], d["qual"].astype(np.float32)[:rows], ocfg).numpy()
        err = float(np.abs(got - d["prediction"][:rows]).max())
        assert err <= real_ckpt.TOL_LOGIT, f"{os.path.basename(path)}: oracle max |dlogit| = {err:.3e}"
        budget -= rows * d["seq"].shape[1]
        if budget <= 0:
            break
