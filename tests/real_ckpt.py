"""Where a user-supplied REAL DeepChopper checkpoint is looked for by the skipped-unless-present pin tests.

The weights (`yangliz5/deepchopper`, `yangliz5/deepchopper-rna004`) live only on the HF Hub (deepchopper/cli.py:117-121)
and cannot be fetched in this pipeline.  Point $DEEPCHOPPER_MODEL_DIR at a folder that holds the `model.safetensors`
PyTorchModelHubMixin wrote (or a Lightning `.ckpt`), or at a folder with the Hub layout below it."""
import glob
import os

TOL_LOGIT = 1e-3  # BASELINE.json north_star: per-base logits within 1e-3


def find_real_checkpoint():
    root = os.environ.get("DEEPCHOPPER_MODEL_DIR", "")
    if not root or not os.path.isdir(root):
        return None
    for sub in ("", "yangliz5/deepchopper", "deepchopper", "rna002"):
        d = os.path.join(root, sub) if sub else root
        for name in ("model.safetensors", "pytorch_model.bin"):
            if os.path.exists(os.path.join(d, name)):
                return d
        ck = sorted(glob.glob(os.path.join(d, "*.ckpt")))
        if ck:
            return ck[0]
    return None


SKIP_REASON = ("no real checkpoint: set DEEPCHOPPER_MODEL_DIR to a folder holding yangliz5/deepchopper's model.safetensors "
               "(the HF Hub is unreachable from this pipeline); this is the test that pins the HyenaDNA backbone (F2)")


def fixture_names():
    return [f"eval_chunk{c}_{i}.npz" for c in (0, 1) for i in (0, 1, 2)]
