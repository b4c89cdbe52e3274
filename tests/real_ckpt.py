"""Where the skipped-unless-present F2 pin tests look for a user-supplied REAL DeepChopper checkpoint and for the
reference's own output with that checkpoint.

What can pin the HyenaDNA backbone (F2) and what cannot
-------------------------------------------------------
* The weights (`yangliz5/deepchopper`, `yangliz5/deepchopper-rna004`) live only on the HF Hub (deepchopper/cli.py:117-121)
  and cannot be fetched in this pipeline: $DEEPCHOPPER_MODEL_DIR = a folder that holds the `model.safetensors`
  PyTorchModelHubMixin wrote (or a Lightning `.ckpt`), or a folder with the Hub layout below it.
* The reference's committed `.pt` fixtures (/root/reference/tests/data/eval/chunk{0,1}/{0,1,2}.pt = tests/golden/eval_chunk*.npz)
  were NOT produced by HyenaDNA: their logits have a finite symmetric receptive field of +-12 tokens -- three "same"-padded
  convolutions of widths 7, 9, 11 = `BenchmarkCNN` (deepchopper/models/cnn.py:24-37, configs/model/cnn.yaml) -- see
  tests/test_reference_fixture_model.py.  They pin T3-T5 (batch layout) and S0-S5 (argmax -> intervals) and nothing about F2;
  no HyenaDNA logits exist anywhere in the reference tree.
* What does pin F2: a directory of `{rank}_{batch}.pt` files written by the reference's OWN `deepchopper predict` with the real
  model (`CustomWriter`, deepchopper/models/callbacks.py:12-25): $DEEPCHOPPER_REF_PT_DIR (+ $DEEPCHOPPER_MODEL_DIR for the
  weights that produced it).  A `.pt` whose pad-prefix logits show look-ahead cannot come from a causal model and is refused."""
import glob
import os

import numpy as np

TOL_LOGIT = 1e-3  # BASELINE.json north_star: per-base logits within 1e-3
PAD_ID = 4        # [PAD] of the HyenaDNA character tokenizer (src/smooth/utils.rs:6-25; fixtures)
LOOKAHEAD_TOL = 1e-2  # rounding noise of an fp32 FFT convolution on O(50) logits is ~1e-4; BenchmarkCNN's look-ahead is ~80


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


def find_reference_pt_files():
    """`{out}/{dataloader_idx}/{rank}_{batch}.pt` (callbacks.py:25): the directory itself or its `0/` sub-folder."""
    root = os.environ.get("DEEPCHOPPER_REF_PT_DIR", "")
    if not root or not os.path.isdir(root):
        return []
    for d in (root, os.path.join(root, "0")):
        files = sorted(glob.glob(os.path.join(d, "*.pt")))
        if files:
            return files
    return []


SKIP_NO_CKPT = ("no real checkpoint: set DEEPCHOPPER_MODEL_DIR to a folder holding yangliz5/deepchopper's model.safetensors "
                "(the HF Hub is unreachable from this pipeline)")
SKIP_NO_REF_OUTPUT = ("F2 (the HyenaDNA backbone) can only be pinned by the reference's own output with the real model: set "
                      "DEEPCHOPPER_REF_PT_DIR to a directory of {rank}_{batch}.pt files written by the reference's `deepchopper "
                      "predict` (callbacks.py:12-25) and DEEPCHOPPER_MODEL_DIR to the checkpoint that produced them.  The "
                      "reference's committed tests/data/eval/*.pt are BenchmarkCNN outputs and cannot serve")


def fixture_names():
    return [f"eval_chunk{c}_{i}.npz" for c in (0, 1) for i in (0, 1, 2)]


def pad_counts(seq: np.ndarray) -> np.ndarray:
    """Length of the leading [PAD] run of every row of a left-padded `seq[B, L]`."""
    is_pad = np.asarray(seq) == PAD_ID
    return np.where(is_pad.all(axis=1), is_pad.shape[1], np.argmin(is_pad, axis=1)).astype(np.int64)


def max_lookahead(prediction: np.ndarray, seq: np.ndarray):
    """Evidence of look-ahead in a left-padded batch: over all row pairs, the largest |dlogit| at positions where BOTH rows
    have seen nothing but [PAD] so far.  A causal model (HyenaDNA: `short_filter(...)[..., :L]`, 2L-point `fftconv`; restated
    in oracle/forward_ref.py:68-77,106-113) gives rounding noise there; a model with a symmetric receptive field differs in
    the last positions of the shorter pad run.  Returns (max diff, pairs compared); (0.0, 0) when no two rows share a pad
    prefix (nothing can be said)."""
    npad = pad_counts(seq)
    rows = [b for b in range(len(npad)) if npad[b] > 0]
    worst, pairs = 0.0, 0
    for i, a in enumerate(rows):
        for b in rows[i + 1:]:
            m = int(min(npad[a], npad[b]))
            if npad[a] == npad[b]:
                continue  # identical pad runs: the next token differs, look-ahead is not visible in the prefix
            worst = max(worst, float(np.abs(prediction[a, :m] - prediction[b, :m]).max()))
            pairs += 1
    return worst, pairs


def load_reference_pt(path: str) -> dict:
    """One `CustomWriter` file -> numpy dict with the five keys (callbacks.py:17-25)."""
    import torch

    d = torch.load(path, map_location="cpu", weights_only=True)
    missing = {"prediction", "target", "seq", "qual", "id"} - set(d)
    if missing:
        raise ValueError(f"{path}: not a CustomWriter file, keys {sorted(missing)} are missing")
    return {k: v.numpy() for k, v in d.items()}


def require_causal(batch: dict, tag: str):
    worst, pairs = max_lookahead(batch["prediction"], batch["seq"])
    if pairs and worst > LOOKAHEAD_TOL:
        raise AssertionError(
            f"{tag}: the stored logits over a shared [PAD] prefix differ between rows by {worst:.3g} -- the model that wrote "
            f"this file looks ahead, so it is not the causal HyenaDNA model `deepchopper predict` ships (dc_hg.py:120-163); "
            f"it cannot pin F2 (the reference's own tests/data/eval/*.pt are such files: BenchmarkCNN)")
    return pairs
