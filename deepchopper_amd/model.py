"""`DeepChopper.from_pretrained()` surface (deepchopper/models/dc_hg.py:26-163) over the HIP model.

The returned object answers the two calls `deepchopper predict` makes on the reference's
`TokenClassificationLit` (deepchopper/models/basic_module.py:90-100,197-207):
`forward(input_ids, input_quals) -> logits[B, L, 2]` and `predict_step(batch) -> (logits, labels)`.
"""
from __future__ import annotations

import os

import torch

from . import _lib, ops
from .weights import HyenaDNAConfig, load_checkpoint, synthetic_state_dict

# names `deepchopper predict --model` resolves on the HF Hub (cli.py:117-121)
HUB_NAMES = {"rna002": "yangliz5/deepchopper", "rna004": "yangliz5/deepchopper-rna004"}


class TokenClassificationModel:
    """Inference-only stand-in for `TokenClassificationLit`; all arithmetic is in libdeepchopper_hip."""

    def __init__(self, state_dict, cfg: HyenaDNAConfig | None = None):
        self.cfg = cfg or HyenaDNAConfig()
        self.net = ops.HipModel(state_dict, self.cfg)

    def forward(self, input_ids: torch.Tensor, input_quals: torch.Tensor) -> torch.Tensor:
        """`TokenClassificationLit.forward` (basic_module.py:90-100).  SYNCHRONISES the stream in the default fp16x3
        mode: the range guard's device counter is read back before the logits are handed out (a batch that left
        the fp16 range is redone with the bf16 split).  Callers that pipeline forwards use
        `self.net.forward(...)` with `set_overflow_target` (what `PredictPipeline.predict_stream` does)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        ids = input_ids.to(dev)
        if ids.dtype not in (torch.int64, torch.uint8):
            ids = ids.to(torch.int64)
        q = input_quals.to(dev, torch.float32)
        return self.net.forward_checked(ids.contiguous(), q.contiguous())  # fp16x3 range guard + bf16x6 redo

    __call__ = forward

    def predict_step(self, batch: dict, batch_idx: int = 0):
        return self.forward(batch["input_ids"], batch["input_quals"]), batch["labels"]

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self


class DeepChopper:
    @staticmethod
    def from_state_dict(state_dict, cfg: HyenaDNAConfig | None = None) -> TokenClassificationModel:
        return TokenClassificationModel(state_dict, cfg)

    @staticmethod
    def from_checkpoint(checkpoint_path: str) -> TokenClassificationModel:
        """dc_hg.py:70-117: a Lightning `.ckpt` (its `state_dict`)."""
        return TokenClassificationModel(*load_checkpoint(checkpoint_path))

    @staticmethod
    def from_pretrained(model_name: str) -> TokenClassificationModel:
        """dc_hg.py:119-163.  `model_name` is a local directory / file holding the checkpoint the
        reference would download (`model.safetensors` written by PyTorchModelHubMixin).  Hub ids
        (`yangliz5/deepchopper`) are resolved through $DEEPCHOPPER_MODEL_DIR/<id>, then through the HF Hub client: its
        cache, and -- unless HF_HUB_OFFLINE is set -- a download, as the reference's mixin does.  The file is read as the mixin wrote it: aliases of
        shared tensors that `safetensors.torch.save_model` dropped are restored, buffers are recomputed if absent,
        extra tensors are ignored, and the architecture is inferred from the tensor shapes and cross-checked
        against a `config.json` next to the weights (`weights.load_checkpoint`)."""
        for cand in (model_name, os.path.join(os.environ.get("DEEPCHOPPER_MODEL_DIR", ""), model_name)):
            if cand and os.path.exists(cand):
                return TokenClassificationModel(*load_checkpoint(cand))
        # PyTorchModelHubMixin.from_pretrained downloads (dc_hg.py:140): so does this, unless the Hub client is told to stay
        # offline (HF_HUB_OFFLINE=1, as in this build's containers) -- then only the local cache is consulted.  Weights and
        # their config only: no remote code is fetched or executed.
        offline = os.environ.get("HF_HUB_OFFLINE", "").strip().upper() in ("1", "ON", "YES", "TRUE")
        try:
            from huggingface_hub import snapshot_download

            path = snapshot_download(model_name, local_files_only=offline, allow_patterns=["*.safetensors", "*.json", "*.ckpt"])
        except Exception as e:
            why = ("the HF Hub client is offline (HF_HUB_OFFLINE) and the checkpoint is not in its cache" if offline
                   else "the download from the HF Hub failed")
            raise FileNotFoundError(
                f"checkpoint '{model_name}' not found locally (set DEEPCHOPPER_MODEL_DIR or pass a path to "
                f"model.safetensors / .ckpt); {why}: {e}"
            ) from None
        return TokenClassificationModel(*load_checkpoint(path))

    @staticmethod
    def synthetic(seed: int = 0) -> TokenClassificationModel:
        """Seeded synthetic weights of the exact reference shapes (benchmarks / tests)."""
        return TokenClassificationModel(synthetic_state_dict(seed))
