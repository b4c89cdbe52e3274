"""Model configuration, checkpoint loading and seeded synthetic weights.

The architecture constants are those of `DeepChopper.from_pretrained` in the reference
(deepchopper/models/dc_hg.py:140-163): HyenaDNA `hyenadna-small-32k-seqlen` backbone (hidden 256)
plus `TokenClassificationHead(256, 2, 1024, 1024, use_identity_layer_for_qual=True,
use_qual=True)`.  Tensor names follow the checkpoint's module tree
(`net.backbone.backbone.*`, `net.head.*`) so a user-supplied `model.safetensors` / `.ckpt`
loads unchanged; real weights live only on the HF Hub and are not available offline.
"""
from __future__ import annotations

import math
import os
from dataclasses import asdict, dataclass

import numpy as np

BB = "net.backbone.backbone."
HD = "net.head."


@dataclass
class HyenaDNAConfig:
    vocab_size: int = 12
    pad_vocab_size_multiple: int = 8
    d_model: int = 256
    d_inner: int = 1024
    n_layer: int = 4
    hyena_order: int = 2
    short_filter_order: int = 3
    filter_order: int = 64
    emb_dim: int = 5
    num_inner_mlps: int = 2
    max_seq_len: int = 32770
    activation_freq: float = 10.0
    layer_norm_epsilon: float = 1e-5
    gelu_approximate: str = "tanh"
    head_hidden: int = 1024
    num_class: int = 2
    filter_shift: float = 0.05

    @property
    def padded_vocab(self) -> int:
        v, m = self.vocab_size, self.pad_vocab_size_multiple
        return v if v % m == 0 else v + m - v % m

    def to_dict(self):
        return asdict(self)


def positional_embedding(cfg: HyenaDNAConfig) -> tuple[np.ndarray, np.ndarray]:
    """Initial value of HyenaPositionalEmbedding.z / .t (z is a trained parameter in real
    checkpoints; this is its initialisation, used for synthetic weights)."""
    L = cfg.max_seq_len
    t = np.linspace(0.0, 1.0, L, dtype=np.float64)[:, None]
    bands = (cfg.emb_dim - 1) // 2
    t_rescaled = np.linspace(0.0, L - 1, L, dtype=np.float64)[:, None]
    w = 2.0 * math.pi * t_rescaled / L
    f = np.linspace(1e-4, bands - 1, bands, dtype=np.float64)[None, :]
    z = np.concatenate([t, np.cos(-f * w), np.sin(-f * w)], axis=-1)
    return z.astype(np.float32)[None], t.astype(np.float32)[None]


def modulation_deltas(cfg: HyenaDNAConfig) -> np.ndarray:
    target, fast, slow = 1e-2, 0.3, 1.5
    max_decay = math.log(target) / fast
    min_decay = math.log(target) / slow
    return np.linspace(min_decay, max_decay, cfg.d_model, dtype=np.float64).astype(np.float32)[None, None]


def synthetic_state_dict(seed: int = 0, cfg: HyenaDNAConfig | None = None) -> dict[str, np.ndarray]:
    """Deterministic synthetic weights of the exact reference shapes.

    Scales are variance-preserving (N(0, 1/fan_in) matrices, non-trivial LayerNorm affine and
    biases) so that activations stay O(1) through the stack and the logits change sign along a
    read -- a constant-label model would not exercise the argmax / interval path."""
    cfg = cfg or HyenaDNAConfig()
    rng = np.random.default_rng(seed)
    D, DI, HH = cfg.d_model, cfg.d_inner, cfg.head_hidden

    def lin(out_f, in_f, gain=1.0):
        return (rng.standard_normal((out_f, in_f)) * (gain / math.sqrt(in_f))).astype(np.float32)

    def vec(n, s):
        return (rng.standard_normal(n) * s).astype(np.float32)

    sd: dict[str, np.ndarray] = {}
    sd[BB + "embeddings.word_embeddings.weight"] = rng.standard_normal((cfg.padded_vocab, D)).astype(np.float32)
    z, t = positional_embedding(cfg)
    for i in range(cfg.n_layer):
        p = BB + f"layers.{i}."
        sd[p + "norm1.weight"] = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
        sd[p + "norm1.bias"] = vec(D, 0.1)
        sd[p + "norm2.weight"] = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
        sd[p + "norm2.bias"] = vec(D, 0.1)
        m = p + "mixer."
        sd[m + "in_proj.weight"] = lin(3 * D, D)
        sd[m + "in_proj.bias"] = vec(3 * D, 0.1)
        sd[m + "out_proj.weight"] = lin(D, D, 0.5)
        sd[m + "out_proj.bias"] = vec(D, 0.05)
        sd[m + "short_filter.weight"] = (rng.standard_normal((3 * D, 1, cfg.short_filter_order)) * 0.5).astype(np.float32)
        sd[m + "short_filter.bias"] = vec(3 * D, 0.1)
        f = m + "filter_fn."
        sd[f + "bias"] = vec(D, 0.5)
        sd[f + "pos_emb.z"] = z.copy()
        sd[f + "pos_emb.t"] = t.copy()
        fo = cfg.filter_order
        dims = [cfg.emb_dim] + [fo] * (cfg.num_inner_mlps + 1)
        for j in range(cfg.num_inner_mlps + 1):
            bound = 1.0 / math.sqrt(dims[j])
            sd[f + f"implicit_filter.{2 * j}.weight"] = rng.uniform(-bound, bound, (fo, dims[j])).astype(np.float32)
            sd[f + f"implicit_filter.{2 * j}.bias"] = rng.uniform(-bound, bound, fo).astype(np.float32)
            sd[f + f"implicit_filter.{2 * j + 1}.freq"] = np.full((1, fo), cfg.activation_freq, dtype=np.float32)
        bound = 1.0 / math.sqrt(fo)
        sd[f + f"implicit_filter.{2 * (cfg.num_inner_mlps + 1)}.weight"] = rng.uniform(-bound, bound, (D, fo)).astype(np.float32)
        sd[f + "modulation.deltas"] = modulation_deltas(cfg)
        sd[p + "mlp.fc1.weight"] = lin(DI, D)
        sd[p + "mlp.fc1.bias"] = vec(DI, 0.05)
        sd[p + "mlp.fc2.weight"] = lin(D, DI, 0.5)
        sd[p + "mlp.fc2.bias"] = vec(D, 0.05)
    sd[BB + "ln_f.weight"] = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    sd[BB + "ln_f.bias"] = vec(D, 0.1)
    sd[HD + "linear1.weight"] = lin(HH, D)
    sd[HD + "linear1.bias"] = vec(HH, 0.05)
    sd[HD + "linear2.weight"] = lin(HH, HH, 0.7)
    sd[HD + "linear2.bias"] = vec(HH, 0.05)
    sd[HD + "linear3.weight"] = lin(cfg.num_class, HH, 2.0)
    sd[HD + "linear3.bias"] = vec(cfg.num_class, 0.05)
    return sd


def token_gated_state_dict(seed: int = 0, token_id: int = 11, backbone_gain: float = 0.003,
                            cfg: HyenaDNAConfig | None = None) -> dict[str, np.ndarray]:
    """Synthetic weights whose per-base label is a DESIGNED function of the input: label 1 exactly where the
    base's token is `token_id` (default 11 = 'N'), with a logit margin of ~1 -- far above any arithmetic noise.
    For flow tests and end-to-end benchmarks of predict -> chop, where adapters are planted as runs of that
    base and the expected records must not depend on near-tie logits.  Same shapes, same kernels: the
    embedding rows are Walsh functions (mean 0, variance 1, mutually orthogonal), the backbone blocks keep
    their seeded weights but write into the residual stream with a small gain, and the head reads the token
    off the final LayerNorm output (linear1 = matched filters with a -0.5 bias, linear2 = 0 so that its
    residual + ReLU passes linear1's output on, linear3 picks the gate unit)."""
    cfg = cfg or HyenaDNAConfig()
    sd = synthetic_state_dict(seed, cfg)
    D, HH = cfg.d_model, cfg.head_hidden
    assert D & (D - 1) == 0 and cfg.padded_vocab < D
    h = np.array([[1.0]])
    while h.shape[0] < D:
        h = np.block([[h, h], [h, -h]])
    emb = h[1: cfg.padded_vocab + 1].astype(np.float32)  # row 0 (all ones) has no variance
    sd[BB + "embeddings.word_embeddings.weight"] = emb
    for i in range(cfg.n_layer):
        p = BB + f"layers.{i}."
        for k in ("mixer.out_proj.weight", "mixer.out_proj.bias", "mlp.fc2.weight", "mlp.fc2.bias"):
            sd[p + k] = (sd[p + k] * backbone_gain).astype(np.float32)
    sd[BB + "ln_f.weight"] = np.ones(D, np.float32)
    sd[BB + "ln_f.bias"] = np.zeros(D, np.float32)
    w1 = np.zeros((HH, D), np.float32)
    b1 = np.zeros(HH, np.float32)
    w1[: cfg.padded_vocab] = emb / D
    b1[: cfg.padded_vocab] = -0.5
    sd[HD + "linear1.weight"], sd[HD + "linear1.bias"] = w1, b1
    sd[HD + "linear2.weight"] = np.zeros((HH, HH), np.float32)
    sd[HD + "linear2.bias"] = np.zeros(HH, np.float32)
    w3 = np.zeros((cfg.num_class, HH), np.float32)
    w3[1, token_id] = 4.0
    sd[HD + "linear3.weight"] = w3
    sd[HD + "linear3.bias"] = np.array([1.0, 0.0], np.float32)[: cfg.num_class]
    return sd


def _strip(sd: dict) -> dict[str, np.ndarray]:
    out = {}
    for k, v in sd.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().float().numpy()
        out[k] = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
    return out


def load_state_dict(path: str) -> dict[str, np.ndarray]:
    """Load a DeepChopper checkpoint from a local path: a directory holding `model.safetensors`
    (the layout `PyTorchModelHubMixin.save_pretrained` writes, dc_hg.py:140), a `.safetensors`
    file, or a Lightning `.ckpt` / torch `.pt` with a `state_dict` (dc_hg.py:114-116)."""
    if os.path.isdir(path):
        for name in ("model.safetensors", "pytorch_model.bin"):
            cand = os.path.join(path, name)
            if os.path.exists(cand):
                path = cand
                break
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file

        return _strip(load_file(path))
    import torch

    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj:
        obj = obj["state_dict"]
    return _strip(obj)


def save_state_dict(sd: dict[str, np.ndarray], path: str) -> None:
    from safetensors.numpy import save_file

    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, path)


def required_keys(cfg: HyenaDNAConfig) -> list[str]:
    keys = [BB + "embeddings.word_embeddings.weight", BB + "ln_f.weight", BB + "ln_f.bias"]
    for i in range(cfg.n_layer):
        p = BB + f"layers.{i}."
        keys += [p + n for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias",
                                 "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")]
        m = p + "mixer."
        keys += [m + n for n in ("in_proj.weight", "in_proj.bias", "out_proj.weight", "out_proj.bias",
                                 "short_filter.weight", "short_filter.bias", "filter_fn.bias",
                                 "filter_fn.pos_emb.z", "filter_fn.pos_emb.t", "filter_fn.modulation.deltas")]
        for j in range(cfg.num_inner_mlps + 1):
            keys += [m + f"filter_fn.implicit_filter.{2 * j}.weight", m + f"filter_fn.implicit_filter.{2 * j}.bias",
                     m + f"filter_fn.implicit_filter.{2 * j + 1}.freq"]
        keys.append(m + f"filter_fn.implicit_filter.{2 * (cfg.num_inner_mlps + 1)}.weight")
    keys += [HD + n for n in ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                              "linear3.weight", "linear3.bias")]
    return keys
