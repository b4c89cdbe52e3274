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
        if not isinstance(v, np.ndarray) and not np.isscalar(v):
            continue  # non-tensor entries of a Lightning checkpoint's state (e.g. metric bookkeeping)
        out[k] = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
    return out


def _read_file(path: str) -> tuple[dict[str, np.ndarray], dict[str, str]]:
    """(tensors, safetensors metadata).  `safetensors.torch.save_model` -- what PyTorchModelHubMixin.save_pretrained
    calls -- drops every alias of a shared tensor and records `dropped name -> kept name` in the file's metadata."""
    if path.endswith(".safetensors"):
        from safetensors import safe_open

        tensors, meta = {}, {}
        with safe_open(path, framework="pt") as fh:  # "pt": bf16 / fp16 checkpoints load too (numpy has no bf16)
            meta = dict(fh.metadata() or {})
            for k in fh.keys():
                tensors[k] = fh.get_tensor(k)
        return _strip(tensors), meta
    import torch

    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj:
        obj = obj["state_dict"]
    return _strip(obj), {}


def _find_weights(path: str) -> str:
    if os.path.isdir(path):
        for name in ("model.safetensors", "pytorch_model.bin"):
            cand = os.path.join(path, name)
            if os.path.exists(cand):
                return cand
        raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
    return path


def _norm_name(k: str) -> str:
    k = k.replace("._orig_mod.", ".").replace("_orig_mod.", "")
    if k.startswith(("backbone.", "head.")):
        k = "net." + k
    return k


def normalise_names(sd: dict) -> dict:
    """Bring a state dict to the `net.backbone.backbone.* / net.head.*` names of `TokenClassificationLit`
    (basic_module.py:58): accepts the bare `TokenClassificationModule` tree (no `net.`) and torch.compile's
    `_orig_mod.` infix (basic_module.py `compile=True`)."""
    return {_norm_name(k): v for k, v in sd.items()}


def resolve_shared_tensors(sd: dict[str, np.ndarray], metadata: dict[str, str] | None = None) -> dict[str, np.ndarray]:
    """Restore the aliases `safetensors.torch.save_model` removed.  HyenaDNA's filter MLP applies ONE `Sin` module
    after each of its linears, so `implicit_filter.{1,3,5}.freq` are one tensor under three names and only the
    first survives in `model.safetensors`; its metadata maps each dropped name to the kept one."""
    out = dict(sd)
    for dropped, kept in (metadata or {}).items():
        if dropped not in out and kept in out:
            out[dropped] = out[kept]
    # without (or beyond) the metadata: any `...implicit_filter.<odd>.freq` stands for all of that filter's Sin slots
    groups: dict[str, list[str]] = {}
    for k in out:
        if ".implicit_filter." in k and k.endswith(".freq"):
            groups.setdefault(k.split(".implicit_filter.")[0], []).append(k)
    for prefix, have in groups.items():
        idx = sorted(int(k.split(".implicit_filter.")[1].split(".")[0]) for k in have)
        # Sin slots sit at odd indices up to the last linear (an even index): fill every odd slot below it
        last_linear = max([int(k.split(".implicit_filter.")[1].split(".")[0]) for k in out
                           if k.startswith(prefix + ".implicit_filter.") and k.endswith(".weight")] or [idx[-1] + 1])
        src = out[f"{prefix}.implicit_filter.{idx[0]}.freq"]
        for j in range(1, last_linear, 2):
            out.setdefault(f"{prefix}.implicit_filter.{j}.freq", src)
    return out


def infer_config(sd: dict[str, np.ndarray]) -> HyenaDNAConfig:
    """Architecture constants read off the tensor shapes (the reference hard-wires them in dc_hg.py:142-152 and the
    Hub config of LongSafari/hyenadna-small-32k-seqlen-hf; a checkpoint is the only thing that can confirm them)."""
    def need(k):
        if k not in sd:
            raise KeyError(f"checkpoint has no tensor '{k}' (names must follow net.backbone.backbone.* / net.head.*; "
                           f"found e.g. {sorted(sd)[:3]})")
        return sd[k]

    emb = need(BB + "embeddings.word_embeddings.weight")
    n_layer = 0
    while BB + f"layers.{n_layer}.mixer.in_proj.weight" in sd:
        n_layer += 1
    if n_layer == 0:
        raise KeyError(f"checkpoint has no tensor '{BB}layers.0.mixer.in_proj.weight'")
    f = BB + "layers.0.mixer.filter_fn."
    w0 = need(f + "implicit_filter.0.weight")
    n_lin = 0
    while f + f"implicit_filter.{2 * n_lin}.weight" in sd:
        n_lin += 1
    cfg = HyenaDNAConfig()
    cfg.d_model = int(emb.shape[1])
    # vocab_size itself is not recoverable from the padded embedding; keep the multiple and report the padded size
    cfg.vocab_size = int(emb.shape[0])
    cfg.pad_vocab_size_multiple = 1
    cfg.n_layer = n_layer
    cfg.d_inner = int(need(BB + "layers.0.mlp.fc1.weight").shape[0])
    cfg.filter_order = int(w0.shape[0])
    cfg.emb_dim = int(w0.shape[1])
    cfg.num_inner_mlps = n_lin - 2
    cfg.short_filter_order = int(need(BB + "layers.0.mixer.short_filter.weight").shape[-1])
    cfg.hyena_order = int(need(BB + "layers.0.mixer.in_proj.weight").shape[0]) // cfg.d_model - 1
    if f + "pos_emb.z" in sd:
        cfg.max_seq_len = int(np.asarray(sd[f + "pos_emb.z"]).reshape(-1, cfg.emb_dim).shape[0])
    elif f + "pos_emb.t" in sd:
        cfg.max_seq_len = int(np.asarray(sd[f + "pos_emb.t"]).size)
    cfg.head_hidden = int(need(HD + "linear1.weight").shape[0])
    cfg.num_class = int(need(HD + "linear3.weight").shape[0])
    return cfg


# HyenaDNAConfig field -> name in the Hub `config.json` of the backbone (configuration_hyena.py); same names
_JSON_FIELDS = ("d_model", "d_inner", "n_layer", "hyena_order", "short_filter_order", "filter_order", "emb_dim",
                "num_inner_mlps", "max_seq_len", "activation_freq", "layer_norm_epsilon")


def check_config_json(cfg: HyenaDNAConfig, js: dict, where: str = "config.json") -> HyenaDNAConfig:
    """Every architecture field `js` states must agree with what the tensors say; fields only the JSON knows
    (activation_freq, layer_norm_epsilon, vocab_size / pad multiple) are taken from it.  A PyTorchModelHubMixin
    `config.json` (the Lightning module's init kwargs) simply has none of these keys."""
    shape_fields = ("d_model", "d_inner", "n_layer", "hyena_order", "short_filter_order", "filter_order", "emb_dim",
                    "num_inner_mlps", "max_seq_len")
    for name in _JSON_FIELDS:
        if name not in js:
            continue
        want = js[name]
        if name in shape_fields:
            if int(want) != int(getattr(cfg, name)):
                raise ValueError(f"{where}: field '{name}' = {want}, but the checkpoint's tensors have {name} = {getattr(cfg, name)}")
        else:
            setattr(cfg, name, float(want))
    if "vocab_size" in js:
        v, mult = int(js["vocab_size"]), int(js.get("pad_vocab_size_multiple", 1))
        padded = v if v % mult == 0 else v + mult - v % mult
        if padded != cfg.padded_vocab:
            raise ValueError(f"{where}: field 'vocab_size' = {v} (padded to {padded} by pad_vocab_size_multiple = {mult}), "
                             f"but the embedding table has {cfg.padded_vocab} rows")
        cfg.vocab_size, cfg.pad_vocab_size_multiple = v, mult
    return cfg


def expected_shapes(cfg: HyenaDNAConfig) -> dict[str, tuple[tuple[int, ...], str]]:
    """name -> (shape, the config expression it comes from)."""
    D, DI, fo, e, H = cfg.d_model, cfg.d_inner, cfg.filter_order, cfg.emb_dim, cfg.head_hidden
    W = (cfg.hyena_order + 1) * D
    out = {BB + "embeddings.word_embeddings.weight": ((cfg.padded_vocab, D), "(padded vocab_size, d_model)"),
           BB + "ln_f.weight": ((D,), "(d_model,)"), BB + "ln_f.bias": ((D,), "(d_model,)")}
    for i in range(cfg.n_layer):
        p = BB + f"layers.{i}."
        for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "mlp.fc2.bias", "mixer.out_proj.bias",
                  "mixer.filter_fn.bias"):
            out[p + n] = ((D,), "(d_model,)")
        out[p + "mlp.fc1.weight"] = ((DI, D), "(d_inner, d_model)")
        out[p + "mlp.fc1.bias"] = ((DI,), "(d_inner,)")
        out[p + "mlp.fc2.weight"] = ((D, DI), "(d_model, d_inner)")
        m = p + "mixer."
        out[m + "in_proj.weight"] = ((W, D), "((hyena_order + 1) * d_model, d_model)")
        out[m + "in_proj.bias"] = ((W,), "((hyena_order + 1) * d_model,)")
        out[m + "out_proj.weight"] = ((D, D), "(d_model, d_model)")
        out[m + "short_filter.weight"] = ((W, 1, cfg.short_filter_order), "((hyena_order + 1) * d_model, 1, short_filter_order)")
        out[m + "short_filter.bias"] = ((W,), "((hyena_order + 1) * d_model,)")
        f = m + "filter_fn."
        out[f + "pos_emb.z"] = ((1, cfg.max_seq_len, e), "(1, max_seq_len, emb_dim)")
        out[f + "pos_emb.t"] = ((1, cfg.max_seq_len, 1), "(1, max_seq_len, 1)")
        out[f + "modulation.deltas"] = ((1, 1, D), "(1, 1, d_model)")
        dims = [e] + [fo] * (cfg.num_inner_mlps + 1)
        for j in range(cfg.num_inner_mlps + 1):
            out[f + f"implicit_filter.{2 * j}.weight"] = ((fo, dims[j]), "(filter_order, emb_dim | filter_order)")
            out[f + f"implicit_filter.{2 * j}.bias"] = ((fo,), "(filter_order,)")
            out[f + f"implicit_filter.{2 * j + 1}.freq"] = ((1, fo), "(1, filter_order)")
        out[f + f"implicit_filter.{2 * (cfg.num_inner_mlps + 1)}.weight"] = ((D, fo), "(d_model, filter_order)")
    out[HD + "linear1.weight"] = ((H, D), "(head lin1_size, d_model)")
    out[HD + "linear1.bias"] = ((H,), "(head lin1_size,)")
    out[HD + "linear2.weight"] = ((H, H), "(head lin2_size, lin1_size)")
    out[HD + "linear2.bias"] = ((H,), "(head lin2_size,)")
    out[HD + "linear3.weight"] = ((cfg.num_class, H), "(num_class, head lin2_size)")
    out[HD + "linear3.bias"] = ((cfg.num_class,), "(num_class,)")
    return out


def complete_and_validate(sd: dict[str, np.ndarray], cfg: HyenaDNAConfig) -> dict[str, np.ndarray]:
    """Recompute the input-independent buffers a checkpoint may omit (`pos_emb.t`, `modulation.deltas`: registered
    buffers; `pos_emb.z` only if absent -- it is a trained parameter), then check every tensor the model needs
    against the configuration.  Extra tensors (loss / metric state of the Lightning module) are ignored."""
    out = dict(sd)
    z0 = t0 = None
    for i in range(cfg.n_layer):
        f = BB + f"layers.{i}.mixer.filter_fn."
        if f + "pos_emb.t" not in out or f + "pos_emb.z" not in out:
            if z0 is None:
                z0, t0 = positional_embedding(cfg)
            out.setdefault(f + "pos_emb.t", t0.reshape(1, cfg.max_seq_len, 1))
            out.setdefault(f + "pos_emb.z", z0)
        out.setdefault(f + "modulation.deltas", modulation_deltas(cfg))
    for name, (shape, expr) in expected_shapes(cfg).items():
        if name not in out:
            raise KeyError(f"checkpoint is missing tensor '{name}' {shape} = {expr}")
        got = tuple(np.asarray(out[name]).shape)
        if int(np.prod(got)) != int(np.prod(shape)) or (len(got) == len(shape) and got != shape):
            raise ValueError(f"{name}: shape {got} does not match the configuration, expected {shape} = {expr}")
    return out


def load_checkpoint(path: str, cfg: HyenaDNAConfig | None = None) -> tuple[dict[str, np.ndarray], HyenaDNAConfig]:
    """What `DeepChopper.from_pretrained` / `from_checkpoint` read (dc_hg.py:70-163): a directory holding
    `model.safetensors` as `PyTorchModelHubMixin.save_pretrained` writes it (shared tensors de-duplicated), a
    `.safetensors` file, or a Lightning `.ckpt` / torch `.pt` with a `state_dict` (dc_hg.py:114-116).  Returns the
    completed state dict and the configuration inferred from it, cross-checked against a `config.json` /
    `backbone_config.json` next to the weights when one states HyenaDNA fields."""
    import json

    wpath = _find_weights(path)
    sd, meta = _read_file(wpath)
    aliases = {_norm_name(k): _norm_name(v) for k, v in meta.items() if isinstance(v, str)}  # dropped name -> kept name
    sd = resolve_shared_tensors(normalise_names(sd), aliases)
    inferred = infer_config(sd)
    if cfg is not None:  # an explicit configuration must agree with the tensors
        check_config_json(inferred, {k: getattr(cfg, k) for k in _JSON_FIELDS if k not in ("activation_freq", "layer_norm_epsilon")},
                          "explicit configuration")
        for k in ("activation_freq", "layer_norm_epsilon", "gelu_approximate", "filter_shift"):
            setattr(inferred, k, getattr(cfg, k))
    folder = os.path.dirname(os.path.abspath(wpath))
    for name in ("config.json", "backbone_config.json"):
        cj = os.path.join(folder, name)
        if os.path.exists(cj):
            with open(cj) as fh:
                js = json.load(fh)
            if isinstance(js, dict):
                check_config_json(inferred, js, cj)
    return complete_and_validate(sd, inferred), inferred


def load_state_dict(path: str) -> dict[str, np.ndarray]:
    """The completed state dict of `load_checkpoint` (kept for callers that bring their own configuration)."""
    return load_checkpoint(path)[0]


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
