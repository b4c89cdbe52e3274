"""CPU oracle (torch fp32) for the token-classification forward F1-F3.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

What it restates
----------------
F1  TokenClassificationModule.forward            deepchopper/models/llm/hyena.py:29-41
      hidden = backbone(input_ids)[0]; logits = head(hidden, input_quals)
F2  the HyenaDNA backbone `LongSafari/hyenadna-small-32k-seqlen-hf` -- THIRD-PARTY remote code
      that the reference loads from the HF Hub (hyena.py:22, `trust_remote_code=True`, no
      revision pinned; transformers 5.12.1 per uv.lock).  It is absent from /root/reference and
      loading it was denied in this pipeline, so this is a restatement of the
      *published* HyenaDNA algorithm (Nguyen et al. 2023, `modeling_hyena.py` of that Hub repo as
      publicly documented): embedding -> n_layer x [LN -> Hyena operator -> +res ; LN -> MLP
      (tanh-GELU) -> +res] -> final LN, with the order-2 Hyena operator
          u = in_proj(x); uc = causal depthwise conv3(u); (x0, x1, v) = split(uc)
          v = v * x1; v = fftconv(v, k) + bias * v; y = out_proj(v * x0)
      and the implicit filter k = MLP_sin(z_t) * (exp(-t |delta|) + shift).
      **PARITY UNPINNED** for F2: no reference test or fixture pins these numbers without the
      Hub weights; constants the reference does pin (hidden 256, context 32768, vocab ids) are
      honoured.  State-dict key names follow the HF module tree so a user-supplied checkpoint
      loads unchanged.
F3  TokenClassificationHead.forward               deepchopper/models/llm/head.py:75-102
      pinned by golden vectors generated from the reference's own head.py
      (tests/golden/make_head_golden.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class HyenaDNAConfig:
    """hyenadna-small-32k-seqlen (public config.json); `d_model` is pinned by
    deepchopper/models/dc_hg.py:96,147."""

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

    @property
    def padded_vocab(self) -> int:
        v, m = self.vocab_size, self.pad_vocab_size_multiple
        return v if v % m == 0 else v + m - v % m


BB = "net.backbone.backbone."
HD = "net.head."


def fftconv(u: torch.Tensor, k: torch.Tensor, D: torch.Tensor) -> torch.Tensor:
    """Published HyenaDNA `fftconv`: causal convolution through an FFT of size 2L, plus the
    per-channel skip `u * D`.  u: [B, C, L], k: [C, L], D: [C]."""
    seqlen = u.shape[-1]
    fft_size = 2 * seqlen
    k_f = torch.fft.rfft(k, n=fft_size) / fft_size
    u_f = torch.fft.rfft(u.to(dtype=k.dtype), n=fft_size)
    y = torch.fft.irfft(u_f * k_f, n=fft_size, norm="forward")[..., :seqlen]
    out = y + u * D.unsqueeze(-1)
    return out.to(dtype=u.dtype)


def hyena_filter(sd: dict, prefix: str, L: int, cfg: HyenaDNAConfig) -> torch.Tensor:
    """HyenaFilter.filter(L): h = implicit_filter(z[:, :L]); h = h * (exp(-t |deltas|) + shift).
    Returns [L, d_model]."""
    z = sd[prefix + "pos_emb.z"][0, :L]  # [L, emb_dim]
    t = sd[prefix + "pos_emb.t"][0, :L]  # [L, 1]
    h = z
    n_lin = cfg.num_inner_mlps + 1
    for i in range(n_lin):
        w = sd[prefix + f"implicit_filter.{2 * i}.weight"]
        b = sd[prefix + f"implicit_filter.{2 * i}.bias"]
        freq = sd[prefix + f"implicit_filter.{2 * i + 1}.freq"]
        h = torch.sin(freq * F.linear(h, w, b))
    h = F.linear(h, sd[prefix + f"implicit_filter.{2 * n_lin}.weight"])  # bias=False
    deltas = sd[prefix + "modulation.deltas"][0]  # [1, d_model]
    shift = 0.05
    decay = torch.exp(-t * deltas.abs())
    return h * (decay + shift)


def hyena_operator(sd: dict, prefix: str, x: torch.Tensor, cfg: HyenaDNAConfig) -> torch.Tensor:
    """HyenaOperator.forward, order 2.  x: [B, L, D]."""
    B, L, D = x.shape
    l_filter = min(L, cfg.max_seq_len)
    u = F.linear(x, sd[prefix + "in_proj.weight"], sd[prefix + "in_proj.bias"])
    u = u.transpose(1, 2)  # b d l
    w = cfg.short_filter_order
    uc = F.conv1d(
        u, sd[prefix + "short_filter.weight"], sd[prefix + "short_filter.bias"], padding=w - 1, groups=u.shape[1]
    )[..., :l_filter]
    x0, x1, v = uc.split(D, dim=1)
    k = hyena_filter(sd, prefix + "filter_fn.", l_filter, cfg).transpose(0, 1)  # [D, L]
    bias = sd[prefix + "filter_fn.bias"]
    v = v * x1
    v = fftconv(v, k, bias)
    y = (v * x0).transpose(1, 2)
    return F.linear(y, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def backbone_forward(sd: dict, input_ids: torch.Tensor, cfg: HyenaDNAConfig) -> torch.Tensor:
    eps = cfg.layer_norm_epsilon
    h = F.embedding(input_ids, sd[BB + "embeddings.word_embeddings.weight"])
    for i in range(cfg.n_layer):
        p = BB + f"layers.{i}."
        residual = h
        hn = F.layer_norm(residual, (cfg.d_model,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        h = hyena_operator(sd, p + "mixer.", hn, cfg)
        residual = h + residual
        hn = F.layer_norm(residual, (cfg.d_model,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
        m = F.linear(hn, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        m = F.gelu(m, approximate=cfg.gelu_approximate)
        m = F.linear(m, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        h = m + residual
    return F.layer_norm(h, (cfg.d_model,), sd[BB + "ln_f.weight"], sd[BB + "ln_f.bias"], eps)


def head_forward(sd: dict, x: torch.Tensor, input_quals: torch.Tensor) -> torch.Tensor:
    """head.py:94-102 with use_qual=True, use_identity_layer_for_qual=True (dc_hg.py:146-153)."""
    out = F.relu(F.linear(x, sd[HD + "linear1.weight"], sd[HD + "linear1.bias"]))
    residual = out + input_quals.unsqueeze(-1)
    out = F.relu(F.linear(residual, sd[HD + "linear2.weight"], sd[HD + "linear2.bias"]) + residual)
    return F.linear(out, sd[HD + "linear3.weight"], sd[HD + "linear3.bias"])


@torch.no_grad()
def forward(sd: dict, input_ids, input_quals, cfg: HyenaDNAConfig | None = None, dtype=torch.float32) -> torch.Tensor:
    """logits [B, L, 2] = head(backbone(input_ids), input_quals).  dtype=float32 is the reference's
    arithmetic; float64 gives an error yardstick for the GPU paths."""
    cfg = cfg or HyenaDNAConfig()
    ids = torch.as_tensor(input_ids, dtype=torch.long)
    q = torch.as_tensor(input_quals, dtype=torch.float32).to(dtype)
    sd = {k: torch.as_tensor(v).to(dtype) for k, v in sd.items()}
    hidden = backbone_forward(sd, ids, cfg)
    return head_forward(sd, hidden, q)


@torch.no_grad()
def causal_conv_direct(v: torch.Tensor, k: torch.Tensor, D: torch.Tensor) -> torch.Tensor:
    """Time-domain statement of `fftconv` (fp64 accumulation) used to cross-check the FFT form:
    y[t] = sum_{s<=t} k[s] v[t-s] + D v[t]."""
    B, C, L = v.shape
    vd, kd = v.double(), k.double()
    y = torch.zeros_like(vd)
    for s in range(L):
        y[..., s:] += kd[:, s : s + 1] * vd[..., : L - s]
    return (y + vd * D.double().unsqueeze(-1)).float()
