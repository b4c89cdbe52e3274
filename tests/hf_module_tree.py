"""A torch module tree with the NAMES and the tensor SHARING of the reference's model object
(`TokenClassificationLit(net=TokenClassificationModule(backbone=AutoModel(...), head=TokenClassificationHead))`,
deepchopper/models/dc_hg.py:140-163, llm/hyena.py:11-27, llm/head.py:39-73) -- test infrastructure: no forward, only
what `state_dict()` / `safetensors.torch.save_model` see.  The one structural fact that matters: HyenaDNA's filter MLP
is `Sequential(Linear, act, Linear, act, Linear, act, Linear)` with ONE `Sin` module `act`, so
`implicit_filter.{1,3,5}.freq` are the same Parameter under three names (public HyenaDNA `HyenaFilter.__init__`)."""
import torch
from torch import nn


class Sin(nn.Module):
    def __init__(self, fo, freq):
        super().__init__()
        self.freq = nn.Parameter(freq * torch.ones(1, fo))


class PosEmb(nn.Module):
    def __init__(self, L, emb):
        super().__init__()
        self.z = nn.Parameter(torch.zeros(1, L, emb))
        self.register_buffer("t", torch.zeros(1, L, 1))


class Modulation(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.register_buffer("deltas", torch.zeros(1, 1, D))


class Filter(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        fo, D = cfg.filter_order, cfg.d_model
        self.bias = nn.Parameter(torch.zeros(D))
        self.pos_emb = PosEmb(cfg.max_seq_len, cfg.emb_dim)
        act = Sin(fo, cfg.activation_freq)  # shared by every slot
        layers = [nn.Linear(cfg.emb_dim, fo), act]
        for _ in range(cfg.num_inner_mlps):
            layers += [nn.Linear(fo, fo), act]
        layers.append(nn.Linear(fo, D, bias=False))
        self.implicit_filter = nn.Sequential(*layers)
        self.modulation = Modulation(D)


class Mixer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        D, W = cfg.d_model, (cfg.hyena_order + 1) * cfg.d_model
        self.in_proj = nn.Linear(D, W)
        self.out_proj = nn.Linear(D, D)
        self.short_filter = nn.Conv1d(W, W, cfg.short_filter_order, groups=W, padding=cfg.short_filter_order - 1)
        self.filter_fn = Filter(cfg)


class Mlp(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc1 = nn.Linear(cfg.d_model, cfg.d_inner)
        self.fc2 = nn.Linear(cfg.d_inner, cfg.d_model)


class Block(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.mixer = Mixer(cfg)
        self.norm1 = nn.LayerNorm(cfg.d_model)
        self.mlp = Mlp(cfg)
        self.norm2 = nn.LayerNorm(cfg.d_model)


class Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.padded_vocab, cfg.d_model)


class LMBackbone(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = Embeddings(cfg)
        self.layers = nn.ModuleList([Block(cfg) for _ in range(cfg.n_layer)])
        self.ln_f = nn.LayerNorm(cfg.d_model)


class HyenaDNAModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = LMBackbone(cfg)


class Head(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.linear1 = nn.Linear(cfg.d_model, cfg.head_hidden)
        self.linear2 = nn.Linear(cfg.head_hidden, cfg.head_hidden)
        self.linear3 = nn.Linear(cfg.head_hidden, cfg.num_class)


class TokenClassificationModule(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = HyenaDNAModel(cfg)
        self.head = Head(cfg)


class Lit(nn.Module):
    """Stands for TokenClassificationLit: `net` plus a parameter-free criterion and some non-model state."""

    def __init__(self, cfg):
        super().__init__()
        self.net = TokenClassificationModule(cfg)
        self.register_buffer("some_metric_state", torch.zeros(3))  # extra tensors a loader must tolerate


def build(cfg, state_dict):
    """Module tree filled from a (complete) numpy state dict; the shared `Sin` takes slot 1's values."""
    m = Lit(cfg)
    own = m.state_dict()
    with torch.no_grad():
        for k, v in own.items():
            if k in state_dict:
                v.copy_(torch.as_tensor(state_dict[k]).reshape(v.shape))
    return m
