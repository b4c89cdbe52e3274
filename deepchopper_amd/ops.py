"""Tensor-level wrappers over the C ABI.

torch is used here only as plumbing: device allocations, the current HIP stream and host<->device
copies.  All arithmetic happens inside libdeepchopper_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib

SM_MAX_LEN = 65536  # longest read the smoothing kernel's LDS bit-vectors hold (csrc/smooth.hip)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return t.data_ptr()


def _dev() -> torch.device:
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def pack_reads(reads: list[bytes]) -> tuple[np.ndarray, np.ndarray]:
    """Concatenate byte strings -> (uint8 buffer, int64 offsets[n+1])."""
    lens = np.fromiter((len(r) for r in reads), dtype=np.int64, count=len(reads))
    offs = np.zeros(len(reads) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    buf = np.frombuffer(b"".join(reads), dtype=np.uint8) if len(reads) else np.zeros(0, np.uint8)
    return buf, offs


def to_device(a: np.ndarray) -> torch.Tensor:
    a = np.ascontiguousarray(a)
    if not a.flags.writeable:
        a = a.copy()
    t = torch.from_numpy(a)
    return t.to(_dev(), non_blocking=False)


def pt_pack(logits, labels, input_ids, input_quals, table: np.ndarray, out: torch.Tensor) -> None:
    """dc_pt_pack on the current stream: the storages data/0..3 of every `.pt` file of a fused launch, cut out of the launch's
    dense device arrays and laid out as the files hold them, at the files' offsets of `out` (uint8, device).
    `table`: int64 [n_files, 4] = row0, rows, first column, byte offset (multiple of 64) -- host array; uploaded here."""
    table = np.ascontiguousarray(table, np.int64)
    d_table = torch.from_numpy(table).pin_memory().to(logits.device, non_blocking=True)  # (no wait for the stream's earlier work)
    n, stride = input_ids.shape
    assert logits.is_contiguous() and labels.is_contiguous() and input_ids.is_contiguous() and input_quals.is_contiguous()
    assert logits.shape == (n, stride, 2) and labels.dtype == torch.int8 and out.dtype == torch.uint8 and out.is_contiguous()
    _lib.check(_lib.lib().dc_pt_pack(_stream(), _ptr(logits), _ptr(labels), _ptr(input_ids), int(input_ids.dtype == torch.int64),
                                     _ptr(input_quals), n, stride, int(table.shape[0]), C.c_void_p(table.ctypes.data), _ptr(d_table),
                                     _ptr(out), out.numel()))


# ------------------------------------------------------------------------------------------
# tokenise
# ------------------------------------------------------------------------------------------
def normalize_seq_batch(seq: torch.Tensor, offsets: torch.Tensor, iupac: bool):
    n = offsets.numel() - 1
    out = torch.empty_like(seq)
    out_len = torch.empty(n, dtype=torch.int32, device=seq.device)
    _lib.check(_lib.lib().dc_normalize_seq(_stream(), _ptr(seq), _ptr(offsets), n, int(iupac), _ptr(out), _ptr(out_len)))
    return out, out_len


def encode_qual_bytes(qual: torch.Tensor, qual_offset: int) -> torch.Tensor:
    out = torch.empty_like(qual)
    _lib.check(_lib.lib().dc_encode_qual(_stream(), _ptr(qual), qual.numel(), qual_offset, _ptr(out)))
    return out


def tokenize_batch(seq, qual, offsets, *, max_length: int, row_stride: int, pad_left: bool = True,
                   ids_i64: bool = True, qual_offset: int = 33, want_labels: bool = True):
    """Fused T1-T5.  Returns dict(input_ids, input_quals, labels, n_tokens, status)."""
    n = offsets.numel() - 1
    dev = seq.device
    ids = torch.empty((n, row_stride), dtype=torch.int64 if ids_i64 else torch.uint8, device=dev)
    quals = torch.empty((n, row_stride), dtype=torch.float32, device=dev)
    labels = torch.empty((n, row_stride), dtype=torch.int8, device=dev) if want_labels else None
    ntok = torch.empty(n, dtype=torch.int32, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().dc_tokenize_batch(
        _stream(), _ptr(seq), _ptr(qual), _ptr(offsets), n, max_length, qual_offset, row_stride,
        int(pad_left), int(ids_i64), _ptr(ids), _ptr(quals), _ptr(labels), _ptr(ntok), _ptr(status)))
    return {"input_ids": ids, "input_quals": quals, "labels": labels, "n_tokens": ntok, "status": status}


# ------------------------------------------------------------------------------------------
# k-mer encoder
# ------------------------------------------------------------------------------------------
def kmer_validate(seq, offsets, k: int, bases: bytes) -> torch.Tensor:
    n = offsets.numel() - 1
    status = torch.empty(n, dtype=torch.int32, device=seq.device)
    _lib.check(_lib.lib().dc_kmer_validate(_stream(), _ptr(seq), _ptr(offsets), n, k, bases, len(bases), _ptr(status)))
    return status


def kmer_encode(seq, qual, offsets, out_row, n_rows: int, k: int, bases: bytes, qual_offset: int, width: int,
                max_seq_len: int):
    n = offsets.numel() - 1
    dev = seq.device
    inp = torch.empty((n_rows, 2, width), dtype=torch.int32, device=dev)
    qout = torch.empty((n_rows, max_seq_len), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().dc_kmer_encode(_stream(), _ptr(seq), _ptr(qual), _ptr(offsets), _ptr(out_row), n, k, bases,
                                         len(bases), qual_offset, width, max_seq_len, _ptr(inp), _ptr(qout)))
    return inp, qout


def kmer_vectorize_targets(ranges: torch.Tensor, width: int) -> torch.Tensor:
    n_rows, n_targets, _ = ranges.shape
    out = torch.empty((n_rows, n_targets, width), dtype=torch.int32, device=ranges.device)
    _lib.check(_lib.lib().dc_kmer_vectorize_targets(_stream(), _ptr(ranges), n_rows, n_targets, width, _ptr(out)))
    return out


# ------------------------------------------------------------------------------------------
# smoothing / intervals
# ------------------------------------------------------------------------------------------
def smooth_intervals_from_logits(logits, first, length, *, window=21, min_interval=13, approved=20,
                                 max_intervals=None, want_labels=True, out_offsets=None, total: int | None = None):
    """logits f32[n_rows, row_stride, 2]; first/length int32[n].  Returns dict with pred / smooth
    (ragged int8 at out_offsets), intervals [n, max_intervals, 2], n_intervals, keep, n_keep."""
    n = first.numel()
    dev = logits.device
    max_intervals = max_intervals or max(approved, 1)
    row_stride = logits.shape[1]
    if want_labels and out_offsets is None:
        out_offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        out_offsets[1:] = torch.cumsum(length.to(torch.int64), 0)
    if not want_labels:
        total = 0
    elif total is None:  # callers that know the host-side sum pass it: reading it back would block on the forward
        total = int(out_offsets[-1].item())
    pred = torch.empty(total, dtype=torch.int8, device=dev) if want_labels else None
    smooth = torch.empty(total, dtype=torch.int8, device=dev) if want_labels else None
    iv = torch.zeros((n, max_intervals, 2), dtype=torch.int32, device=dev)
    niv = torch.empty(n, dtype=torch.int32, device=dev)
    keep = torch.zeros((n, max_intervals + 1, 2), dtype=torch.int32, device=dev)
    nkeep = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().dc_smooth_intervals_from_logits(
        _stream(), _ptr(logits), row_stride, _ptr(first), _ptr(length), n, _ptr(out_offsets), window, min_interval,
        approved, max_intervals, _ptr(pred), _ptr(smooth), _ptr(iv), _ptr(niv), _ptr(keep), _ptr(nkeep)))
    return {"pred": pred, "smooth": smooth, "offsets": out_offsets, "intervals": iv, "n_intervals": niv,
            "keep": keep, "n_keep": nkeep}


def smooth_intervals_from_labels(labels, offsets, *, window=21, min_interval=13, approved=20,
                                 apply_approved_rule=True, max_intervals=None, want_smooth=True):
    n = offsets.numel() - 1
    dev = labels.device
    max_intervals = max_intervals or max(approved, 1)
    smooth = torch.empty_like(labels) if want_smooth else None
    iv = torch.zeros((n, max_intervals, 2), dtype=torch.int32, device=dev)
    niv = torch.empty(n, dtype=torch.int32, device=dev)
    keep = torch.zeros((n, max_intervals + 1, 2), dtype=torch.int32, device=dev)
    nkeep = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().dc_smooth_intervals_from_labels(
        _stream(), _ptr(labels), _ptr(offsets), n, window, min_interval, approved, int(apply_approved_rule),
        max_intervals, _ptr(smooth), _ptr(iv), _ptr(niv), _ptr(keep), _ptr(nkeep)))
    return {"smooth": smooth, "intervals": iv, "n_intervals": niv, "keep": keep, "n_keep": nkeep}


def summary_predict_dense(pred: torch.Tensor, labels: torch.Tensor, ignore_label: int = -100):
    """pred/labels int8[n_rows, row_stride] -> (compacted pred, compacted labels, offsets)."""
    n_rows, row_stride = pred.shape
    dev = pred.device
    counts = torch.empty(n_rows, dtype=torch.int32, device=dev)
    L = _lib.lib()
    _lib.check(L.dc_summary_predict(_stream(), _ptr(pred), _ptr(labels), n_rows, row_stride, ignore_label, None, None,
                                    None, _ptr(counts)))
    offs = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    offs[1:] = torch.cumsum(counts.to(torch.int64), 0)
    total = int(offs[-1].item())
    op = torch.empty(total, dtype=torch.int8, device=dev)
    ol = torch.empty(total, dtype=torch.int8, device=dev)
    _lib.check(L.dc_summary_predict(_stream(), _ptr(pred), _ptr(labels), n_rows, row_stride, ignore_label, _ptr(offs),
                                    _ptr(op), _ptr(ol), _ptr(counts)))
    return op, ol, offs


# ------------------------------------------------------------------------------------------
# model handle
# ------------------------------------------------------------------------------------------
GEMM_MODES = ("f32", "bf16x6", "fp16x3")


class HipModel:
    """Owns a `dc_model*` (weights repacked on the device, filters, workspace)."""

    def __init__(self, state_dict: dict[str, np.ndarray], cfg):
        from .weights import BB, HD, required_keys

        _lib.require_gpu()
        missing = [k for k in required_keys(cfg) if k not in state_dict]
        if missing:
            raise KeyError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
        self.cfg = cfg
        keep = []  # keep numpy buffers alive during dc_model_create

        def fp(name, shape=None):
            a = np.ascontiguousarray(np.asarray(state_dict[name], dtype=np.float32))
            if shape is not None and tuple(a.reshape(-1).shape) != (int(np.prod(shape)),):
                raise ValueError(f"{name}: expected {shape}, got {a.shape}")
            keep.append(a)
            return a.ctypes.data_as(_lib.f32p)

        D, DI, H, fo, emb = cfg.d_model, cfg.d_inner, cfg.head_hidden, cfg.filter_order, cfg.emb_dim
        if cfg.num_inner_mlps != 2:
            raise ValueError("the HIP filter kernel is specialised for num_inner_mlps == 2")
        layers = (_lib.LayerWeights * cfg.n_layer)()
        for i in range(cfg.n_layer):
            p = BB + f"layers.{i}."
            m = p + "mixer."
            f = m + "filter_fn."
            lw = layers[i]
            lw.norm1_w, lw.norm1_b = fp(p + "norm1.weight", (D,)), fp(p + "norm1.bias", (D,))
            lw.norm2_w, lw.norm2_b = fp(p + "norm2.weight", (D,)), fp(p + "norm2.bias", (D,))
            lw.in_proj_w, lw.in_proj_b = fp(m + "in_proj.weight", (3 * D, D)), fp(m + "in_proj.bias", (3 * D,))
            lw.out_proj_w, lw.out_proj_b = fp(m + "out_proj.weight", (D, D)), fp(m + "out_proj.bias", (D,))
            lw.short_w, lw.short_b = fp(m + "short_filter.weight", (3 * D, 3)), fp(m + "short_filter.bias", (3 * D,))
            lw.filt_bias = fp(f + "bias", (D,))
            lw.pos_z, lw.pos_t = fp(f + "pos_emb.z", (cfg.max_seq_len, emb)), fp(f + "pos_emb.t", (cfg.max_seq_len,))
            lw.filt_w0, lw.filt_b0 = fp(f + "implicit_filter.0.weight", (fo, emb)), fp(f + "implicit_filter.0.bias", (fo,))
            lw.filt_w1, lw.filt_b1 = fp(f + "implicit_filter.2.weight", (fo, fo)), fp(f + "implicit_filter.2.bias", (fo,))
            lw.filt_w2, lw.filt_b2 = fp(f + "implicit_filter.4.weight", (fo, fo)), fp(f + "implicit_filter.4.bias", (fo,))
            lw.filt_w3 = fp(f + "implicit_filter.6.weight", (D, fo))
            lw.filt_freq0 = fp(f + "implicit_filter.1.freq", (fo,))
            lw.filt_freq1 = fp(f + "implicit_filter.3.freq", (fo,))
            lw.filt_freq2 = fp(f + "implicit_filter.5.freq", (fo,))
            lw.deltas = fp(f + "modulation.deltas", (D,))
            lw.fc1_w, lw.fc1_b = fp(p + "mlp.fc1.weight", (DI, D)), fp(p + "mlp.fc1.bias", (DI,))
            lw.fc2_w, lw.fc2_b = fp(p + "mlp.fc2.weight", (D, DI)), fp(p + "mlp.fc2.bias", (D,))
        w = _lib.ModelWeights()
        w.embedding = fp(BB + "embeddings.word_embeddings.weight", (cfg.padded_vocab, D))
        w.layers = layers
        w.ln_f_w, w.ln_f_b = fp(BB + "ln_f.weight", (D,)), fp(BB + "ln_f.bias", (D,))
        w.head_w1, w.head_b1 = fp(HD + "linear1.weight", (H, D)), fp(HD + "linear1.bias", (H,))
        w.head_w2, w.head_b2 = fp(HD + "linear2.weight", (H, H)), fp(HD + "linear2.bias", (H,))
        w.head_w3, w.head_b3 = fp(HD + "linear3.weight", (cfg.num_class, H)), fp(HD + "linear3.bias", (cfg.num_class,))
        c = _lib.ModelConfig(
            d_model=D, d_inner=DI, n_layer=cfg.n_layer, vocab=cfg.padded_vocab, filter_order=fo, emb_dim=emb,
            short_filter_order=cfg.short_filter_order, max_seq_len=cfg.max_seq_len, head_hidden=H,
            num_class=cfg.num_class, gelu_tanh=1 if cfg.gelu_approximate == "tanh" else 0,
            ln_eps=cfg.layer_norm_epsilon, filter_shift=cfg.filter_shift)
        handle = C.c_void_p()
        _lib.check(_lib.lib().dc_model_create(C.byref(c), C.byref(w), C.byref(handle)))
        self._h = handle
        mode = C.c_int32()
        _lib.check(_lib.lib().dc_model_get_gemm_mode(self._h, C.byref(mode)))  # $DC_GEMM_MODE is parsed by the library only
        self.gemm_mode = GEMM_MODES[mode.value]
        self.prefix_mode = "linear" if os.environ.get("DC_PREFIX_MODE") in ("linear", "2") else "exact"  # (the library reads the same variable)
        del keep

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().dc_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, input_ids: torch.Tensor, input_quals: torch.Tensor, out: torch.Tensor | None = None):
        """Dense batch [B, L] -> logits f32[B, L, 2]."""
        assert input_ids.dim() == 2 and input_ids.shape == input_quals.shape
        assert input_ids.dtype in (torch.int64, torch.uint8) and input_quals.dtype == torch.float32
        B, L = input_ids.shape
        if out is None:
            out = torch.empty((B, L, self.cfg.num_class), dtype=torch.float32, device=input_ids.device)
        _lib.check(_lib.lib().dc_forward(self._h, _stream(), _ptr(input_ids), int(input_ids.dtype == torch.int64),
                                         _ptr(input_quals), B, L, _ptr(out)))
        return out

    def forward_ragged(self, input_ids, input_quals, first: np.ndarray, n_tokens: np.ndarray, out=None):
        """Only tokens [first[r], first[r]+n_tokens[r]) of each row go through the model."""
        n, row_stride = input_ids.shape
        first = np.ascontiguousarray(first, dtype=np.int32)
        n_tokens = np.ascontiguousarray(n_tokens, dtype=np.int32)
        if out is None:
            out = torch.zeros((n, row_stride, self.cfg.num_class), dtype=torch.float32, device=input_ids.device)
        _lib.check(_lib.lib().dc_forward_ragged(
            self._h, _stream(), _ptr(input_ids), int(input_ids.dtype == torch.int64), _ptr(input_quals), n, row_stride,
            first.ctypes.data_as(C.POINTER(C.c_int32)), n_tokens.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(out)))
        return out

    def forward_shared_prefix(self, input_ids, input_quals, first: np.ndarray, n_tokens: np.ndarray, n_pad: np.ndarray,
                              pad_token_id: int = 4, out=None):
        """forward_ragged(first, n_tokens) for rows that begin with n_pad[r] [PAD] tokens (quality 0), as the reference's
        collator pads a batch (tokenizer.py:71-87): the pad runs of all rows go through the model once per launch
        (`dc_forward_shared_prefix`).  Same logits, pad columns included."""
        n, row_stride = input_ids.shape
        first = np.ascontiguousarray(first, dtype=np.int32)
        n_tokens = np.ascontiguousarray(n_tokens, dtype=np.int32)
        n_pad = np.ascontiguousarray(n_pad, dtype=np.int32)
        assert first.shape == n_tokens.shape == n_pad.shape == (n,)
        if out is None:
            out = torch.zeros((n, row_stride, self.cfg.num_class), dtype=torch.float32, device=input_ids.device)
        i32p = C.POINTER(C.c_int32)
        _lib.check(_lib.lib().dc_forward_shared_prefix(
            self._h, _stream(), _ptr(input_ids), int(input_ids.dtype == torch.int64), _ptr(input_quals), n, row_stride,
            first.ctypes.data_as(i32p), n_tokens.ctypes.data_as(i32p), n_pad.ctypes.data_as(i32p), int(pad_token_id), _ptr(out)))
        return out

    def set_prefix_mode(self, mode: str) -> None:
        """How `forward_shared_prefix` shares the pad runs: "exact" (the long convolution still walks every row's padded
        sequence; bit-identical to the padded launch on the block-Toeplitz forms) or "linear" (y = (k * (z - z_pad) + F) * gate:
        the long convolution covers a row's own tokens only; equal to fp32 rounding)."""
        _lib.check(_lib.lib().dc_model_set_prefix_mode(self._h, {"exact": 1, "linear": 2}[mode]))
        self.prefix_mode = mode

    def prefix_violations(self) -> int:
        """Declared pad positions of `forward_shared_prefix` launches that did not hold ([PAD], 0.0) (synchronises)."""
        v = C.c_int32()
        _lib.check(_lib.lib().dc_model_prefix_violations(self._h, _stream(), C.byref(v)))
        return int(v.value)

    def head_forward(self, hidden: torch.Tensor, input_quals: torch.Tensor) -> torch.Tensor:
        """Diagnostic: final LayerNorm + head on a residual stream f32[B, L, d_model] -> logits f32[B, L, 2]."""
        assert hidden.dim() == 3 and hidden.shape[:2] == input_quals.shape and hidden.shape[2] == self.cfg.d_model
        assert hidden.dtype == torch.float32 and input_quals.dtype == torch.float32
        B, L, _ = hidden.shape
        out = torch.empty((B, L, self.cfg.num_class), dtype=torch.float32, device=hidden.device)
        _lib.check(_lib.lib().dc_head_forward(self._h, _stream(), _ptr(hidden), _ptr(input_quals), B, L, _ptr(out)))
        return out

    def forward_checked(self, input_ids, input_quals, first: np.ndarray | None = None, n_tokens: np.ndarray | None = None):
        """forward / forward_ragged with the fp16x3 range guard consulted: if an activation left the fp16 range
        the batch is re-run with the bf16 split (fp32 exponent range).  Synchronises the stream in fp16x3 mode;
        pipelined callers attribute the guard per batch themselves (`set_overflow_target`)."""
        run = (lambda: self.forward(input_ids, input_quals)) if first is None else (
            lambda: self.forward_ragged(input_ids, input_quals, first, n_tokens))
        if self.gemm_mode != "fp16x3":
            return run()
        counter = torch.zeros(1, dtype=torch.int32, device=input_ids.device)
        self.set_overflow_target(counter)
        try:
            out = run()
        finally:
            self.set_overflow_target(None)
        if int(counter.item()) > 0:
            self.set_gemm_mode("bf16x6")
            try:
                out = run()
            finally:
                self.set_gemm_mode("fp16x3")
        return out

    def set_overflow_target(self, counter: torch.Tensor | None):
        """Route the fp16x3 range guard of the forwards launched from now on into `counter` (int32[1] on the
        device, zeroed by the caller); None restores the handle's own counter."""
        if counter is not None:
            assert counter.is_cuda and counter.dtype == torch.int32 and counter.numel() >= 1
        _lib.check(_lib.lib().dc_model_set_overflow_target(self._h, _ptr(counter)))

    def filter(self, layer: int, seq_len: int) -> torch.Tensor:
        out = torch.empty((self.cfg.d_model, seq_len), dtype=torch.float32, device=_dev())
        _lib.check(_lib.lib().dc_model_filter(self._h, _stream(), layer, seq_len, _ptr(out)))
        return out

    def device_bytes(self) -> int:
        return int(_lib.lib().dc_model_device_bytes(self._h))

    def set_gemm_mode(self, mode: str):
        """"f32": fp32-input MFMA; "bf16x6": error-free 3-way bf16 split on bf16 MFMA (fp32-equivalent)."""
        _lib.check(_lib.lib().dc_model_set_gemm_mode(self._h, {"f32": 0, "bf16x6": 1, "fp16x3": 2}[mode]))
        self.gemm_mode = mode

    def set_fft_threshold(self, min_tokens: int):
        """Reads with more than `min_tokens` tokens take the O(L log L) FFT long convolution (default 2560)."""
        _lib.check(_lib.lib().dc_model_set_fft_threshold(self._h, int(min_tokens)))

    def set_fft_split(self, on: bool):
        """True (default): split / mixed-radix FFT size classes (3 passes, 3 x and 5 x 2^k points); False: the power-of-two
        transforms of rounds 2-4 (the A/B reference)."""
        _lib.check(_lib.lib().dc_model_set_fft_split(self._h, 1 if on else 0))

    def set_mlp_fusion(self, level):
        """fp16x3.  0 / False: fc1 and fc2 as two GEMM launches; 1: fc1 -> GELU -> fc2 as one kernel;
        2 / True (default): that kernel also runs the GEMM that consumes LayerNorm(its output) -- the next
        block's in_proj or the head's linear1."""
        lvl = 2 if level is True else int(level)
        _lib.check(_lib.lib().dc_model_set_mlp_fusion(self._h, lvl))

    def overflow(self) -> int:
        """fp16x3 range guard: workgroups that met an activation outside the fp16 range since the last call
        (synchronises the stream, clears the counter)."""
        n = C.c_int32()
        _lib.check(_lib.lib().dc_model_overflow(self._h, _stream(), C.byref(n)))
        return int(n.value)

    def set_profiling(self, on: bool):
        _lib.check(_lib.lib().dc_model_set_profiling(self._h, int(on)))

    def profile_reset(self):
        _lib.check(_lib.lib().dc_model_profile_reset(self._h))

    def profile(self) -> dict[str, tuple[float, int]]:
        """{stage: (total_ms, launches)} -- call after synchronising the stream."""
        buf = C.create_string_buffer(4096)
        _lib.check(_lib.lib().dc_model_profile_stages(self._h, buf, 4096))
        out = {}
        for name in filter(None, buf.value.decode().split(",")):
            ms, cnt = C.c_double(), C.c_int64()
            _lib.check(_lib.lib().dc_model_profile_get(self._h, name.encode(), C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out
