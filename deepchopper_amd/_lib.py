"""ctypes binding of libdeepchopper_hip.so (the C ABI declared in include/deepchopper_hip.h).

The HIP library is the product: there is no CPU fallback anywhere in this package.  Loading
fails loudly if the shared object is missing, and every compute entry point fails loudly
(non-zero status -> `DeepChopperHipError`) when no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DC_HIP_LIB") or os.path.join(_HERE, "lib", "libdeepchopper_hip.so")  # override: kernel A/B experiments
# the C-ABI header: in the repo next to the package (looked at first: a copy a local build left inside the package may be
# stale), shipped inside an installed package (setup.py copies include/*.h there)
HEADER_PATH = next((p for p in (os.path.join(os.path.dirname(_HERE), "include", "deepchopper_hip.h"), os.path.join(_HERE, "include", "deepchopper_hip.h"))
                    if os.path.exists(p)), os.path.join(os.path.dirname(_HERE), "include", "deepchopper_hip.h"))
CSRC_DIR = os.path.join(_HERE, "csrc")


class DeepChopperHipError(Exception):
    """Raised with `dc_last_error()` when a C-ABI call returns non-zero (the PyO3 layer of the
    reference raises plain `Exception` with the Rust error text, src/error.rs:28-53)."""


def build(force: bool = False, jobs: int = 6) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], check=True, capture_output=True)
    proc = subprocess.run(["make", "-C", CSRC_DIR, f"-j{jobs}"], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("building libdeepchopper_hip.so failed:\n" + proc.stdout + proc.stderr)
    return LIB_PATH


i32, i64, f32p = C.c_int32, C.c_int64, C.POINTER(C.c_float)
vp = C.c_void_p


class ModelConfig(C.Structure):
    _fields_ = [
        ("d_model", i32), ("d_inner", i32), ("n_layer", i32), ("vocab", i32), ("filter_order", i32),
        ("emb_dim", i32), ("short_filter_order", i32), ("max_seq_len", i32), ("head_hidden", i32),
        ("num_class", i32), ("gelu_tanh", i32), ("ln_eps", C.c_float), ("filter_shift", C.c_float),
    ]


_LAYER_FIELDS = [
    "norm1_w", "norm1_b", "norm2_w", "norm2_b", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b",
    "short_w", "short_b", "filt_bias", "pos_z", "pos_t", "filt_w0", "filt_b0", "filt_w1", "filt_b1",
    "filt_w2", "filt_b2", "filt_w3", "filt_freq0", "filt_freq1", "filt_freq2", "deltas",
    "fc1_w", "fc1_b", "fc2_w", "fc2_b",
]


class LayerWeights(C.Structure):
    _fields_ = [(n, f32p) for n in _LAYER_FIELDS]


class ModelWeights(C.Structure):
    _fields_ = [
        ("embedding", f32p), ("layers", C.POINTER(LayerWeights)), ("ln_f_w", f32p), ("ln_f_b", f32p),
        ("head_w1", f32p), ("head_b1", f32p), ("head_w2", f32p), ("head_b2", f32p),
        ("head_w3", f32p), ("head_b3", f32p),
    ]


_SIGNATURES = {
    "dc_last_error": (C.c_char_p, []),
    "dc_version": (C.c_int, []),
    "dc_device_count": (C.c_int, []),
    "dc_set_device": (C.c_int, [C.c_int]),
    "dc_normalize_seq": (C.c_int, [vp, vp, vp, i32, i32, vp, vp]),
    "dc_encode_qual": (C.c_int, [vp, vp, i64, i32, vp]),
    "dc_tokenize_batch": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i64, i32, i32, vp, vp, vp, vp, vp]),
    "dc_kmer_validate": (C.c_int, [vp, vp, vp, i32, i32, C.c_char_p, i32, vp]),
    "dc_kmer_encode": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, C.c_char_p, i32, i32, i32, i32, vp, vp]),
    "dc_kmer_vectorize_targets": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "dc_smooth_intervals_from_logits": (C.c_int, [vp, vp, i64, vp, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "dc_smooth_intervals_from_labels": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "dc_summary_predict": (C.c_int, [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp]),
    "dc_model_create": (C.c_int, [C.POINTER(ModelConfig), C.POINTER(ModelWeights), C.POINTER(vp)]),
    "dc_model_free": (None, [vp]),
    "dc_forward": (C.c_int, [vp, vp, vp, i32, vp, i32, i32, vp]),
    "dc_forward_ragged": (C.c_int, [vp, vp, vp, i32, vp, i32, i64, C.POINTER(i32), C.POINTER(i32), vp]),
    "dc_forward_shared_prefix": (C.c_int, [vp, vp, vp, i32, vp, i32, i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32, vp]),
    "dc_model_prefix_violations": (C.c_int, [vp, vp, C.POINTER(i32)]),
    "dc_model_set_prefix_mode": (C.c_int, [vp, i32]),
    "dc_head_forward": (C.c_int, [vp, vp, vp, vp, i32, i32, vp]),
    "dc_model_device_bytes": (i64, [vp]),
    "dc_model_filter": (C.c_int, [vp, vp, i32, i32, vp]),
    "dc_model_set_gemm_mode": (C.c_int, [vp, i32]),
    "dc_model_set_mlp_fusion": (C.c_int, [vp, i32]),
    "dc_model_set_fft_threshold": (C.c_int, [vp, i32]),
    "dc_model_set_fft_split": (C.c_int, [vp, i32]),
    "dc_pt_pack": (C.c_int, [vp, vp, vp, vp, i32, vp, i64, i64, i32, vp, vp, vp, i64]),
    "dc_model_overflow": (C.c_int, [vp, vp, C.POINTER(i32)]),
    "dc_model_set_overflow_target": (C.c_int, [vp, vp]),
    "dc_model_get_gemm_mode": (C.c_int, [vp, C.POINTER(i32)]),
    "dc_model_set_profiling": (C.c_int, [vp, i32]),
    "dc_model_profile_reset": (C.c_int, [vp]),
    "dc_model_profile_get": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]),
    "dc_model_profile_stages": (C.c_int, [vp, C.c_char_p, i32]),
}

_lib = None


def header_symbols() -> list[str]:
    """Every function the public header declares (used by the CPU test that checks the shared
    object exports the whole ABI)."""
    with open(HEADER_PATH) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dc_[a-z0-9_]+)\s*\(", text)))


def lib() -> C.CDLL:
    """Load the shared object (no GPU needed to load; compute calls need one)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeepChopperHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C deepchopper_amd/csrc`).  deepchopper_amd has no CPU fallback."
        )
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return handle


def check(status: int) -> None:
    if status != 0:
        msg = lib().dc_last_error()
        raise DeepChopperHipError(msg.decode(errors="replace") if msg else f"libdeepchopper_hip error {status}")


def device_count() -> int:
    return int(lib().dc_device_count())


def require_gpu() -> None:
    if device_count() < 1:
        raise DeepChopperHipError(
            "no HIP device visible: deepchopper_amd runs its hot path only on the GPU (no CPU fallback)"
        )
