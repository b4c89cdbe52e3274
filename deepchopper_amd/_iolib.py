"""ctypes binding of libdeepchopper_io.so (include/deepchopper_io.h): host-side FASTQ / gzip / BGZF
streaming I/O.  Host code only -- it loads and runs without a GPU."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DC_IO_LIB") or os.path.join(_HERE, "lib", "libdeepchopper_io.so")
# the C-ABI header: in the repo next to the package (looked at first: a copy a local build left inside the package may be
# stale), shipped inside an installed package (setup.py copies include/*.h there)
HEADER_PATH = next((p for p in (os.path.join(os.path.dirname(_HERE), "include", "deepchopper_io.h"), os.path.join(_HERE, "include", "deepchopper_io.h"))
                    if os.path.exists(p)), os.path.join(os.path.dirname(_HERE), "include", "deepchopper_io.h"))

i64, vp = C.c_int64, C.c_void_p

COMPRESSED_TYPES = ("Uncompress", "Gzip", "Bgzip", "Zip", "Bzip2", "Xz", "Zstd", "Unknown")  # writefq.rs:51-60
FQ_OK, FQ_ERR_IO, FQ_ERR_FORMAT, FQ_ERR_INCOMPLETE, FQ_ERR_LENGTH, FQ_ERR_ARG = range(6)

_SIGNATURES = {
    "dc_io_last_error": (C.c_char_p, []),
    "dc_io_check_compressed_type": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "dc_fq_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp)]),
    "dc_fq_open_range": (C.c_int, [C.c_char_p, i64, i64, C.c_int, C.POINTER(vp)]),
    "dc_fq_read_chunk": (C.c_int, [vp, i64, vp, vp, i64, vp, vp, i64, vp, C.POINTER(i64)]),
    "dc_fq_read_chunk_full": (C.c_int, [vp, i64, vp, vp, i64, vp, vp, i64, vp, vp, C.POINTER(i64)]),
    "dc_fq_records_read": (i64, [vp]),
    "dc_fq_set_strict": (C.c_int, [vp, C.c_int]),
    "dc_fq_helper_cpu_seconds": (C.c_double, [vp]),
    "dc_fq_bytes_consumed": (i64, [vp]),
    "dc_fq_close": (None, [vp]),
    "dc_bgzf_open": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]),
    "dc_bgzf_write": (C.c_int, [vp, vp, i64]),
    "dc_bgzf_write_fastq": (C.c_int, [vp, i64, vp, vp, vp, vp, vp]),
    "dc_bgzf_close": (C.c_int, [vp]),
    "dc_bgzf_mark": (C.c_int, [vp]),
    "dc_bgzf_mark_count": (C.c_int64, [vp]),
    "dc_bgzf_close2": (C.c_int, [vp, vp, i64, vp, vp, vp]),
    "dc_deflate_block": (C.c_int, [vp, i64, vp, i64, C.POINTER(i64)]),
    "dc_pt_open": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "dc_pt_submit": (C.c_int, [vp, C.c_char_p, C.c_int32, i64, C.c_int32, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp, vp, vp, C.POINTER(i64)]),
    "dc_pt_submit_packed": (C.c_int, [vp, C.c_char_p, C.c_int32, i64, C.c_int32, vp, vp, vp, vp, vp, i64, i64, vp, vp, vp, C.POINTER(i64)]),
    "dc_pt_wait": (C.c_int, [vp, i64]),
    "dc_pt_close": (C.c_int, [vp]),
    "dc_pt_load": (C.c_int, [vp, i64, C.c_int32, i64, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "dc_pt_export": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "dc_pt_free": (None, [vp]),
    "dc_chop_write_chunk": (C.c_int, [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, C.c_int32, vp,
                                      C.POINTER(i64)]),
}


class ChopParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("min_read_len", "max_process_intervals", "min_read_length_after_chop",
                                         "chop_type", "output_chopped_seqs", "id_annotation")]

_lib = None


def header_symbols() -> list[str]:
    with open(HEADER_PATH) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(dc_[a-z0-9_]+)\s*\(", text)))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and "DC_IO_LIB" not in os.environ:
            import subprocess  # host-only g++ build, a few seconds

            subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "../lib/libdeepchopper_io.so"], capture_output=True)
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: build it with `make -C deepchopper_amd/csrc` "
                          "(or `python -c 'import __graft_entry__ as g; g.build()'`)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def last_error() -> str:
    msg = lib().dc_io_last_error()
    return msg.decode(errors="replace") if msg else ""


def raise_for(status: int) -> None:
    """Maps reader status codes to the exception types of parse_fastq_file (only_fq.py:21-85):
    incomplete / length-mismatch records -> ValueError, parse and I/O failures -> RuntimeError."""
    if status == 0:
        return
    msg = last_error() or f"libdeepchopper_io error {status}"
    if status in (FQ_ERR_INCOMPLETE, FQ_ERR_LENGTH, FQ_ERR_ARG):
        raise ValueError(msg)
    raise RuntimeError(msg)
