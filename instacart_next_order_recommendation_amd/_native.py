"""ctypes binding of libicrec.so (include/icrec.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the
shared object is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libicrec.so"

ICREC_MAX_K = 128
COMM_ID_BYTES = 128
GEMM_F32, GEMM_F16X3 = 0, 1
GEMM_MODES = {"f32": GEMM_F32, "f16x3": GEMM_F16X3}

#: every symbol include/icrec.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "icrec_encoder_weight_count", "icrec_encoder_create", "icrec_encoder_destroy",
    "icrec_encode_workspace_bytes", "icrec_encode", "icrec_encode_batch_split",
    "icrec_index_create", "icrec_index_create_ex", "icrec_index_destroy", "icrec_index_rows", "icrec_index_storage",
    "icrec_index_export", "icrec_index_dim", "icrec_index_device",
    "icrec_comm_unique_id", "icrec_comm_init", "icrec_comm_destroy", "icrec_comm_rank", "icrec_comm_world",
    "icrec_search_sharded_workspace_bytes", "icrec_search_sharded",
    "icrec_search_sharded_excl_workspace_bytes", "icrec_search_sharded_excl", "icrec_index_row_offset",
    "icrec_exclusions_to_shard_csr_workspace_bytes", "icrec_exclusions_to_shard_csr",
    "icrec_search_workspace_bytes", "icrec_search", "icrec_search_partial", "icrec_merge_topk",
    "icrec_scores", "icrec_normalize_rows", "icrec_rank_all_workspace_bytes", "icrec_rank_all",
    "icrec_tokenizer_create", "icrec_tokenizer_destroy", "icrec_tokenizer_vocab_size", "icrec_tokenize",
    "icrec_last_error", "icrec_version",
    "icrec_timing_enable", "icrec_timing_reset", "icrec_timing_query",
]


class IcrecError(RuntimeError):
    """A libicrec call returned a non-zero status."""


class BertCfg(C.Structure):
    """icrec_bert_cfg (include/icrec.h)."""

    _fields_ = [
        ("vocab_size", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32),
        ("heads", C.c_int32), ("intermediate", C.c_int32), ("max_position", C.c_int32),
        ("type_vocab", C.c_int32), ("ln_eps", C.c_float), ("n_normalize", C.c_int32),
        ("gemm_mode", C.c_int32),
    ]


def build(force: bool = False) -> Path:
    """Compile libicrec.so for gfx950 with hipcc (in-tree, next to this file)."""
    csrc = _PKG / "csrc"
    if force:
        subprocess.run(["make", "-C", str(csrc), "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", str(csrc), "-j", str(min(8, os.cpu_count() or 1))],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libicrec.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load libicrec.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise IcrecError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: it brings its own libamdhip64 / librccl, and libicrec must resolve to THAT runtime - loaded before
    # torch it binds /opt/rocm's copy instead, the process then holds two HIP runtimes and the second one sees no
    # device ("no ROCm-capable device is detected" from hipSetDevice inside icrec_encoder_create)
    import torch  # noqa: F401

    L = C.CDLL(str(LIB_PATH))
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    sig = {
        "icrec_encoder_weight_count": (sz, [C.POINTER(BertCfg)]),
        "icrec_encoder_create": (C.c_int, [vp, sz, C.POINTER(BertCfg), C.c_int, C.POINTER(vp)]),
        "icrec_encoder_destroy": (C.c_int, [vp]),
        "icrec_encode_workspace_bytes": (sz, [vp, i64, i32]),
        "icrec_encode": (C.c_int, [vp, vp, vp, i32, i64, i32, vp, vp, sz, vp]),
        "icrec_encode_batch_split": (C.c_int, [vp, i64, C.POINTER(i64), C.POINTER(i64)]),
        "icrec_index_create": (C.c_int, [vp, i64, i32, i64, C.c_int, C.POINTER(vp)]),
        "icrec_index_create_ex": (C.c_int, [vp, i64, i32, i64, C.c_int, i32, C.POINTER(vp)]),
        "icrec_index_storage": (i32, [vp]),
        "icrec_index_dim": (i32, [vp]),
        "icrec_index_device": (i32, [vp]),
        "icrec_comm_unique_id": (C.c_int, [vp]),
        "icrec_comm_init": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
        "icrec_comm_destroy": (C.c_int, [vp]),
        "icrec_comm_rank": (i32, [vp]),
        "icrec_comm_world": (i32, [vp]),
        "icrec_search_sharded_workspace_bytes": (sz, [vp, vp, i32, i32]),
        "icrec_search_sharded": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
        "icrec_search_sharded_excl_workspace_bytes": (sz, [vp, vp, i32, i32, i32]),
        "icrec_search_sharded_excl": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, sz, vp]),
        "icrec_exclusions_to_shard_csr_workspace_bytes": (sz, [i32, i32]),
        "icrec_exclusions_to_shard_csr": (C.c_int, [vp, vp, i32, i32, i32, i64, i64, vp, vp, vp, sz, C.c_int, vp]),
        "icrec_index_row_offset": (i64, [vp]),
        "icrec_index_destroy": (C.c_int, [vp]),
        "icrec_index_rows": (i64, [vp]),
        "icrec_index_export": (C.c_int, [vp, vp, vp]),
        "icrec_search_workspace_bytes": (sz, [vp, i32, i32]),
        "icrec_search": (C.c_int, [vp, vp, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
        "icrec_search_partial": (C.c_int, [vp, vp, i32, i32, vp, vp, vp, vp, sz, vp]),
        "icrec_merge_topk": (C.c_int, [vp, i32, i32, i32, vp, vp, C.c_int, vp]),
        "icrec_scores": (C.c_int, [vp, vp, i32, vp, vp, sz, vp]),
        "icrec_rank_all_workspace_bytes": (sz, [vp, i32]),
        "icrec_rank_all": (C.c_int, [vp, vp, i32, vp, vp, sz, vp]),
        "icrec_normalize_rows": (C.c_int, [vp, vp, i64, i32, C.c_float, C.c_int, vp]),
        "icrec_tokenizer_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]),
        "icrec_tokenizer_destroy": (C.c_int, [vp]),
        "icrec_tokenizer_vocab_size": (i32, [vp]),
        "icrec_tokenize": (C.c_int, [vp, C.POINTER(C.c_char_p), i32, vp, i64, vp, i32]),
        "icrec_last_error": (C.c_char_p, []),
        "icrec_version": (C.c_char_p, []),
        "icrec_timing_enable": (C.c_int, [C.c_int]),
        "icrec_timing_reset": (C.c_int, []),
        "icrec_timing_query": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(i64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().icrec_last_error().decode("utf-8", "replace")
        raise IcrecError(f"{what} failed (status {rc}): {msg}")


def timing_enable(on: bool) -> None:
    lib().icrec_timing_enable(1 if on else 0)


def timing_reset() -> None:
    lib().icrec_timing_reset()


def timing_query(which: int) -> tuple[float, int]:
    """(average ms, launches) for slot `which` — see icrec_timing_query in include/icrec.h."""
    ms, n = C.c_double(0.0), C.c_int64(0)
    check(lib().icrec_timing_query(which, C.byref(ms), C.byref(n)), "icrec_timing_query")
    return ms.value, n.value
