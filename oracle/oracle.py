"""ctypes wrapper over oracle/_build/libicrec_oracle.so (TEST INFRASTRUCTURE ONLY).

Mirrors the entry points of oracle/icrec_oracle.c with numpy in / numpy out.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libicrec_oracle.so"


class BertCfg(C.Structure):
    """Same field order as icrec_bert_cfg (include/icrec.h)."""

    _fields_ = [
        ("vocab_size", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32),
        ("heads", C.c_int32), ("intermediate", C.c_int32), ("max_position", C.c_int32),
        ("type_vocab", C.c_int32), ("ln_eps", C.c_float), ("n_normalize", C.c_int32),
        ("gemm_mode", C.c_int32),  # engine option of the HIP library; the oracle ignores it
    ]


def build() -> Path:
    """Compile the oracle with gcc if the .so is missing or stale."""
    src = _HERE / "icrec_oracle.c"
    if not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_SO))
        L.icrec_oracle_weight_count.restype = C.c_size_t
        L.icrec_oracle_weight_count.argtypes = [C.POINTER(BertCfg)]
        L.icrec_oracle_encode.restype = C.c_int
        L.icrec_oracle_encode.argtypes = [C.c_void_p, C.POINTER(BertCfg), C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_void_p]
        L.icrec_oracle_normalize_rows.restype = None
        L.icrec_oracle_normalize_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float]
        L.icrec_oracle_scores.restype = None
        L.icrec_oracle_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p]
        L.icrec_oracle_rank.restype = None
        L.icrec_oracle_rank.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int64,
                                        C.c_void_p, C.c_void_p]
        L.icrec_oracle_search.restype = None
        L.icrec_oracle_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.icrec_oracle_search_bf16.restype = None
        L.icrec_oracle_search_bf16.argtypes = L.icrec_oracle_search.argtypes
        L.icrec_oracle_round_bf16.restype = None
        L.icrec_oracle_round_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.icrec_oracle_merge.restype = None
        L.icrec_oracle_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p]
        L.icrec_oracle_threads.restype = C.c_int
        L.icrec_oracle_set_threads.restype = None
        L.icrec_oracle_set_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def make_cfg(vocab_size=30522, hidden=384, layers=6, heads=12, intermediate=1536,
             max_position=512, type_vocab=2, ln_eps=1e-12, n_normalize=2) -> BertCfg:
    return BertCfg(vocab_size, hidden, layers, heads, intermediate, max_position, type_vocab,
                   ln_eps, n_normalize, 0)


def weight_count(cfg: BertCfg) -> int:
    return int(lib().icrec_oracle_weight_count(C.byref(cfg)))


def encode(weights: np.ndarray, cfg: BertCfg, ids: np.ndarray, cu_seqlens: np.ndarray,
           return_hidden: bool = False):
    """Token-packed encode -> [n_seqs, hidden] (and the last hidden states if asked)."""
    w = _f32(weights)
    assert w.size == weight_count(cfg), (w.size, weight_count(cfg))
    ids = _i32(ids)
    cu = _i32(cu_seqlens)
    n = cu.size - 1
    out = np.empty((n, cfg.hidden), np.float32)
    hid = np.empty((int(cu[-1]), cfg.hidden), np.float32) if return_hidden else None
    rc = lib().icrec_oracle_encode(_p(w), C.byref(cfg), _p(ids), _p(cu), n, _p(out), _p(hid))
    if rc != 0:
        raise ValueError("icrec_oracle_encode: bad arguments")
    return (out, hid) if return_hidden else out


def normalize_rows(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    x = _f32(x)
    out = np.empty_like(x)
    lib().icrec_oracle_normalize_rows(_p(x), _p(out), x.shape[0], x.shape[1], eps)
    return out


def scores(qhat: np.ndarray, phat: np.ndarray) -> np.ndarray:
    qhat, phat = _f32(qhat), _f32(phat)
    out = np.empty((qhat.shape[0], phat.shape[0]), np.float32)
    lib().icrec_oracle_scores(_p(qhat), _p(phat), qhat.shape[0], phat.shape[0], qhat.shape[1], _p(out))
    return out


def _csr(excl, n_queries):
    if excl is None:
        return None, None
    off = np.zeros(n_queries + 1, np.int32)
    flat = []
    for i, e in enumerate(excl):
        e = sorted(set(int(v) for v in e))
        flat.extend(e)
        off[i + 1] = len(flat)
    return np.asarray(flat, np.int32), off


def round_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 -> bfloat16 (round to nearest even) -> fp32."""
    x = _f32(x)
    out = np.empty_like(x)
    lib().icrec_oracle_round_bf16(_p(x), _p(out), x.size)
    return out


def search(q: np.ndarray, P: np.ndarray, k: int, excl=None, row_offset: int = 0, storage: str = "f32"):
    """cos_sim + argsort + exclusion loop.  excl: per-query iterables of LOCAL row numbers.
    storage="bf16": the normalised catalog rows are rounded to bfloat16 first (ICREC_ROWS_BF16)."""
    q, P = _f32(q), _f32(P)
    Q = q.shape[0]
    ei, eo = _csr(excl, Q)
    idx = np.empty((Q, k), np.int64)
    sc = np.empty((Q, k), np.float32)
    fn = {"f32": lib().icrec_oracle_search, "bf16": lib().icrec_oracle_search_bf16}[storage]
    fn(_p(q), _p(P), Q, P.shape[0], q.shape[1], k, _p(ei), _p(eo), row_offset,
                              _p(idx), _p(sc))
    return idx, sc


def merge(idx: np.ndarray, score: np.ndarray):
    """[n_lists, Q, k] partial lists -> global top-k [Q, k]."""
    idx = np.ascontiguousarray(idx, np.int64)
    score = _f32(score)
    n_lists, Q, k = idx.shape
    oi = np.empty((Q, k), np.int64)
    os_ = np.empty((Q, k), np.float32)
    lib().icrec_oracle_merge(_p(idx), _p(score), n_lists, Q, k, _p(oi), _p(os_))
    return oi, os_


def threads() -> int:
    return int(lib().icrec_oracle_threads())


def set_threads(n: int) -> None:
    lib().icrec_oracle_set_threads(int(n))


def usable_cpus() -> int:
    """CPUs this process may actually use: min(affinity mask, cgroup v2/v1 CPU quota)."""
    import math
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, math.ceil(q / p)))
        except (OSError, ValueError):
            pass
    return n
