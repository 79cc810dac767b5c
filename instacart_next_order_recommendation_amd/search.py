"""Host side of the device index: cos_sim + top-k over a catalog resident in HBM.

Mirrors what the reference does with `self.product_embeddings` in
Recommender.recommend (src/inference/serve_recommendations.py:213-225): the
catalog matrix is uploaded once, L2-normalised once (the reference re-normalises
it inside cos_sim on every call), and every query batch is scored and ranked on
the GPU by libicrec's fused fp32-MFMA score+select kernel.

torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

import numpy as np
import torch

from . import _native


def _stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def exclusion_csr(exclude: Optional[Sequence[Iterable[int]]], n_queries: int, device: torch.device):
    """Per-query iterables of LOCAL row numbers -> (idx int32[nnz], off int32[Q+1]) on `device`
    (sorted, unique per query), or (None, None) when nothing is excluded."""
    if exclude is None:
        return None, None
    if len(exclude) != n_queries:
        raise ValueError(f"exclude has {len(exclude)} entries for {n_queries} queries")
    off = np.zeros(n_queries + 1, np.int32)
    flat: list[int] = []
    for i, e in enumerate(exclude):
        flat.extend(sorted(set(int(v) for v in e)))
        off[i + 1] = len(flat)
    if not flat:
        return None, None
    idx = torch.from_numpy(np.asarray(flat, np.int32)).to(device)
    return idx, torch.from_numpy(off).to(device)


ROW_STORAGE = {"f32": 0, "bf16": 1, "f32+filter": 2, "bf16+filter": 3}  # ICREC_ROWS_* in include/icrec.h


class DeviceIndex:
    """A row shard of the product-embedding matrix, normalised and resident on one GPU.

    storage="bf16" keeps the normalised rows as bfloat16 (half the HBM; BASELINE config 5's 10M-row
    catalog); the arithmetic stays the exact fp32 fmaf chain over the widened values.
    storage="f32+filter" keeps fp32 rows plus their f16 hi/lo planes (2x the HBM): batches of >= 256 queries are
    ranked on the f16 matrix cores first and the candidates re-scored exactly — same bits out, several times
    faster on large catalogs (see ICREC_ROWS_F32_FILTER in include/icrec.h)."""

    def __init__(self, embeddings, device: str | torch.device = "cuda:0", row_offset: int = 0,
                 storage: str = "f32"):
        if storage not in ROW_STORAGE:
            raise ValueError(f"storage must be one of {sorted(ROW_STORAGE)}, got {storage!r}")
        self.storage = storage
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _native.IcrecError("DeviceIndex needs a CUDA/HIP device; there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        L = _native.lib()
        rows = torch.as_tensor(embeddings)
        if rows.dim() != 2:
            raise ValueError("embeddings must be [n_rows, dim]")
        rows = rows.to(device=self.device, dtype=torch.float32).contiguous()
        self.n_rows, self.dim = int(rows.shape[0]), int(rows.shape[1])
        self.row_offset = int(row_offset)
        h = C.c_void_p()
        torch.cuda.synchronize(self.device)
        _native.check(L.icrec_index_create_ex(_ptr(rows), self.n_rows, self.dim, self.row_offset, self.device.index,
                                              ROW_STORAGE[storage], C.byref(h)), "icrec_index_create_ex")
        self._h = h
        self._ws_by_stream: dict[int, torch.Tensor] = {}

    def close(self) -> None:
        if getattr(self, "_h", None):
            _native.lib().icrec_index_destroy(self._h)
            self._h = None
        self._ws_by_stream = {}

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _workspace(self, n_queries: int, k: int) -> torch.Tensor:
        """Scratch block for one search, ONE PER STREAM: searches issued on different streams (pipeline.py runs batch
        i's search on a side stream while the caller may use the index from its own stream) never share scratch memory,
        and a block is only ever allocated, used and dropped on the stream it belongs to, so the caching allocator's
        stream-ordered reuse is safe when it grows."""
        need = int(_native.lib().icrec_search_workspace_bytes(self._h, n_queries, k))
        if need == 0:
            raise _native.IcrecError(f"bad search shape: n_queries={n_queries}, k={k}")
        key = int(torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._ws_by_stream.get(key)
        if ws is None or ws.numel() < need:
            self._ws_by_stream.pop(key, None)
            ws = self._ws_by_stream[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def _queries(self, q) -> torch.Tensor:
        q = torch.as_tensor(q)
        if q.dim() == 1:
            q = q.unsqueeze(0)
        if q.dim() != 2 or q.shape[1] != self.dim:
            raise ValueError(f"queries must be [Q, {self.dim}], got {tuple(q.shape)}")
        return q.to(device=self.device, dtype=torch.float32).contiguous()

    # ------------------------------------------------------------------ API
    def search(self, q, k: int, exclude: Optional[Sequence[Iterable[int]]] = None):
        """Top-k rows per query: (idx int64[Q,k] with -1 pads, score float32[Q,k]).
        Order: score descending, lower row first on ties.  `exclude`: per-query local rows."""
        q = self._queries(q)
        Q = int(q.shape[0])
        ei, eo = exclusion_csr(exclude, Q, self.device)
        idx = torch.empty((Q, k), dtype=torch.int64, device=self.device)
        sc = torch.empty((Q, k), dtype=torch.float32, device=self.device)
        ws = self._workspace(Q, k)
        _native.check(_native.lib().icrec_search(self._h, _ptr(q), Q, k, _ptr(ei), _ptr(eo), _ptr(idx), _ptr(sc),
                                                 _ptr(ws), ws.numel(), _stream_ptr(self.device)), "icrec_search")
        return idx, sc

    def search_into(self, q: torch.Tensor, k: int, excl_idx: Optional[torch.Tensor], excl_off: Optional[torch.Tensor],
                    out_idx: torch.Tensor, out_score: torch.Tensor) -> None:
        """Allocation-free form of `search` on caller-owned device buffers (hipGraph-capturable once the
        workspace for this (Q, k) exists): q float32 [Q, dim], out_idx int64 [Q, k], out_score float32 [Q, k]."""
        Q = int(q.shape[0])
        ws = self._workspace(Q, k)
        _native.check(_native.lib().icrec_search(self._h, _ptr(q), Q, k, _ptr(excl_idx), _ptr(excl_off), _ptr(out_idx),
                                                 _ptr(out_score), _ptr(ws), ws.numel(), _stream_ptr(self.device)),
                      "icrec_search")

    def search_partial(self, q, k: int, exclude: Optional[Sequence[Iterable[int]]] = None) -> torch.Tensor:
        """Shard-local sorted lists as packed keys, int64-viewed uint64 [Q,k] (see icrec_search_partial)."""
        q = self._queries(q)
        Q = int(q.shape[0])
        ei, eo = exclusion_csr(exclude, Q, self.device)
        keys = torch.empty((Q, k), dtype=torch.int64, device=self.device)
        ws = self._workspace(Q, k)
        _native.check(_native.lib().icrec_search_partial(self._h, _ptr(q), Q, k, _ptr(ei), _ptr(eo), _ptr(keys),
                                                         _ptr(ws), ws.numel(), _stream_ptr(self.device)),
                      "icrec_search_partial")
        return keys

    def scores(self, q) -> torch.Tensor:
        """Full cosine score matrix [Q, n_rows] (parity checks only; never on the serving path)."""
        q = self._queries(q)
        Q = int(q.shape[0])
        out = torch.empty((Q, self.n_rows), dtype=torch.float32, device=self.device)
        ws = self._workspace(Q, 1)
        _native.check(_native.lib().icrec_scores(self._h, _ptr(q), Q, _ptr(out), _ptr(ws), ws.numel(),
                                                 _stream_ptr(self.device)), "icrec_scores")
        return out

    def rank_all(self, q) -> torch.Tensor:
        """The complete ranking of the shard for each query: int64 [Q, n_rows] global rows, best first
        (score descending, lower row first on ties) — `scores.argsort(descending=True)` of the reference's
        evaluation consumers.  Workspace grows with Q * n_rows: call it in passes of a few hundred queries."""
        q = self._queries(q)
        Q = int(q.shape[0])
        L = _native.lib()
        need = int(L.icrec_rank_all_workspace_bytes(self._h, Q))
        if need == 0:
            raise _native.IcrecError(f"bad rank_all shape: n_queries={Q}")
        ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((Q, self.n_rows), dtype=torch.int64, device=self.device)
        _native.check(L.icrec_rank_all(self._h, _ptr(q), Q, _ptr(out), _ptr(ws), ws.numel(), _stream_ptr(self.device)),
                      "icrec_rank_all")
        return out

    def export(self) -> torch.Tensor:
        """The normalised rows the index holds, [n_rows, dim] fp32 on the device."""
        out = torch.empty((self.n_rows, self.dim), dtype=torch.float32, device=self.device)
        _native.check(_native.lib().icrec_index_export(self._h, _ptr(out), _stream_ptr(self.device)),
                      "icrec_index_export")
        return out


def merge_topk(keys: torch.Tensor, k: int):
    """Merge sorted partial key lists [n_lists, Q, k] (int64-viewed uint64) into (idx, score) [Q,k]."""
    if keys.dim() != 3 or keys.shape[2] != k:
        raise ValueError("keys must be [n_lists, Q, k]")
    keys = keys.contiguous()
    n_lists, Q = int(keys.shape[0]), int(keys.shape[1])
    idx = torch.empty((Q, k), dtype=torch.int64, device=keys.device)
    sc = torch.empty((Q, k), dtype=torch.float32, device=keys.device)
    _native.check(_native.lib().icrec_merge_topk(_ptr(keys), n_lists, Q, k, _ptr(idx), _ptr(sc), keys.device.index,
                                                 _stream_ptr(keys.device)), "icrec_merge_topk")
    return idx, sc


def normalize_rows(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """x / max(|x|_2, eps) row-wise on the device (torch.nn.functional.normalize as cos_sim uses it)."""
    x = x.to(dtype=torch.float32).contiguous()
    out = torch.empty_like(x)
    _native.check(_native.lib().icrec_normalize_rows(_ptr(x), _ptr(out), x.shape[0], x.shape[1], eps,
                                                     x.device.index, _stream_ptr(x.device)), "icrec_normalize_rows")
    return out
