"""Host side of the device encoder: SentenceTransformer.encode's GPU work.

The reference calls `self.model.encode(texts, batch_size=64, normalize_embeddings=True)`
(src/inference/serve_recommendations.py:195-200, :213, :246).  Here tokenisation is a
separate host stage (tokenizer.py); this module takes token ids, packs them back to back
(no padding), and runs libicrec's fp32-MFMA BERT forward + mean-pool + L2-normalise.

torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native
from .synthetic import BertShape

DEFAULT_GEMM_MODE = "f16x3"
MAX_SEQ_LEN = 256  # configs/train.yaml:11 (max_seq_length), also the attention kernel's limit


def pack_token_ids(seqs: Sequence[Sequence[int]]):
    """List of id lists -> (ids int32[T], cu_seqlens int32[n+1], max_len)."""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    if len(seqs) == 0 or (lens < 1).any():
        raise ValueError("every sequence needs at least one token")
    if lens.max() > MAX_SEQ_LEN:
        raise ValueError(f"sequence longer than max_seq_length={MAX_SEQ_LEN}; truncate on the host")
    cu = np.zeros(len(seqs) + 1, np.int32)
    np.cumsum(lens, out=cu[1:])
    ids = np.concatenate([np.asarray(s, np.int32) for s in seqs]) if len(seqs) > 1 else np.asarray(seqs[0], np.int32)
    return ids, cu, int(lens.max())


class DeviceEncoder:
    """all-MiniLM-L6-v2-shaped BERT encoder resident on one GPU."""

    SPLIT_MIN_SEQS = 128       # two-stream split only when each half has at least this many sequences ...
    SPLIT_MIN_TOKENS = 16384   # ... and this many tokens (enough blocks to fill the chip on its own)

    def __init__(self, weights: np.ndarray, shape: BertShape = BertShape(), device: str | torch.device = "cuda:0",
                 gemm_mode: Optional[str] = None):
        """gemm_mode: "f32" (exact f32 MFMA, bit-identical GEMMs) or "f16x3" (3-term split on the f16
        MFMA, fp32-level accuracy, ~4x faster); default from $ICREC_GEMM_MODE, else DEFAULT_GEMM_MODE."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _native.IcrecError("DeviceEncoder needs a CUDA/HIP device; there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.shape = shape
        L = _native.lib()
        import os

        self.gemm_mode = gemm_mode or os.getenv("ICREC_GEMM_MODE") or DEFAULT_GEMM_MODE
        if self.gemm_mode not in _native.GEMM_MODES:
            raise ValueError(f"gemm_mode must be one of {sorted(_native.GEMM_MODES)}, got {self.gemm_mode!r}")
        self._cfg = _native.BertCfg(shape.vocab_size, shape.hidden, shape.layers, shape.heads, shape.intermediate,
                                    shape.max_position, shape.type_vocab, shape.ln_eps, shape.n_normalize,
                                    _native.GEMM_MODES[self.gemm_mode])
        w = np.ascontiguousarray(weights, dtype=np.float32).reshape(-1)
        want = int(L.icrec_encoder_weight_count(C.byref(self._cfg)))
        if w.size != want:
            raise ValueError(f"weight blob has {w.size} floats, expected {want} (layout: include/icrec.h)")
        h = C.c_void_p()
        _native.check(L.icrec_encoder_create(w.ctypes.data_as(C.c_void_p), w.size, C.byref(self._cfg),
                                             self.device.index, C.byref(h)), "icrec_encoder_create")
        self._h = h
        self._ws_slots: dict[int, Optional[torch.Tensor]] = {}
        self._side: Optional[torch.cuda.Stream] = None

    def close(self) -> None:
        if getattr(self, "_h", None):
            _native.lib().icrec_encoder_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _workspace(self, total_tokens: int, n_seqs: int, slot: int = 0) -> torch.Tensor:
        need = int(_native.lib().icrec_encode_workspace_bytes(self._h, total_tokens, n_seqs))
        ws = self._ws_slots.get(slot)
        if ws is None or ws.numel() < need:
            self._ws_slots[slot] = None
            ws = self._ws_slots[slot] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def _encode_call(self, ids: torch.Tensor, cu: torch.Tensor, n: int, T: int, max_seqlen: int, out: torch.Tensor,
                     slot: int) -> None:
        ws = self._workspace(T, n, slot)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _native.check(_native.lib().icrec_encode(self._h, C.c_void_p(ids.data_ptr()), C.c_void_p(cu.data_ptr()), n, T,
                                                 int(max_seqlen), C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(ws.data_ptr()), ws.numel(), st), "icrec_encode")

    def encode_packed(self, ids: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int,
                      out: Optional[torch.Tensor] = None, cu_host: Optional[np.ndarray] = None) -> torch.Tensor:
        """Device tensors in (int32 ids[T], int32 cu_seqlens[n+1]) -> float32 [n, hidden] on the device.

        With `cu_host` (the host copy of cu_seqlens) and a large batch, the two halves of the batch run
        concurrently on two HIP streams (own workspaces, fork/join by events): one half's bandwidth-bound
        kernels overlap the other half's MFMA-bound GEMMs (-6 % per batch of 1,024 contexts).  Results are
        identical: every kernel is per-sequence / per-token."""
        if ids.dtype != torch.int32 or cu_seqlens.dtype != torch.int32:
            raise TypeError("ids and cu_seqlens must be int32")
        ids = ids.to(self.device).contiguous()
        cu = cu_seqlens.to(self.device).contiguous()
        n, T = int(cu.numel()) - 1, int(ids.numel())
        if out is None:
            out = torch.empty((n, self.shape.hidden), dtype=torch.float32, device=self.device)
        if cu_host is None or n < 2 * self.SPLIT_MIN_SEQS or T < 2 * self.SPLIT_MIN_TOKENS:
            self._encode_call(ids, cu, n, T, max_seqlen, out, 0)
            return out
        half = n // 2
        t_half = int(cu_host[half])
        # the second half's rebased cu_seqlens, computed per call on the caller's stream (one tiny kernel).  Never
        # cached by address: the allocator hands the next batch's cu tensor the same address, and two different
        # batches with equal n and t_half would then share stale sequence boundaries.
        cu_b = cu[half:] - t_half
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            self._encode_call(ids[t_half:], cu_b, n - half, T - t_half, max_seqlen, out[half:], 1)
        self._encode_call(ids[:t_half], cu[: half + 1], half, t_half, max_seqlen, out[:half], 0)
        main.wait_stream(self._side)
        return out

    def encode_packed_host(self, ids: np.ndarray, cu: np.ndarray, max_tokens_per_call: int = 1 << 18) -> torch.Tensor:
        """Host arrays in the packed form (ids int32[T], cu_seqlens int32[n+1], e.g. HostTokenizer.packed) ->
        embeddings [n, hidden] on the device, in input order; split into calls of at most
        `max_tokens_per_call` tokens like encode_ids."""
        n = int(cu.shape[0]) - 1
        out = torch.empty((n, self.shape.hidden), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        lens = np.diff(cu)
        if (lens < 1).any():
            raise ValueError("every sequence needs at least one token")
        if int(lens.max()) > MAX_SEQ_LEN:
            raise ValueError(f"sequence longer than max_seq_length={MAX_SEQ_LEN}; truncate on the host")
        if int(ids.min()) < 0 or int(ids.max()) >= self.shape.vocab_size:
            raise ValueError(f"token id out of range [0, {self.shape.vocab_size})")
        start = 0
        while start < n:
            end = int(np.searchsorted(cu, cu[start] + max_tokens_per_call, side="right")) - 1
            end = min(max(end, start + 1), n)
            t0, t1 = int(cu[start]), int(cu[end])
            cu_c = np.ascontiguousarray(cu[start:end + 1] - t0, dtype=np.int32)
            self.encode_packed(torch.from_numpy(np.ascontiguousarray(ids[t0:t1])).to(self.device, non_blocking=True),
                               torch.from_numpy(cu_c).to(self.device, non_blocking=True), int(lens[start:end].max()),
                               out=out[start:end], cu_host=cu_c)
            start = end
        return out

    def encode_ids(self, seqs: Sequence[Sequence[int]], max_tokens_per_call: int = 1 << 18) -> torch.Tensor:
        """Host token-id lists -> embeddings [n, hidden] on the device, in input order.
        Long inputs are split into calls of at most `max_tokens_per_call` tokens."""
        n = len(seqs)
        out = torch.empty((n, self.shape.hidden), dtype=torch.float32, device=self.device)
        start = 0
        while start < n:
            tok, end = 0, start
            while end < n and (end == start or tok + len(seqs[end]) <= max_tokens_per_call):
                tok += len(seqs[end])
                end += 1
            ids, cu, mx = pack_token_ids(seqs[start:end])
            vmax = int(ids.max()) if ids.size else 0
            if ids.min() < 0 or vmax >= self.shape.vocab_size:
                raise ValueError(f"token id out of range [0, {self.shape.vocab_size})")
            self.encode_packed(torch.from_numpy(ids).to(self.device, non_blocking=True),
                               torch.from_numpy(cu).to(self.device, non_blocking=True), mx, out=out[start:end],
                               cu_host=cu)
            start = end
        return out
