"""Single-request fast path: icrec_encode + icrec_search replayed from a hipGraph.

The reference's primary call is one query per request (serve_recommendations.py:206-225).  At
Q = 1 the ~44 kernels of a request are launch-bound; libicrec's hot calls never allocate or
synchronise, so the whole request is captured once per (token bucket, k) and replayed: the host
does one small H2D (ids + exclusions), one graph launch and one D2H of k results.

Token buckets coincide with the attention kernel's length buckets (32/64/128/256 tokens): the
captured launch processes `bucket` rows, the real length travels in cu_seqlens on the device, so
rows past it are computed and ignored (pooling and attention read cu_seqlens).
"""
from __future__ import annotations

from typing import Optional, Sequence

import ctypes as C

import numpy as np
import torch

from . import _native
from .encoder import DeviceEncoder
from .search import DeviceIndex

BUCKETS = (32, 64, 128, 256)
MAX_EXCLUDED = 1024


class _Captured:
    """One captured (token bucket, k) shape.  Every graph node costs ~5 us on this system whatever it does (kernel trace
    of a replay: profiles/r04_single_request_anatomy.txt), so the request carries no copy nodes at all: the kernels read
    ids / cu_seqlens / exclusions as VIEWS of the one device buffer the host fills with a single H2D copy, and the last
    kernel writes the k results straight into pinned host memory (device-accessible at the same address under ROCm's
    unified addressing) — round 3's graph had four device-to-device and two device-to-host copy nodes around the 47
    kernels."""

    def __init__(self, enc: DeviceEncoder, index: DeviceIndex, bucket: int, k: int):
        dev = enc.device
        self.bucket, self.k = bucket, k
        # device staging, one H2D per request: [cu (2) | excl_off (2) | ids (bucket) | excl_idx (MAX_EXCLUDED)]; the views
        # the kernels read (cu first: 8-byte aligned pairs, ids 16-byte aligned)
        self.h_in = torch.zeros(4 + bucket + MAX_EXCLUDED, dtype=torch.int32).pin_memory()
        self.d_in = torch.zeros_like(self.h_in, device=dev)
        self.cu = self.d_in[0:2]
        self.excl_off = self.d_in[2:4]
        self.ids = self.d_in[4:4 + bucket]
        self.excl_idx = self.d_in[4 + bucket:]
        self.emb = torch.empty((1, enc.shape.hidden), dtype=torch.float32, device=dev)
        # results: written by icrec_search directly into pinned host memory
        self.h_idx = torch.full((1, k), -1, dtype=torch.int64).pin_memory()
        self.h_sc = torch.zeros((1, k), dtype=torch.float32).pin_memory()
        self.out_idx, self.out_sc = self.h_idx, self.h_sc

        # the graph bakes raw pointers: it must own its scratch memory (the encoder's / index's shared
        # workspaces are re-allocated when a later, larger un-captured call needs more room)
        L = _native.lib()
        self.enc_ws = torch.empty(int(L.icrec_encode_workspace_bytes(enc._h, bucket, 1)), dtype=torch.uint8, device=dev)
        self.srch_ws = torch.empty(int(L.icrec_search_workspace_bytes(index._h, 1, k)), dtype=torch.uint8, device=dev)
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

        def body_encode():
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _native.check(L.icrec_encode(enc._h, P(self.ids), P(self.cu), 1, bucket, bucket, P(self.emb),
                                         P(self.enc_ws), self.enc_ws.numel(), st), "icrec_encode")

        def body_search():
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _native.check(L.icrec_search(index._h, P(self.emb), 1, k, P(self.excl_idx), P(self.excl_off),
                                         P(self.out_idx), P(self.out_sc), P(self.srch_ws), self.srch_ws.numel(), st),
                          "icrec_search")

        self._bodies = (body_encode, body_search)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up: function attributes, workspaces
            for _ in range(2):
                body_encode()
                body_search()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            body_encode()
            body_search()
        self.timed: Optional[tuple] = None

    def timed_graphs(self):
        """(encode graph, search graph, three timing events): the same request as `graph` cut at the encode /
        search seam, so that MonitoredRecommender's query_embedding_time_ms / similarity_compute_time_ms come
        from HIP events around the two replays (same buffers, same kernels; captured on first use)."""
        if self.timed is None:
            torch.cuda.synchronize(self.ids.device)
            graphs = []
            for body in self._bodies:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    body()
                graphs.append(g)
            self.timed = (graphs[0], graphs[1], tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)))
        return self.timed


class SingleRequestPath:
    """encode + search for ONE query through a replayed hipGraph."""

    def __init__(self, encoder: DeviceEncoder, index: DeviceIndex):
        self.encoder, self.index = encoder, index
        self._graphs: dict[tuple[int, int], _Captured] = {}
        self._handles = self._handle_key()

    def _handle_key(self):
        return (self.encoder._h.value if self.encoder._h else None, self.index._h.value if self.index._h else None)

    def invalidate(self) -> None:
        """Drop every captured graph (they bake raw handles, buffers and workspaces)."""
        self._graphs.clear()
        self._handles = self._handle_key()

    def supports(self, n_tokens: int, k: int, n_excluded: int) -> bool:
        return 1 <= n_tokens <= BUCKETS[-1] and n_excluded <= MAX_EXCLUDED and k <= self.index.n_rows

    def run(self, ids: Sequence[int], k: int, excluded_rows: Optional[Sequence[int]] = None, timed: bool = False):
        """-> (row indices int64 [k], scores float32 [k]) on the host; -1 / 0 padded like icrec_search.
        timed=True: the request runs as two replays (encode, search) bracketed by HIP events and the call
        returns (idx, scores, encode_ms, search_ms)."""
        n = len(ids)
        ex = sorted(set(int(r) for r in excluded_rows)) if excluded_rows else []
        if not self.supports(n, k, len(ex)):
            raise ValueError("request outside the captured fast path")
        if self._handle_key() != self._handles:  # encoder / index re-created under us (close() + new handle)
            self.invalidate()
        if self._handles[0] is None or self._handles[1] is None:
            raise _native.IcrecError("SingleRequestPath: encoder or index handle is closed")
        bucket = next(b for b in BUCKETS if n <= b)
        c = self._graphs.get((bucket, k))
        if c is None:
            c = self._graphs[(bucket, k)] = _Captured(self.encoder, self.index, bucket, k)
        h = c.h_in
        h[0] = 0
        h[1] = n
        h[2] = 0
        h[3] = len(ex)
        h[4:4 + n] = torch.as_tensor(ids, dtype=torch.int32)
        h[4 + n:4 + bucket] = 0
        if ex:
            h[4 + bucket:4 + bucket + len(ex)] = torch.as_tensor(ex, dtype=torch.int32)
        stream = torch.cuda.current_stream(self.encoder.device)
        if timed:
            g_enc, g_srch, (e0, e1, e2) = c.timed_graphs()
        c.d_in.copy_(h, non_blocking=True)
        if timed:
            e0.record(stream)
            g_enc.replay()
            e1.record(stream)
            g_srch.replay()
            e2.record(stream)
        else:
            c.graph.replay()
        stream.synchronize()  # the last kernel wrote the results into pinned host memory
        if timed:
            return c.h_idx[0].numpy().copy(), c.h_sc[0].numpy().copy(), e0.elapsed_time(e1), e1.elapsed_time(e2)
        return c.h_idx[0].numpy().copy(), c.h_sc[0].numpy().copy()
