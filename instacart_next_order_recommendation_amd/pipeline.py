"""Two-deep pipeline over a stream of query batches: host tokenisation of batch i+1 runs on a worker thread
(icrec_tokenize releases the GIL) while the GPU encodes and searches batch i, and batch i's results are read
back (pinned buffer + event) only after batch i+1 has been launched, so the device never waits for the host.

The reference serves one request at a time (`routes/recommend.py:139-151` calls `recommend()` synchronously inside
the event loop); this is the throughput form of the same path for callers that hold many contexts — the evaluation
consumers (`src/baselines/content_based.py:38-64`) and the micro-batching worker.  Per batch the launches are
exactly those of `Recommender.recommend_batch`: results are identical.  The similarity search of batch i (0.3 ms that
cannot fill the chip: sixteen query blocks) runs on a second stream behind an event, beside the first kernels of batch
i+1's encode on the caller's stream.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, Iterator, Optional, Sequence

import numpy as np
import torch


def pipelined_search(tokenizer: Callable[[Sequence[str]], list], encoder, search: Callable, batches: Iterable[Sequence[str]],
                     k: int, exclude: Optional[Callable[[int], Optional[list]]] = None
                     ) -> Iterator[tuple[np.ndarray, np.ndarray]]:
    """Yields (idx int64 [n, k], score float32 [n, k]) per batch, in batch order.

    tokenizer: texts -> list of token-id lists (host); if it has a `packed(texts)` method (HostTokenizer,
    NativeTokenizer) that form is used instead: (ids, cu_seqlens) arrays without per-sequence Python lists.
    encoder: DeviceEncoder.  search: (emb, k, exclude_lists) -> (idx, score) device tensors.
    exclude(i): per-query exclusion rows of batch i, or None."""
    device = encoder.device
    stream = torch.cuda.current_stream(device)
    # searches + read-back; one search at a time (the index workspace is per handle).  ICREC_PIPELINE_SIDE=0: everything on
    # the caller's stream (A/B)
    side = torch.cuda.Stream(device) if os.getenv("ICREC_PIPELINE_SIDE", "1") != "0" else stream
    it = iter(batches)

    packed = getattr(tokenizer, "packed", None)

    def fetch(pool):
        b = next(it, None)
        return None if b is None else pool.submit(packed or tokenizer, list(b))

    def collect(p):
        ev, idx_h, sc_h = p
        ev.synchronize()
        return idx_h.numpy().copy(), sc_h.numpy().copy()

    def _run(pool):
        fut = fetch(pool)
        pending = None
        i = 0
        while fut is not None:
            ids = fut.result()
            fut = fetch(pool)                      # batch i+1 is tokenised while batch i is launched and runs
            if (ids[1].shape[0] > 1) if packed else bool(ids):
                emb = encoder.encode_packed_host(*ids) if packed else encoder.encode_ids(ids)
                encoded = torch.cuda.Event()
                encoded.record(stream)
                excl = exclude(i) if exclude is not None else None
                with torch.cuda.stream(side):
                    side.wait_event(encoded)
                    emb.record_stream(side)  # allocated on the caller's stream, last read here
                    idx_d, sc_d = search(emb, k, excl)
                    idx_h = torch.empty(idx_d.shape, dtype=idx_d.dtype, pin_memory=True)
                    sc_h = torch.empty(sc_d.shape, dtype=sc_d.dtype, pin_memory=True)
                    idx_h.copy_(idx_d, non_blocking=True)
                    sc_h.copy_(sc_d, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(side)
                cur = (ev, idx_h, sc_h)
            else:
                cur = None
            if pending is not None:
                yield collect(pending) if pending != "empty" else (np.zeros((0, k), np.int64), np.zeros((0, k), np.float32))
            pending = cur if cur is not None else "empty"
            i += 1
        if pending is not None:
            yield collect(pending) if pending != "empty" else (np.zeros((0, k), np.int64), np.zeros((0, k), np.float32))

    with ThreadPoolExecutor(max_workers=1) as pool:
        try:
            yield from _run(pool)
        finally:
            # also when the consumer abandons the generator: later work on the caller's stream (another search on this
            # index, whose workspace is per handle) stays ordered behind the last search issued here
            if side is not stream:
                stream.wait_stream(side)
