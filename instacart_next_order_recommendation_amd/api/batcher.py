"""Micro-batching in front of the recommender.

The reference calls the blocking recommend() inside an `async def` endpoint
(src/api/routes/recommend.py:89,139-151): one request in flight per process.  Here concurrent
requests are coalesced: everything queued while the previous GPU pass was running (plus, when the
server has seen concurrency, arrivals within `max_wait_ms`, up to `max_batch`) goes into ONE
recommend_batch() GPU pass in a worker thread and every caller gets its own slice.  Results are identical to per-request recommend() calls
because packed varlen encoding and per-query top-k are batch-invariant.
"""
from __future__ import annotations

import asyncio
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Optional


@dataclass
class BatchTimings:
    batch_size: int
    encode_ms: float
    search_ms: float
    total_ms: float


@dataclass
class _Pending:
    query: str
    top_k: int
    exclude: Optional[set]
    future: asyncio.Future
    user_id: Optional[str] = None


class BatcherStopped(RuntimeError):
    """submit() on a batcher that has been stopped (its recommender was swapped out): fetch the current one."""


class MicroBatcher:
    def __init__(self, recommender, max_batch: int = 256, max_wait_ms: float = 2.0):
        self.recommender = recommender
        self.max_batch = int(max_batch)
        self.max_wait = float(max_wait_ms) / 1000.0
        self._queue: Optional[asyncio.Queue] = None
        self._task: Optional[asyncio.Task] = None
        self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="icrec-gpu")  # one GPU stream
        self._closed = False

    async def start(self) -> None:
        if self._task is None:
            self._queue = asyncio.Queue()
            self._task = asyncio.create_task(self._run())

    async def stop(self) -> None:
        """Graceful: no new requests are taken, everything already queued (and the batch on the GPU) is served,
        then the loop task ends.  A request that raced the stop and is still queued afterwards fails with an
        error instead of waiting forever (a cancelled loop task would never resolve its future)."""
        self._closed = True
        task, self._task = self._task, None
        if task is not None:
            await self._queue.put(None)  # wake the loop; it drains what is queued ahead of the sentinel
            try:
                await task
            except asyncio.CancelledError:  # our own caller was cancelled: do not leave waiters behind
                task.cancel()
                raise
            finally:
                self._fail_queued(BatcherStopped("batcher stopped"))
        self._pool.shutdown(wait=False)

    def _fail_queued(self, exc: Exception) -> None:
        q = self._queue
        while q is not None and not q.empty():
            p = q.get_nowait()
            if p is not None and not p.future.done():
                p.future.set_exception(exc)

    async def submit(self, query: str, top_k: int, exclude: Optional[set], user_id: Optional[str] = None):
        """-> (results, BatchTimings) for this request."""
        if self._closed:
            raise BatcherStopped("batcher stopped")
        if self._task is None:
            await self.start()
        fut = asyncio.get_running_loop().create_future()
        await self._queue.put(_Pending(query, top_k, exclude, fut, user_id))
        return await fut

    def _execute(self, batch: list[_Pending]):
        t0 = time.perf_counter()
        k = max(p.top_k for p in batch)
        excl = [p.exclude for p in batch]
        rec = self.recommender
        if hasattr(rec, "recommend_batch_timed"):
            results, enc_ms, srch_ms = rec.recommend_batch_timed([p.query for p in batch], k, excl)
        else:
            results, enc_ms, srch_ms = rec.recommend_batch([p.query for p in batch], k, excl), 0.0, 0.0
        tm = BatchTimings(len(batch), enc_ms, srch_ms, (time.perf_counter() - t0) * 1000)
        out = [r[: p.top_k] for r, p in zip(results, batch)]
        # per-request observability as in the reference's MonitoredRecommender.recommend (:268-278): last_metrics and
        # one `recommendation_served` record per request, with its user_id; the timings are the batch's
        note = getattr(rec, "note_served", None)
        if note is not None:
            for p, r in zip(batch, out):
                note(r, p.user_id, enc_ms, srch_ms, tm.total_ms)
        return out, tm

    def _drain(self, batch: list) -> bool:
        """Move queued requests into `batch`; True when the stop sentinel was reached (it stays consumed)."""
        q = self._queue
        while len(batch) < self.max_batch and not q.empty():
            p = q.get_nowait()
            if p is None:
                return True
            batch.append(p)
        return False

    async def _run(self) -> None:
        """Continuous batching: while one GPU pass runs in the worker thread, new requests pile up in the
        queue and the next pass takes all of them at once (no per-request timers).  The `max_wait` window is
        only opened when the server is idle AND has recently seen concurrency — a lone sequential client is
        never made to wait for company that is not coming."""
        loop = asyncio.get_running_loop()
        last_size = 1
        stopping = False
        while not stopping:
            first = await self._queue.get()
            if first is None:  # stop(): nothing was queued ahead of the sentinel
                return
            batch = [first]
            await asyncio.sleep(0)  # let every request that is already scheduled enqueue itself
            stopping = self._drain(batch)
            if not stopping and len(batch) < self.max_batch and self.max_wait > 0 and (last_size > 1 or len(batch) > 1):
                deadline = loop.time() + self.max_wait
                while len(batch) < self.max_batch and not stopping:
                    remaining = deadline - loop.time()
                    if remaining <= 0:
                        break
                    await asyncio.sleep(min(remaining, 0.0005))
                    stopping = self._drain(batch)
            last_size = len(batch)
            try:
                results, tm = await loop.run_in_executor(self._pool, self._execute, batch)
                for p, r in zip(batch, results):
                    if not p.future.done():
                        p.future.set_result((r, tm))
            except Exception as exc:  # noqa: BLE001 - every waiter must see the failure
                for p in batch:
                    if not p.future.done():
                        p.future.set_exception(exc)
