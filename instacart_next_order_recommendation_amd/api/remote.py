"""Front-end side of the multi-process server: what `app.state.recommender` / `app.state.batcher` are in a
process that does not own the GPU (ICREC_GPU_WORKER_SOCKET set).  See worker.py for the protocol."""
from __future__ import annotations

import asyncio
import itertools
import json
from pathlib import Path
from typing import Optional

from .batcher import BatchTimings
from .worker import frame, read_frame


class CorpusView:
    """The attributes the routes read from a recommender (pid_to_text, corpus_path, product_ids) loaded from the
    corpus JSON alone — no model, no GPU (reference: Recommender._load_corpus, serve_recommendations.py:158-164)."""

    reports_stats = True   # the worker runs a MonitoredRecommender: responses carry `stats`
    remote = True

    def __init__(self, corpus_path):
        self.corpus_path = Path(corpus_path).resolve()
        with open(self.corpus_path) as f:
            corpus = json.load(f)
        self.product_ids = list(corpus.keys())
        self.pid_to_text = corpus
        self.model_dir = None

    def recommend_batch(self, *a, **k):  # pragma: no cover - marks the object as batch-capable for the route
        raise RuntimeError("front-end processes forward to the GPU worker")


class WorkerUnavailable(RuntimeError):
    """The GPU-owner process cannot be reached (the route answers 503)."""


class RemoteBatcher:
    """MicroBatcher's `submit` interface over the worker socket."""

    def __init__(self, sock_path: str, on_corpus=None, call_timeout: float = 30.0):
        self.sock_path = sock_path
        self.call_timeout = float(call_timeout)
        self._reader: Optional[asyncio.StreamReader] = None
        self._writer: Optional[asyncio.StreamWriter] = None
        self._pending: dict[int, asyncio.Future] = {}
        self._deadline: dict[int, float] = {}  # rid -> loop time after which the sweeper fails the call
        self._ids = itertools.count(1)
        self._task: Optional[asyncio.Task] = None
        self._on_corpus = on_corpus
        self._lock = asyncio.Lock()
        self._sweeper: Optional[asyncio.Task] = None

    async def start(self) -> None:
        async with self._lock:
            if self._writer is None:
                self._reader, self._writer = await asyncio.open_unix_connection(self.sock_path, limit=1 << 26)
                self._task = asyncio.create_task(self._read_loop())
                if self._sweeper is None:
                    self._sweeper = asyncio.create_task(self._sweep())

    async def stop(self) -> None:
        task, self._task = self._task, None
        if task is not None:
            task.cancel()  # the read loop's `finally` closes the transport and fails what is in flight
        sw, self._sweeper = self._sweeper, None
        if sw is not None:
            sw.cancel()

    async def _sweep(self) -> None:
        """Fails calls the worker has not answered within call_timeout.  ONE timer for all requests in flight (a
        per-request asyncio.wait_for costs a timer handle and two callbacks on the hot path)."""
        while True:
            await asyncio.sleep(min(1.0, max(self.call_timeout / 4, 0.01)))
            now = asyncio.get_running_loop().time()
            for rid in [r for r, d in self._deadline.items() if d <= now]:
                self._deadline.pop(rid, None)
                fut = self._pending.pop(rid, None)
                if fut is not None and not fut.done():
                    fut.set_exception(WorkerUnavailable(f"GPU worker did not answer within {self.call_timeout:.0f}s"))

    @property
    def connected(self) -> bool:
        return self._writer is not None and not self._writer.is_closing()

    async def _read_loop(self) -> None:
        reader = self._reader
        try:
            while True:
                msg = await read_frame(reader)
                kind, rid = msg[0], msg[1]
                if kind == "corpus" and self._on_corpus is not None:
                    self._on_corpus(msg[2])
                fut = self._pending.pop(rid, None) if rid is not None else None
                self._deadline.pop(rid, None)
                if fut is None or fut.done():
                    continue
                if kind == "ok":
                    fut.set_result(([(p, float(s)) for p, s in msg[2]], BatchTimings(int(msg[5]), msg[3], msg[4], 0.0)))
                elif kind == "corpus":
                    fut.set_result((msg[2], int(msg[3])))
                else:
                    fut.set_exception(RuntimeError(msg[2]))
        except (asyncio.IncompleteReadError, ConnectionError, OSError, asyncio.CancelledError):
            pass
        finally:
            # the connection is gone (worker died / socket reset / stop()): fail everything in flight and forget the
            # transport, so that later calls reconnect or fail at once instead of writing into a dead socket
            w, self._writer, self._reader = self._writer, None, None
            if w is not None:
                w.close()
            for fut in self._pending.values():
                if not fut.done():
                    fut.set_exception(WorkerUnavailable("GPU worker connection lost"))
            self._pending.clear()
            self._deadline.clear()

    async def _call(self, msg_tail, kind: str):
        if self._writer is None:
            try:
                await self.start()  # first use, or the previous connection was lost: (re)connect
            except OSError as exc:
                raise WorkerUnavailable(f"GPU worker unreachable: {exc}") from exc
        writer = self._writer
        if writer is None or writer.is_closing():
            raise WorkerUnavailable("GPU worker connection lost")
        rid = next(self._ids)
        loop = asyncio.get_running_loop()
        fut = loop.create_future()
        self._pending[rid] = fut
        if kind == "rec":  # a re-index legitimately takes long: no deadline
            self._deadline[rid] = loop.time() + self.call_timeout
        try:
            writer.write(frame([kind, rid, *msg_tail]))
        except Exception as exc:  # noqa: BLE001 - a transport error must not leave the future registered
            self._pending.pop(rid, None)
            self._deadline.pop(rid, None)
            raise WorkerUnavailable(f"GPU worker connection lost: {exc}") from exc
        return await fut

    async def submit(self, query: str, top_k: int, exclude, user_id: Optional[str] = None):
        return await self._call([query, int(top_k), sorted(exclude) if exclude else [], user_id], "rec")

    async def reindex(self, corpus_path: str):
        """-> (corpus_path, n_products) once the worker has swapped in the new catalog."""
        return await self._call([str(corpus_path)], "corpus")


class MultiRemoteBatcher:
    """RemoteBatcher's interface over SEVERAL GPU-owner processes on one GPU (serve.py --gpu-workers N): one worker is a
    single Python thread and saturates at ~20 k requests per second (measured: 1.03 CPUs busy beside eight front-ends at
    0.45 each); requests go to the workers round-robin, a re-index goes to every one of them."""

    def __init__(self, sock_paths, on_corpus=None, call_timeout: float = 30.0):
        self._workers = [RemoteBatcher(p, on_corpus, call_timeout) for p in sock_paths]
        self._next = itertools.cycle(range(len(self._workers)))

    async def start(self) -> None:
        for w in self._workers:
            await w.start()

    async def stop(self) -> None:
        for w in self._workers:
            await w.stop()

    @property
    def connected(self) -> bool:
        return all(w.connected for w in self._workers)

    async def submit(self, query: str, top_k: int, exclude, user_id: Optional[str] = None):
        n = len(self._workers)
        first = next(self._next)
        last_exc: Optional[Exception] = None
        for i in range(n):  # a worker that is gone must not fail the request while another one is up
            w = self._workers[(first + i) % n]
            try:
                return await w.submit(query, top_k, exclude, user_id)
            except WorkerUnavailable as exc:
                last_exc = exc
        raise last_exc if last_exc is not None else WorkerUnavailable("no GPU worker configured")

    async def reindex(self, corpus_path: str):
        """Every worker swaps in the new catalog (one after the other: each re-encodes it on the shared GPU)."""
        out = None
        for w in self._workers:
            out = await w.reindex(corpus_path)
        return out
