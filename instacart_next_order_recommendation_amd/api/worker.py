"""The GPU-owner process of the multi-process server (serve.py).

One process holds the MonitoredRecommender (model + catalog index in HBM) and runs the MicroBatcher; N
front-end processes (FastAPI app, one event loop each) parse / validate / serialise HTTP and forward
(query, top_k, exclusions) over a Unix socket.  Frames: 4-byte little-endian length + msgpack.

    front-end -> worker   ["rec", rid, query, top_k, [excluded product ids], user_id | None]
                          ["corpus", rid, corpus_path]              (POST /admin/corpus: re-index)
    worker -> front-end   ["ok", rid, [[pid, score], ...], encode_ms, search_ms, batch_size]
                          ["err", rid, message]
                          ["corpus", rid | None, corpus_path, n_products]   (broadcast after a re-index)

Per request the worker pays one msgpack decode/encode and its share of a batched tokenise + GPU pass; the
per-request Python that caps a single FastAPI process (~0.16 ms: pydantic, routing, JSON) runs in parallel in
the front-ends.  The reference has a single blocking process (src/api/routes/recommend.py:139-151).
"""
from __future__ import annotations

import asyncio
import logging
import os
import struct
from pathlib import Path

import msgpack

from .batcher import BatcherStopped

logger = logging.getLogger(__name__)


async def read_frame(reader: asyncio.StreamReader):
    head = await reader.readexactly(4)
    return msgpack.unpackb(await reader.readexactly(struct.unpack("<I", head)[0]), raw=False)


def frame(obj) -> bytes:
    b = msgpack.packb(obj, use_bin_type=True)
    return struct.pack("<I", len(b)) + b


class GpuWorker:
    def __init__(self, model_dir, corpus_path, factory=None):
        """factory(corpus_path) -> recommender; default: MonitoredRecommender on this process's GPU."""
        from .batcher import MicroBatcher

        if factory is None:
            from ..recommender import MonitoredRecommender

            def factory(cp):
                return MonitoredRecommender(model_dir=model_dir, corpus_path=cp)
        self._mk = factory
        self._Batcher = MicroBatcher
        self.model_dir, self.corpus_path = model_dir, Path(corpus_path)
        self.recommender = self._mk(self.corpus_path)
        self.batcher = self._new_batcher(self.recommender)
        self.writers: set[asyncio.StreamWriter] = set()

    def _new_batcher(self, rec):
        return self._Batcher(rec, max_batch=int(os.getenv("BATCH_MAX_SIZE", "1024")),
                             max_wait_ms=float(os.getenv("BATCH_MAX_WAIT_MS", "2")))

    async def _one(self, writer, msg):
        kind, rid = msg[0], msg[1]
        try:
            if kind == "rec":
                for attempt in (0, 1):
                    try:
                        results, tm = await self.batcher.submit(msg[2], int(msg[3]), set(msg[4]) if msg[4] else None,
                                                                msg[5] if len(msg) > 5 else None)
                        break
                    except BatcherStopped:  # raced a corpus swap: self.batcher is the new one by now
                        if attempt:
                            raise
                writer.write(frame(["ok", rid, results, tm.encode_ms, tm.search_ms, tm.batch_size]))
            elif kind == "corpus":
                # build the NEW recommender off the event loop (a full GPU re-encode), keep serving the old one
                # meanwhile, then swap batcher + recommender in one step and tell every front-end; the old batcher
                # stops gracefully: what it has queued (and the batch on the GPU) is still answered
                new_rec = await asyncio.get_running_loop().run_in_executor(None, self._mk, Path(msg[2]))
                old = self.batcher
                self.recommender, self.batcher, self.corpus_path = new_rec, self._new_batcher(new_rec), Path(msg[2])
                await old.stop()
                note = frame(["corpus", None, str(self.corpus_path), len(new_rec.product_ids)])
                for w in list(self.writers):
                    if w is not writer:
                        w.write(note)
                writer.write(frame(["corpus", rid, str(self.corpus_path), len(new_rec.product_ids)]))
            else:
                writer.write(frame(["err", rid, f"unknown message {kind!r}"]))
        except Exception as exc:  # noqa: BLE001 - the caller must see the failure
            writer.write(frame(["err", rid, f"{type(exc).__name__}: {exc}"]))

    async def handle(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter):
        self.writers.add(writer)
        try:
            while True:
                msg = await read_frame(reader)
                asyncio.get_running_loop().create_task(self._one(writer, msg))
        except (asyncio.IncompleteReadError, ConnectionResetError):
            pass
        finally:
            self.writers.discard(writer)
            writer.close()


def _settle_heap() -> None:
    """The catalog (two 49,688-entry containers of strings), the tokenizer vocabulary and the imported modules are
    millions of objects that live as long as the process: a full collection walks all of them - tens of milliseconds
    in which this single-threaded process answers nobody, every few seconds at >10 k requests per second (each leaves a
    handful of short-lived containers behind).  Seen as a p95 of 64 ms against round 2's 24 ms at the same throughput
    (tools/http_ab.sh, round 4).  gc.freeze() moves everything alive now into the permanent generation: later full
    collections only look at what was allocated since."""
    import gc

    gc.collect()
    gc.freeze()


def run(sock_path: str, model_dir, corpus_path, ready=None) -> None:
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    # The reference logs one `recommendation_served` record per request at INFO (serve_recommendations.py:268-278); so
    # does this process by default, in either launch mode.  At >10 k requests per second on ONE Python thread
    # the records cost measurable throughput: METRICS_LOG_LEVEL=WARNING turns them off (bench.py's HTTP leg and
    # tools/http_load.py runs say so where they do); last_metrics / the response's `stats` are unaffected.
    logging.getLogger("recommender.metrics").setLevel(os.getenv("METRICS_LOG_LEVEL", "INFO").upper())

    async def main():
        w = GpuWorker(model_dir, corpus_path)
        _settle_heap()
        server = await asyncio.start_unix_server(w.handle, path=sock_path, limit=1 << 26)
        logger.info("GPU worker ready on %s (%d products)", sock_path, len(w.recommender.product_ids))
        if ready is not None:
            ready()
        async with server:
            await server.serve_forever()

    asyncio.run(main())
