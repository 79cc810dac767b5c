"""Failure-path and observability tests of the serving loop (CPU only; the recommender is a stub):
graceful batcher stop under a corpus swap, per-request `recommendation_served` records under batching
(reference: src/inference/serve_recommendations.py:268-293, read back at src/api/routes/recommend.py:163-173),
loss of the GPU-owner process seen from a front-end, HTTP framing of the small asyncio server, child supervision."""
from __future__ import annotations

import asyncio
import contextlib
import logging
import multiprocessing as mp
import os
import tempfile
import time

import pytest

from instacart_next_order_recommendation_amd.api import fastserve
from instacart_next_order_recommendation_amd.api.batcher import BatcherStopped, MicroBatcher
from instacart_next_order_recommendation_amd.api.remote import RemoteBatcher, WorkerUnavailable
from instacart_next_order_recommendation_amd.api.serve import supervise
from instacart_next_order_recommendation_amd.api.worker import GpuWorker, frame, read_frame
from instacart_next_order_recommendation_amd.recommender import MonitoredRecommender


class SlowRec:
    """recommend_batch that takes a while, so that requests pile up behind the batch on the 'GPU'."""

    def __init__(self, tag: str, delay: float = 0.05):
        self.tag, self.delay, self.batches = tag, delay, []
        self.product_ids = ["a", "b"]

    def recommend_batch(self, queries, top_k, excl):
        time.sleep(self.delay)
        self.batches.append(len(queries))
        return [[(f"{self.tag}:{q}", 1.0)] for q in queries]


def test_stop_serves_everything_already_queued():
    async def main():
        b = MicroBatcher(SlowRec("old"), max_batch=4, max_wait_ms=0)
        futs = [asyncio.ensure_future(b.submit(f"q{i}", 1, None)) for i in range(13)]
        await asyncio.sleep(0.01)  # the first batch is on the "GPU", the rest are queued
        await b.stop()
        done = await asyncio.wait_for(asyncio.gather(*futs), 5)
        assert [r[0][0][0] for r in done] == [f"old:q{i}" for i in range(13)]
        with pytest.raises(BatcherStopped):
            await b.submit("late", 1, None)

    asyncio.run(main())


def test_stop_of_an_idle_batcher_returns():
    async def main():
        b = MicroBatcher(SlowRec("x"))
        await b.start()
        await asyncio.wait_for(b.stop(), 2)
        b2 = MicroBatcher(SlowRec("y"))
        await asyncio.wait_for(b2.stop(), 2)  # never started

    asyncio.run(main())


def test_worker_corpus_swap_completes_every_request():
    """Requests fired concurrently with a `corpus` message: each one is answered, by the old catalog or the new."""
    made = []

    def factory(cp):
        made.append(str(cp))
        return SlowRec(os.path.basename(str(cp)), delay=0.02)

    class W:
        def __init__(self):
            self.frames = []

        def write(self, b):
            self.frames.append(b)

    async def main():
        w = GpuWorker("model", "/tmp/c0.json", factory=factory)
        out = W()
        tasks = [asyncio.ensure_future(w._one(out, ["rec", i, f"q{i}", 1, [], f"user{i}"])) for i in range(40)]
        await asyncio.sleep(0.005)
        tasks.append(asyncio.ensure_future(w._one(out, ["corpus", 1000, "/tmp/c1.json"])))
        tasks += [asyncio.ensure_future(w._one(out, ["rec", 100 + i, f"r{i}", 1, [], None])) for i in range(40)]
        await asyncio.wait_for(asyncio.gather(*tasks), 10)
        await w.batcher.stop()
        import msgpack
        import struct

        msgs = [msgpack.unpackb(f[4:4 + struct.unpack("<I", f[:4])[0]], raw=False) for f in out.frames]
        kinds = [m[0] for m in msgs]
        assert kinds.count("ok") == 80 and kinds.count("corpus") == 1 and "err" not in kinds, kinds
        assert made == ["/tmp/c0.json", "/tmp/c1.json"]

    asyncio.run(main())


class StubMonitored(MonitoredRecommender):
    """A MonitoredRecommender without a model: only what the batcher and note_served touch."""

    def __init__(self):  # noqa: D107 - no super().__init__: nothing is loaded
        self.metrics_logger = logging.getLogger("recommender.metrics")
        self.last_metrics = None
        self.product_ids = ["p1", "p2"]

    def recommend_batch_timed(self, queries, top_k, excl):
        return [[("p1", 0.9), ("p2", 0.5)][:top_k] for _ in queries], 1.5, 0.25


def test_batched_requests_log_recommendation_served_with_user_id(caplog):
    rec = StubMonitored()

    async def main():
        b = MicroBatcher(rec, max_batch=8, max_wait_ms=1)
        res = await asyncio.gather(*[b.submit(f"q{i}", 2, None, f"user{i}" if i else None) for i in range(5)])
        await b.stop()
        return res

    with caplog.at_level(logging.INFO, logger="recommender.metrics"):
        res = asyncio.run(main())
    assert all(r[0] == [("p1", 0.9), ("p2", 0.5)] for r in res)
    recs = [r for r in caplog.records if r.getMessage() == "recommendation_served"]
    assert sorted(r.user_id for r in recs) == ["anonymous", "user1", "user2", "user3", "user4"]
    r0 = recs[0]
    assert r0.encode_time_ms == 1.5 and r0.similarity_time_ms == 0.25 and r0.num_results == 2
    assert r0.top_score == 0.9 and abs(r0.avg_score - 0.7) < 1e-12
    m = rec.last_metrics
    assert m.num_recommendations == 2 and m.query_embedding_time_ms == 1.5 and m.similarity_compute_time_ms == 0.25
    # level-gated: nothing is built or emitted when the metrics logger is above INFO
    caplog.clear()
    logging.getLogger("recommender.metrics").setLevel(logging.WARNING)
    try:
        rec.note_served([("p1", 0.9)], "u", 1.0, 1.0, 2.0)
        assert not caplog.records and rec.last_metrics.user_id == "u"
    finally:
        logging.getLogger("recommender.metrics").setLevel(logging.NOTSET)


def test_front_end_fails_fast_when_the_gpu_worker_is_gone_and_reconnects():
    sock = os.path.join(tempfile.mkdtemp(prefix="icrec_t_"), "w.sock")

    async def main():
        async def handle(reader, writer):  # answers ONE request, then drops the connection (a crashed worker)
            msg = await read_frame(reader)
            writer.write(frame(["ok", msg[1], [["p", 0.5]], 1.0, 2.0, 1]))
            await writer.drain()
            writer.close()

        server = await asyncio.start_unix_server(handle, path=sock)
        rb = RemoteBatcher(sock, call_timeout=2.0)
        res, tm = await rb.submit("q", 1, None, "u1")
        assert res == [("p", 0.5)] and tm.batch_size == 1
        await asyncio.sleep(0.05)  # the read loop sees EOF
        assert not rb.connected
        server.close()
        await server.wait_closed()
        os.unlink(sock)
        with pytest.raises(WorkerUnavailable):  # nothing listens: fail at once, no hang
            await asyncio.wait_for(rb.submit("q", 1, None), 1)
        server = await asyncio.start_unix_server(handle, path=sock)  # the supervisor brought a new worker up
        res, _ = await asyncio.wait_for(rb.submit("q2", 1, None), 2)
        assert res == [("p", 0.5)]
        await rb.stop()
        server.close()

    asyncio.run(main())


def test_front_end_times_out_on_a_silent_worker():
    sock = os.path.join(tempfile.mkdtemp(prefix="icrec_t_"), "w.sock")

    async def main():
        async def handle(reader, writer):
            await asyncio.sleep(5)

        server = await asyncio.start_unix_server(handle, path=sock)
        rb = RemoteBatcher(sock, call_timeout=0.2)
        with pytest.raises(WorkerUnavailable):
            await rb.submit("q", 1, None)
        await rb.stop()
        server.close()

    asyncio.run(main())


class _EchoApp:
    class router:
        @staticmethod
        def lifespan_context(app):
            @contextlib.asynccontextmanager
            async def ctx():
                yield

            return ctx()

    async def __call__(self, scope, receive, send):
        body = (await receive())["body"]
        await send({"type": "http.response.start", "status": 200, "headers": []})
        await send({"type": "http.response.body", "body": b"len=%d" % len(body)})


def test_fastserve_request_framing():
    async def main():
        import socket

        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        task = asyncio.create_task(fastserve.serve(_EchoApp(), "127.0.0.1", port, reuse_port=False))
        await asyncio.sleep(0.2)

        async def req(raw: bytes, more: bytes = b""):
            r, w = await asyncio.open_connection("127.0.0.1", port)
            w.write(raw)
            await w.drain()
            data = await asyncio.wait_for(r.read(4096), 2)
            if more:
                w.write(more)
                await w.drain()
                data += await asyncio.wait_for(r.read(4096), 2)
            w.close()
            return data

        assert (await req(b"POST / HTTP/1.1\r\ncontent-length: 3\r\n\r\nabc")).endswith(b"len=3")
        assert (await req(b"POST / HTTP/1.1\r\ncontent-length: abc\r\n\r\nabc")).startswith(b"HTTP/1.1 400")
        assert (await req(b"POST / HTTP/1.1\r\ncontent-length: -5\r\n\r\nabc")).startswith(b"HTTP/1.1 400")
        assert (await req(b"POST / HTTP/1.1\r\ncontent-length: 99999999999\r\n\r\n")).startswith(b"HTTP/1.1 413")
        assert (await req(b"POST / HTTP/1.1\r\ntransfer-encoding: chunked\r\n\r\n3\r\nabc\r\n0\r\n\r\n")).startswith(b"HTTP/1.1 501")
        got = await req(b"POST / HTTP/1.1\r\ncontent-length: 3\r\nexpect: 100-continue\r\n\r\n", b"abc")
        assert got.startswith(b"HTTP/1.1 100 Continue\r\n\r\n") and got.endswith(b"len=3")
        task.cancel()

    asyncio.run(main())


def _sleeper(t):
    time.sleep(t)


def _crasher():
    os._exit(3)


def test_supervise_stops_the_server_when_a_child_dies():
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sleeper, args=(60,), daemon=True), ctx.Process(target=_crasher, daemon=True)]
    for p in procs:
        p.start()
    t0 = time.time()
    code = supervise(procs, poll_s=0.05)
    assert code == 3 and time.time() - t0 < 20
    assert not any(p.is_alive() for p in procs)
