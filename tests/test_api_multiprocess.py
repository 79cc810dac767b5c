"""The multi-process server's plumbing without a GPU: the asyncio HTTP server (fastserve), the front-end app in
ICREC_GPU_WORKER_SOCKET mode and the worker protocol, with a stub recommender behind the worker's MicroBatcher."""
from __future__ import annotations

import asyncio
import json
import os
import socket
import tempfile
from pathlib import Path

import pytest


class _StubRecommender:
    """Deterministic stand-in: scores derive from the query length, ids from the corpus."""

    def __init__(self, corpus_path):
        self.corpus_path = Path(corpus_path)
        self.product_ids = list(json.loads(self.corpus_path.read_text()).keys())

    def recommend_batch_timed(self, queries, top_k, excl):
        out = []
        for q, e in zip(queries, excl):
            pids = [p for p in self.product_ids if not e or p not in e][:top_k]
            out.append([(p, 1.0 - 0.01 * i - 0.001 * (len(q) % 7)) for i, p in enumerate(pids)])
        return out, 1.5, 0.25


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


async def _http(port: int, method: str, path: str, obj=None, reader_writer=None):
    """One keep-alive HTTP/1.1 exchange on a raw socket -> (status, json or text)."""
    r, w = reader_writer or await asyncio.open_connection("127.0.0.1", port)
    body = json.dumps(obj).encode() if obj is not None else b""
    w.write(f"{method} {path} HTTP/1.1\r\nhost: t\r\ncontent-type: application/json\r\ncontent-length: {len(body)}\r\n\r\n".encode() + body)
    head = await r.readuntil(b"\r\n\r\n")
    status = int(head.split(b" ", 2)[1])
    clen = 0
    for ln in head.split(b"\r\n")[1:]:
        if ln.lower().startswith(b"content-length:"):
            clen = int(ln.split(b":")[1])
    data = await r.readexactly(clen)
    try:
        return status, json.loads(data), (r, w)
    except ValueError:
        return status, data.decode(), (r, w)


def test_frontend_worker_roundtrip_over_sockets(tmp_path, monkeypatch):
    from instacart_next_order_recommendation_amd.api import fastserve
    from instacart_next_order_recommendation_amd.api.worker import GpuWorker

    corpus = {str(i): f"Product: P{i}. Aisle: a. Department: d." for i in range(1, 31)}
    corpus_path = tmp_path / "eval_corpus.json"
    corpus_path.write_text(json.dumps(corpus))
    (tmp_path / "eval_queries.json").write_text(json.dumps({"7": "[+1d w1h1] Milk."}))
    sock_path = os.path.join(tempfile.mkdtemp(prefix="icrec_t_"), "w.sock")
    monkeypatch.setenv("ICREC_GPU_WORKER_SOCKET", sock_path)
    monkeypatch.setenv("CORPUS_PATH", str(corpus_path))
    monkeypatch.setenv("MODEL_DIR", str(tmp_path))
    from instacart_next_order_recommendation_amd.api.app import app

    port = _free_port()

    async def scenario():
        worker = GpuWorker(None, corpus_path, factory=_StubRecommender)
        wsrv = await asyncio.start_unix_server(worker.handle, path=sock_path)
        ready = asyncio.Event()
        server = asyncio.create_task(fastserve.serve(app, "127.0.0.1", port, reuse_port=False, ready=ready.set))
        await asyncio.wait_for(ready.wait(), 20)
        try:
            st, js, conn = await _http(port, "GET", "/health")
            assert st == 200 and js == {"status": "ok"}
            # keep-alive: several requests on the same connection, concurrent ones on others
            st, js, conn = await _http(port, "POST", "/recommend", {"user_context": "[+7d] Milk, Bread.", "top_k": 3}, conn)
            assert st == 200 and [r["product_id"] for r in js["recommendations"]] == ["1", "2", "3"]
            assert js["recommendations"][0]["product_text"] == corpus["1"]
            assert js["stats"]["query_embedding_time_ms"] == 1.5 and js["stats"]["num_recommendations"] == 3
            st, js, conn = await _http(port, "POST", "/recommend", {"user_id": "7", "top_k": 2, "exclude_product_ids": ["1"]}, conn)
            assert st == 200 and [r["product_id"] for r in js["recommendations"]] == ["2", "3"]
            assert js["purchase_history_used"] == "[+1d w1h1] Milk."
            st, js, conn = await _http(port, "POST", "/recommend", {"top_k": 2}, conn)
            assert st == 400
            st, js, conn = await _http(port, "POST", "/recommend", {"user_context": "x", "top_k": 0}, conn)
            assert st == 422
            many = await asyncio.gather(*[_http(port, "POST", "/recommend", {"user_context": "q" * (i + 1), "top_k": 5})
                                          for i in range(40)])
            assert all(m[0] == 200 and len(m[1]["recommendations"]) == 5 for m in many)
            # re-index through the worker: new catalog visible to this front-end afterwards
            new_corpus = {f"n{i}": f"Product: N{i}." for i in range(5)}
            st, js, conn = await _http(port, "POST", "/admin/corpus", {"corpus": new_corpus}, conn)
            assert st == 200 and js == {"status": "ok", "n_products": 5}
            st, js, conn = await _http(port, "POST", "/recommend", {"user_context": "abc", "top_k": 2}, conn)
            assert st == 200 and [r["product_id"] for r in js["recommendations"]] == ["n0", "n1"]
            assert js["recommendations"][0]["product_text"] == "Product: N0."
            st, txt, conn = await _http(port, "GET", "/metrics", None, conn)
            assert st == 200 and "recommendation_requests_total" in txt
            conn[1].close()
            for m in many:
                m[2][1].close()
        finally:
            server.cancel()
            wsrv.close()
            await worker.batcher.stop()
            try:
                await server
            except (asyncio.CancelledError, Exception):
                pass

    asyncio.run(scenario())


def test_two_gpu_workers_behind_one_frontend(tmp_path):
    """serve.py --gpu-workers 2: the front-end's MultiRemoteBatcher spreads requests over both GPU-owner processes, a
    re-index reaches BOTH of them, and a request survives one of them going away (the other answers; 503 only when all
    are gone)."""
    from instacart_next_order_recommendation_amd.api.remote import MultiRemoteBatcher, WorkerUnavailable
    from instacart_next_order_recommendation_amd.api.worker import GpuWorker

    corpus_path = tmp_path / "eval_corpus.json"
    corpus_path.write_text(json.dumps({str(i): f"Product: P{i}." for i in range(1, 21)}))
    new_path = tmp_path / "new_corpus.json"
    new_path.write_text(json.dumps({f"n{i}": f"Product: N{i}." for i in range(6)}))
    d = tempfile.mkdtemp(prefix="icrec_t2_")
    socks = [os.path.join(d, "w0.sock"), os.path.join(d, "w1.sock")]

    class Counting(_StubRecommender):
        calls = {}

        def recommend_batch_timed(self, queries, top_k, excl):
            Counting.calls[id(self)] = Counting.calls.get(id(self), 0) + len(queries)
            return super().recommend_batch_timed(queries, top_k, excl)

    async def scenario():
        workers = [GpuWorker(None, corpus_path, factory=Counting) for _ in socks]
        servers = [await asyncio.start_unix_server(w.handle, path=p) for w, p in zip(workers, socks)]
        seen = []
        mb = MultiRemoteBatcher(socks, on_corpus=seen.append, call_timeout=5.0)
        await mb.start()
        try:
            assert mb.connected
            res = await asyncio.gather(*[mb.submit("q" * (i + 1), 3, None) for i in range(40)])
            assert all([p for p, _ in r[0]] == ["1", "2", "3"] for r in res)
            per_worker = sorted(Counting.calls.values())
            assert len(per_worker) == 2 and per_worker == [20, 20]          # round-robin
            path, n = await mb.reindex(str(new_path))
            assert n == 6 and all(w.recommender.product_ids[0] == "n0" for w in workers)
            r, _ = await mb.submit("abc", 2, None)
            assert [p for p, _ in r] == ["n0", "n1"]
            # one worker goes away: requests keep being answered by the other one
            servers[0].close()
            for w in list(workers[0].writers):
                w.close()
            await asyncio.sleep(0.2)
            for _ in range(6):
                r, _ = await mb.submit("abc", 1, None)
                assert [p for p, _ in r] == ["n0"]
            assert not mb.connected
            servers[1].close()
            for w in list(workers[1].writers):
                w.close()
            await asyncio.sleep(0.2)
            with pytest.raises(WorkerUnavailable):
                await mb.submit("abc", 1, None)
        finally:
            await mb.stop()
            for s in servers:
                s.close()
            for w in workers:
                await w.batcher.stop()

    asyncio.run(scenario())
