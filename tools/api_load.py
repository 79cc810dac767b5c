#!/usr/bin/env python3
"""In-process load test of the FastAPI /recommend surface (no sockets and no HTTP client library: requests are
raw ASGI calls into the app), so the figure is the ASGI + pydantic + MicroBatcher + GPU ceiling of ONE
server process, not of a network stack or of a Python HTTP client.
N concurrent clients each POST `--requests-per-client` user contexts (top_k 20) against the full synthetic
49,688-product catalog; prints one JSON line with QPS, latency percentiles and the mean micro-batch size.
usage: python tools/api_load.py [--clients 256] [--requests-per-client 20]"""
import argparse, asyncio, json, os, statistics, sys, tempfile, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
ap = argparse.ArgumentParser()
ap.add_argument("--clients", type=int, default=256)
ap.add_argument("--requests-per-client", type=int, default=20)
ap.add_argument("--products", type=int, default=49688)
ap.add_argument("--max-wait-ms", default="2")
args = ap.parse_args()

from instacart_next_order_recommendation_amd import synthetic as syn
from instacart_next_order_recommendation_amd.model_io import write_synthetic_model_dir

tmp = Path(tempfile.mkdtemp(prefix="icrec_load_"))
model_dir = write_synthetic_model_dir(tmp / "model", seed=2)
(tmp / "processed").mkdir()
corpus = tmp / "processed" / "eval_corpus.json"
corpus.write_text(json.dumps(syn.synthetic_catalog(args.products)))
os.environ.update(MODEL_DIR=str(model_dir), CORPUS_PATH=str(corpus), BATCH_MAX_WAIT_MS=args.max_wait_ms)
from instacart_next_order_recommendation_amd.api.app import app  # noqa: E402

ctxs = syn.synthetic_user_contexts(4096, seed=5)


async def asgi_call(method: str, path: str, body: bytes = b""):
    """One request straight into the ASGI app -> (status, body bytes)."""
    done, out, chunks = asyncio.Event(), {}, []
    scope = {"type": "http", "asgi": {"version": "3.0"}, "http_version": "1.1", "method": method, "path": path,
             "raw_path": path.encode(), "query_string": b"", "root_path": "", "scheme": "http",
             "headers": [(b"content-type", b"application/json"), (b"content-length", str(len(body)).encode())],
             "client": ("127.0.0.1", 1), "server": ("icrec", 80), "app": app, "state": {}}
    sent = False

    async def receive():
        nonlocal sent
        if not sent:
            sent = True
            return {"type": "http.request", "body": body, "more_body": False}
        await done.wait()
        return {"type": "http.disconnect"}

    async def send(msg):
        if msg["type"] == "http.response.start":
            out["status"] = msg["status"]
        elif msg["type"] == "http.response.body":
            chunks.append(msg.get("body", b""))
            if not msg.get("more_body"):
                done.set()

    await app(scope, receive, send)
    return out["status"], b"".join(chunks)


async def main():
    lat = []
    bodies = [json.dumps({"user_context": c, "top_k": 20}).encode() for c in ctxs]
    async with app.router.lifespan_context(app):
        async def client(ci):
            for j in range(args.requests_per_client):
                t = time.perf_counter()
                st, body = await asgi_call("POST", "/recommend", bodies[(ci * 131 + j) % len(bodies)])
                assert st == 200, body[:200]
                lat.append(time.perf_counter() - t)
        await asyncio.gather(*[client(i) for i in range(min(args.clients, 8))])  # warm-up (graphs, workspaces)
        lat.clear()
        t0 = time.perf_counter()
        await asyncio.gather(*[client(i) for i in range(args.clients)])
        wall = time.perf_counter() - t0
        m = (await asgi_call("GET", "/metrics"))[1].decode()
    bs_sum = bs_cnt = 0.0
    for line in m.splitlines():
        if line.startswith("recommendation_batch_size_sum"):
            bs_sum = float(line.split()[-1])
        if line.startswith("recommendation_batch_size_count"):
            bs_cnt = float(line.split()[-1])
    lat.sort()
    n = len(lat)
    print(json.dumps({"clients": args.clients, "requests": n, "wall_s": round(wall, 3), "qps": round(n / wall, 1),
                      "p50_ms": round(lat[n // 2] * 1e3, 2), "p95_ms": round(lat[int(n * 0.95)] * 1e3, 2),
                      "p99_ms": round(lat[int(n * 0.99)] * 1e3, 2), "mean_micro_batch": round(bs_sum / max(bs_cnt, 1), 1),
                      "products": args.products, "note": "single Python process, raw ASGI calls, pydantic + JSON per request"}))


asyncio.run(main())
