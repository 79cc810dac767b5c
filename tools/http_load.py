#!/usr/bin/env python3
"""HTTP load test of the multi-process /recommend server over real sockets.

Starts the server (1 GPU worker + --frontends HTTP processes on one port, instacart_next_order_recommendation_amd/api/serve.py)
on the synthetic 49,688-product catalog, then --client-procs load-generator processes that together hold --clients
keep-alive connections; each connection POSTs user contexts (top_k 20) back to back for --seconds.  Prints one JSON
line: QPS, latency percentiles, and the per-process split of the box's CPUs.
usage: python tools/http_load.py [--frontends 6] [--client-procs 6] [--clients 1024] [--seconds 10]"""
import argparse, asyncio, json, multiprocessing as mp, os, sys, tempfile, time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def client_proc(port, n_conn, seconds, bodies, seed, out_q, start_evt):
    import random

    rnd = random.Random(seed)
    lat, n_ok, n_err = [], 0, 0

    async def conn_loop(ci):
        nonlocal n_ok, n_err
        r, w = await asyncio.open_connection("127.0.0.1", port)
        i = rnd.randrange(len(bodies))
        deadline = time.perf_counter() + seconds
        while time.perf_counter() < deadline:
            body = bodies[i % len(bodies)]
            i += 1
            t = time.perf_counter()
            w.write(b"POST /recommend HTTP/1.1\r\nhost: l\r\ncontent-type: application/json\r\ncontent-length: %d\r\n\r\n" % len(body) + body)
            head = await r.readuntil(b"\r\n\r\n")
            clen = 0
            for ln in head.split(b"\r\n")[1:]:
                if ln[:15].lower() == b"content-length:":
                    clen = int(ln[15:])
            await r.readexactly(clen)
            if head[9:12] == b"200":
                n_ok += 1
                lat.append(time.perf_counter() - t)
            else:
                n_err += 1
        w.close()

    async def main():
        start_evt.wait()
        await asyncio.gather(*[conn_loop(i) for i in range(n_conn)])

    asyncio.run(main())
    lat.sort()
    out_q.put((n_ok, n_err, lat[:: max(1, len(lat) // 2000)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frontends", type=int, default=6)
    ap.add_argument("--client-procs", type=int, default=6)
    ap.add_argument("--clients", type=int, default=1024)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--products", type=int, default=49688)
    ap.add_argument("--port", type=int, default=18080)
    ap.add_argument("--gpu-workers", type=int, default=1)
    args = ap.parse_args()

    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.api import serve
    from instacart_next_order_recommendation_amd.model_io import write_synthetic_model_dir

    tmp = Path(tempfile.mkdtemp(prefix="icrec_http_"))
    model_dir = write_synthetic_model_dir(tmp / "model", seed=2)
    (tmp / "processed").mkdir()
    corpus = tmp / "processed" / "eval_corpus.json"
    corpus.write_text(json.dumps(syn.synthetic_catalog(args.products)))
    os.environ.update(MODEL_DIR=str(model_dir), CORPUS_PATH=str(corpus))
    t0 = time.time()
    procs, _ = serve.start(args.frontends, "127.0.0.1", args.port, str(model_dir), str(corpus), args.gpu_workers)
    startup_s = time.time() - t0
    bodies = [json.dumps({"user_context": c, "top_k": 20}).encode() for c in syn.synthetic_user_contexts(4096, seed=5)]

    ctx = mp.get_context("spawn")
    q, evt = ctx.Queue(), ctx.Event()
    per = max(1, args.clients // args.client_procs)
    warm = [ctx.Process(target=client_proc, args=(args.port, 8, 2.0, bodies, 99, q, evt))]
    warm[0].start(); evt.set(); warm[0].join(); q.get()   # warm-up: graphs, workspaces, code paths
    evt = ctx.Event()
    clients = [ctx.Process(target=client_proc, args=(args.port, per, args.seconds, bodies, i, q, evt)) for i in range(args.client_procs)]
    for p in clients:
        p.start()
    time.sleep(1.5)  # let every generator import and connect-ready

    def cpu_s(ps):  # user + system CPU seconds of each process (and its threads)
        import psutil
        out = []
        for p_ in ps:
            try:
                t = psutil.Process(p_.pid).cpu_times()
                out.append(t.user + t.system)
            except Exception:  # noqa: BLE001
                out.append(float("nan"))
        return out

    cpu0 = cpu_s(procs), cpu_s(clients)
    t0 = time.perf_counter()
    evt.set()
    res = [q.get() for _ in clients]
    wall = time.perf_counter() - t0
    cpu1 = cpu_s(procs), cpu_s(clients)
    busy = lambda a, b: [round((y - x) / wall, 2) for x, y in zip(a, b)]  # noqa: E731
    ng = args.gpu_workers
    cpu_split = {"gpu_workers": busy(cpu0[0][:ng], cpu1[0][:ng]), "front_ends": busy(cpu0[0][ng:], cpu1[0][ng:]),
                 "load_generators": busy(cpu0[1], cpu1[1])}
    for p in clients:
        p.join()
    for p in procs:
        p.terminate()
    n_ok = sum(r[0] for r in res)
    n_err = sum(r[1] for r in res)
    lat = sorted(x for r in res for x in r[2])
    n = len(lat)
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if quota[0] == "max" else float(quota[0]) / float(quota[1])
    except Exception:
        quota = None
    print(json.dumps({"gpu_workers": args.gpu_workers, "frontends": args.frontends, "client_procs": args.client_procs, "connections": per * args.client_procs,
                      "seconds": round(wall, 2), "requests_ok": n_ok, "requests_failed": n_err, "qps": round(n_ok / wall, 1),
                      "p50_ms": round(lat[n // 2] * 1e3, 2), "p95_ms": round(lat[int(n * 0.95)] * 1e3, 2),
                      "p99_ms": round(lat[int(n * 0.99)] * 1e3, 2), "products": args.products, "server_startup_s": round(startup_s, 1),
                      "cpu_quota": quota, "os_cpu_count": os.cpu_count(), "cpus_busy_per_process": cpu_split,
                      "note": "real TCP sockets on loopback, HTTP/1.1 keep-alive; 1 GPU-owner process + N FastAPI front-ends "
                              "(asyncio HTTP server) + M load-generator processes share the box's CPU quota"}))


if __name__ == "__main__":
    main()
