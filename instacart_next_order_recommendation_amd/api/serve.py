"""Multi-process server: one GPU-owner process + N HTTP front-ends on one port (SO_REUSEPORT).

    python -m instacart_next_order_recommendation_amd.api.serve --workers 6 --port 8000 [--gpu-workers 2]

Env (as the reference's API): MODEL_DIR, CORPUS_PATH, API_KEY; BATCH_MAX_SIZE, BATCH_MAX_WAIT_MS for the worker.
A single process is still `uvicorn instacart_next_order_recommendation_amd.api.app:app` (reference: uvicorn src.api.main:app).
"""
from __future__ import annotations

import argparse
import asyncio
import multiprocessing as mp
import os
import signal
import tempfile


def _frontend(sock_path: str, host: str, port: int, ready) -> None:
    os.environ["ICREC_GPU_WORKER_SOCKET"] = sock_path
    from .app import app
    from .fastserve import serve
    from .worker import _settle_heap

    def on_ready():
        _settle_heap()  # the corpus texts are loaded by now (lifespan): keep them out of later full collections
        ready.set()

    asyncio.run(serve(app, host, port, reuse_port=True, ready=on_ready))


def _worker(sock_path: str, model_dir: str, corpus_path: str, ready) -> None:
    from .worker import run

    run(sock_path, model_dir, corpus_path, ready.set)


def start(n_frontends: int, host: str, port: int, model_dir: str, corpus_path: str, n_gpu_workers: int = 1):
    """-> (processes, socket path(s)); returns when every process is accepting.  n_gpu_workers > 1: several GPU-owner
    processes on the one GPU (each with its own encoder + index, ~0.4 GB of HBM): a GPU owner is one Python thread and
    tops out at ~20 k requests per second; the front-ends spread their requests over all of them."""
    ctx = mp.get_context("spawn")  # the GPU owner must not be a fork of a process with live threads / HIP state
    tmp = tempfile.mkdtemp(prefix="icrec_srv_")
    socks = [os.path.join(tmp, f"gpu{i}.sock" if n_gpu_workers > 1 else "gpu.sock") for i in range(max(1, n_gpu_workers))]
    procs, wready = [], []
    for sp in socks:
        ev = ctx.Event()
        p = ctx.Process(target=_worker, args=(sp, model_dir, corpus_path, ev), daemon=True)
        p.start()
        procs.append(p)
        wready.append(ev)
    for p, ev in zip(procs, wready):
        while not ev.wait(0.5):  # model load + catalog encode
            if not p.is_alive():
                raise RuntimeError("GPU worker failed to start")
    sock_arg = ",".join(socks)
    events = []
    for _ in range(n_frontends):
        ev = ctx.Event()
        p = ctx.Process(target=_frontend, args=(sock_arg, host, port, ev), daemon=True)
        p.start()
        procs.append(p)
        events.append(ev)
    for ev in events:
        if not ev.wait(120):
            raise RuntimeError("an HTTP front-end failed to start")
    return procs, sock_arg


def main() -> None:
    ap = argparse.ArgumentParser(description="GPU worker + N HTTP front-ends")
    ap.add_argument("--workers", type=int, default=4, help="HTTP front-end processes")
    ap.add_argument("--gpu-workers", type=int, default=1, help="GPU-owner processes on the one GPU (each ~20 k requests/s)")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8000)
    args = ap.parse_args()
    model_dir = os.getenv("MODEL_DIR", "models/two_tower_sbert/final")
    corpus_path = os.getenv("CORPUS_PATH", "processed/p5_mp20_ef0.1/eval_corpus.json")
    procs, _ = start(args.workers, args.host, args.port, model_dir, corpus_path, args.gpu_workers)
    print(f"serving on {args.host}:{args.port} with {args.workers} front-ends + {args.gpu_workers} GPU worker(s)", flush=True)
    signal.signal(signal.SIGTERM, lambda *_: (_ for _ in ()).throw(KeyboardInterrupt()))
    raise SystemExit(supervise(procs))


def supervise(procs, poll_s: float = 0.5) -> int:
    """Block until any child exits (or SIGTERM / Ctrl-C), then stop every other one.  A dead GPU worker must not
    leave N front-ends accepting requests they cannot serve: the whole server exits non-zero and whoever runs it
    (systemd, k8s) starts a fresh set of processes - a process that touched the GPU is never re-exec'd."""
    import time

    code = 0
    try:
        while True:
            dead = [p for p in procs if not p.is_alive()]
            if dead:
                code = next((p.exitcode for p in dead if p.exitcode), 0) or 1
                print(f"child process {dead[0].name} exited ({dead[0].exitcode}); stopping the server", flush=True)
                break
            time.sleep(poll_s)
    except KeyboardInterrupt:
        pass
    for p in procs:
        if p.is_alive():
            p.terminate()
    for p in procs:
        p.join(10)
    return code


if __name__ == "__main__":
    main()
