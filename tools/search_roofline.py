#!/usr/bin/env python3
"""Search-kernel roofline sweep (similarity is HBM/cache-bound at small Q,
MFMA-bound at large Q).  For each (N, Q): icrec_search top-20, timed with the library's own
hipEvent timers; algorithmic bytes = N*384*4 (catalog read once per query TILE pass is the kernel's
actual behaviour: reported separately), algorithmic FLOPs = 2*Q*N*384.
usage: python tools/search_roofline.py [--rows 49688 2000000]"""
import argparse, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from instacart_next_order_recommendation_amd import _native
from instacart_next_order_recommendation_amd.search import DeviceIndex

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="+", default=[49688, 2_000_000])
ap.add_argument("--storage", default="f32", choices=["f32", "bf16", "f32+filter", "bf16+filter"])
ap.add_argument("--queries", type=int, nargs="+", default=[1, 8, 32, 64, 256, 1024])
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
centres = torch.randn(200, 384, device=dev, generator=g)
out = []
for n in args.rows:
    rows = torch.empty((n, 384), device=dev)
    for s in range(0, n, 1 << 18):
        m = min(1 << 18, n - s)
        rows[s:s + m] = centres[torch.randint(0, 200, (m,), device=dev, generator=g)] + 0.35 * torch.randn(m, 384, device=dev, generator=g)
    ix = DeviceIndex(rows, dev, storage=args.storage)
    esz = 2 if args.storage.startswith("bf16") else 4
    del rows
    for q in args.queries:
        qv = centres[torch.randint(0, 200, (q,), device=dev, generator=g)] + 0.35 * torch.randn(q, 384, device=dev, generator=g)
        for _ in range(3):
            ix.search(qv, 20)
        torch.cuda.synchronize()
        _native.timing_reset(); _native.timing_enable(True)
        reps = 20 if n * q < 5e9 else 5
        for _ in range(reps):
            ix.search(qv, 20)
        torch.cuda.synchronize()
        _native.timing_enable(False)
        k_ms, _ = _native.timing_query(0)
        t_ms, _ = _native.timing_query(3)
        bn = 128 if q > 64 else (64 if q > 32 else 32)
        passes = -(-q // bn)
        rec = {"rows": n, "storage": args.storage, "queries": q, "kernel_ms": round(k_ms, 4), "search_call_ms": round(t_ms, 4),
               "catalog_GBps_once": round(n * 384 * esz / k_ms / 1e6, 1),
               "catalog_GBps_per_query_tile_pass": round(passes * n * 384 * esz / k_ms / 1e6, 1),
               "TFLOPs_algorithmic": round(2.0 * q * n * 384 / k_ms / 1e9, 2),
               "TFLOPs_issued_padded": round(2.0 * passes * bn * n * 384 / k_ms / 1e9, 2), "qps": round(q / t_ms * 1e3)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    ix.close()
