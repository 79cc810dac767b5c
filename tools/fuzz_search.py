#!/usr/bin/env python3
"""Randomised parity sweep of the search paths against the oracle (bit-exact indices and scores):
random (rows, queries, k, storage, exclusions, duplicates, row_offset) — streaming kernel, three MFMA tile
variants, bf16 rows, filter + verify with and without forced fallbacks.  usage: python tools/fuzz_search.py [n_cases] [seed]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from oracle import oracle as o
from instacart_next_order_recommendation_amd.search import DeviceIndex, merge_topk

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for case in range(n_cases):
    storage = rng.choice(["f32", "bf16", "f32+filter", "bf16+filter"])
    nq = int(rng.choice([1, 2, 3, 5, 8, 9, 31, 33, 64, 65, 100, 129, 256, 300, 513]))
    k = int(rng.choice([1, 2, 5, 10, 20, 32, 33, 64, 100, 116, 117, 128]))
    n = int(rng.choice([1, 7, 31, 32, 33, 127, 128, 129, 255, 256, 257, 1000, 4097, 20000]))
    if nq * n > 4_000_000:
        n = max(1, 4_000_000 // nq)
    P = rng.standard_normal((n, 384)).astype(np.float32) * float(rng.choice([1e-3, 1.0, 50.0]))
    if n > 40 and rng.random() < 0.4:  # blocks of identical and near-identical rows
        d = rng.choice(n, min(n // 2, int(rng.integers(2, 300))), replace=False)
        P[d] = P[d[0]] + (0 if rng.random() < 0.5 else 1e-6) * rng.standard_normal((len(d), 384)).astype(np.float32)
    if rng.random() < 0.1:
        P[rng.integers(0, n)] = 0.0  # a zero row (normalised with the eps clamp)
    q = rng.standard_normal((nq, 384)).astype(np.float32)
    if rng.random() < 0.3:
        q[: max(1, nq // 3)] = P[rng.integers(0, n, max(1, nq // 3))] + 0.02 * q[: max(1, nq // 3)]
    if rng.random() < 0.1:
        q[0] = 0.0
    excl = None
    if rng.random() < 0.6:
        excl = [rng.choice(n, size=int(rng.integers(0, min(n, 60) + 1)), replace=False).tolist() for _ in range(nq)]
    off = int(rng.choice([0, 0, 12345, 4_000_000_000 - n - 1]))
    base = "bf16" if storage.startswith("bf16") else "f32"
    wi, ws = o.search(q, P, k, excl, row_offset=off, storage=base)
    ix = DeviceIndex(P, storage=storage, row_offset=off)
    idx, sc = ix.search(q, k, excl)
    gi, gs = idx.cpu().numpy(), sc.cpu().numpy()
    keys = ix.search_partial(q, k, excl)
    pi, ps = merge_topk(keys.unsqueeze(0), k)
    ok = np.array_equal(gi, wi) and np.array_equal(gs, ws) and np.array_equal(pi.cpu().numpy(), wi) and np.array_equal(ps.cpu().numpy(), ws)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: storage={storage} nq={nq} k={k} n={n} excl={excl is not None} off={off} "
              f"idx_bad={int((gi != wi).sum())} score_bad={int((gs != ws).sum())}", flush=True)
    ix.close()
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
