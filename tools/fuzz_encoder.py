#!/usr/bin/env python3
"""Randomised parity sweep of the encoder against the oracle: random batch compositions (1..200 sequences of
1..256 tokens, incl. the 512-token boundary between the small-M and batch GEMM kernels and all four attention
length buckets), both gemm modes; also checks that every sequence encodes to the same bits alone and in the batch.
usage: python tools/fuzz_encoder.py [n_cases] [seed]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from oracle import oracle as o
from instacart_next_order_recommendation_amd import synthetic as syn
from instacart_next_order_recommendation_amd.encoder import DeviceEncoder

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
shape = syn.BertShape()
w = syn.synthetic_bert_weights(shape, seed=0)
cfg = o.make_cfg(vocab_size=shape.vocab_size, n_normalize=shape.n_normalize)
o.set_threads(o.usable_cpus())
encs = {m: DeviceEncoder(w, shape, "cuda:0", gemm_mode=m) for m in ("f16x3", "f32")}
worst = {m: 0.0 for m in encs}
bad = 0
t0 = time.time()
for case in range(n_cases):
    n = int(rng.choice([1, 2, 3, 7, 20, 60, 200]))
    style = rng.choice(["short", "mixed", "long", "edge"])
    if style == "short": lens = rng.integers(1, 33, n)
    elif style == "long": lens = rng.integers(129, 257, n)
    elif style == "edge": lens = rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256], n)
    else: lens = rng.integers(1, 257, n)
    while lens.sum() > 12000: lens = lens[:-1]
    n = len(lens)
    cu = np.zeros(n + 1, np.int32); cu[1:] = np.cumsum(lens)
    ids = rng.integers(0, shape.vocab_size, int(cu[-1])).astype(np.int32)
    want = o.encode(w, cfg, ids, cu)
    for m, enc in encs.items():
        got = enc.encode_packed(torch.from_numpy(ids).cuda(), torch.from_numpy(cu).cuda(), int(lens.max())).cpu().numpy()
        err = float(np.abs(got - want).max())
        worst[m] = max(worst[m], err)
        s = int(rng.integers(0, n))
        one = enc.encode_packed(torch.from_numpy(ids[cu[s]:cu[s + 1]].copy()).cuda(),
                                torch.tensor([0, int(lens[s])], dtype=torch.int32).cuda(), int(lens[s])).cpu().numpy()[0]
        if err > 5e-6 or not np.array_equal(one, got[s]) or not np.isfinite(got).all():
            bad += 1
            print(f"MISMATCH case {case} mode={m} n={n} tokens={int(cu[-1])} style={style} err={err:.3g} "
                  f"alone_equal={np.array_equal(one, got[s])}", flush=True)
print(f"{n_cases} cases, {bad} mismatches, worst |d emb| {worst}, {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
