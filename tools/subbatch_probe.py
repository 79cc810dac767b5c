#!/usr/bin/env python3
"""Experiment: encode 1,024 contexts as ONE call vs as n sub-batches run one after the other on ONE stream (working set
per sub-batch small enough for the 256 MB Infinity Cache: Q/K/V written by the layer kernel are read back by the next
attention launch before they leave it), and as n sub-batches alternating over two streams."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from instacart_next_order_recommendation_amd import synthetic as syn
from instacart_next_order_recommendation_amd.encoder import DeviceEncoder
shape = syn.BertShape()
w = syn.synthetic_bert_weights(shape, seed=0)
ids, cu = syn.synthetic_token_batch(1024, seed=1234)
dev = torch.device("cuda:0")
def run(nsplit, nstreams=1, reps=20):
    encs = [DeviceEncoder(w, shape, dev) for _ in range(nstreams)]
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    parts = []
    per = 1024 // nsplit
    for p in range(nsplit):
        a, b = p * per, (p + 1) * per
        t0, t1 = int(cu[a]), int(cu[b])
        parts.append((torch.from_numpy(ids[t0:t1]).to(dev), torch.from_numpy((cu[a:b + 1] - t0).astype(np.int32)).to(dev), int(np.diff(cu[a:b + 1]).max()),
                      torch.empty((per, 384), device=dev)))
    def step():
        for j, (i, c, m, o) in enumerate(parts):
            with torch.cuda.stream(streams[j % nstreams]):
                encs[j % nstreams].encode_packed(i, c, m, out=o)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print(f"{nsplit} sub-batch(es) over {nstreams} stream(s): {ms:.3f} ms per 1024 contexts -> {1024 / ms * 1e3:.0f} QPS (encode only)", flush=True)
for rnd in range(2):
    for n, s in ((1, 1), (2, 1), (4, 1), (8, 1), (2, 2), (4, 2), (8, 2)):
        run(n, s)
