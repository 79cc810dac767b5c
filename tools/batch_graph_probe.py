#!/usr/bin/env python3
"""Does a hipGraph of the BATCH step (icrec_encode + icrec_search of the bench shape: 1,024 contexts, 49,688 rows) beat launching
its ~35 kernels one by one?    Same buffers, same kernels, 30 timed steps each way, alternated."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from instacart_next_order_recommendation_amd import synthetic as syn
from instacart_next_order_recommendation_amd.encoder import DeviceEncoder
from instacart_next_order_recommendation_amd.search import DeviceIndex

dev = torch.device("cuda:0")
shape = syn.BertShape()
enc = DeviceEncoder(syn.synthetic_bert_weights(shape, seed=0), shape, dev)
ix = DeviceIndex(torch.from_numpy(syn.synthetic_embeddings(49688, 384, seed=1)).to(dev), dev, storage="f32+filter")
ids, cu = syn.synthetic_token_batch(1024, seed=1234)
ids_d, cu_d, mx = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev), int(np.diff(cu).max())
emb = torch.empty((1024, 384), device=dev)
out_i = torch.empty((1024, 20), dtype=torch.int64, device=dev)
out_s = torch.empty((1024, 20), dtype=torch.float32, device=dev)

def step():
    enc.encode_packed(ids_d, cu_d, mx, out=emb)
    ix.search_into(emb, 20, None, None, out_i, out_s)

side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.synchronize()
ref_i = out_i.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        step()
torch.cuda.synchronize()

def timed(fn, n=30):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

with torch.cuda.stream(side):
    for r in range(3):
        a = timed(step); b = timed(g.replay)
        print(f"round {r}: kernel by kernel {a:.4f} ms per step   graph replay {b:.4f} ms per step   same result: {bool(torch.equal(out_i, ref_i))}")
