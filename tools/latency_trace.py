#!/usr/bin/env python3
"""Run N single-query encode+search calls (kernel by kernel, no graph) for rocprofv3 --kernel-trace."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from instacart_next_order_recommendation_amd import synthetic as syn
from instacart_next_order_recommendation_amd.encoder import DeviceEncoder
from instacart_next_order_recommendation_amd.search import DeviceIndex
n_tok = int(sys.argv[1]) if len(sys.argv) > 1 else 99
shape = syn.BertShape()
enc = DeviceEncoder(syn.synthetic_bert_weights(shape, seed=0), shape)
ix = DeviceIndex(syn.synthetic_embeddings(49688, 384, seed=1))
ids, cu = syn.synthetic_token_batch(1, seed=5, mean_len=n_tok, std_len=0, lo=n_tok, hi=n_tok)
ids_d, cu_d = torch.from_numpy(ids).cuda(), torch.from_numpy(cu).cuda()
for _ in range(40):
    e = enc.encode_packed(ids_d, cu_d, n_tok)
    ix.search(e, 20)
torch.cuda.synchronize()
