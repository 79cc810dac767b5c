#!/usr/bin/env python3
"""Where should a batch leave the latency-form kernels for the batch kernels?  Encode time (HIP events, median of 20) of batches
of n sequences of ~128 tokens through an encoder created with ICREC_SMALL_M=512 (round 3's bound: above it the layer kernel,
one 64-token workgroup per CU) and one with ICREC_SMALL_M=1048576 (always the latency form).  usage: python tools/small_m_sweep.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from instacart_next_order_recommendation_amd import synthetic as syn
from instacart_next_order_recommendation_amd.encoder import DeviceEncoder

shape = syn.BertShape()
w = syn.synthetic_bert_weights(shape, seed=0)
encs = {}
for name, v in (("batch>512", "512"), ("latency-form", "1048576")):
    os.environ["ICREC_SMALL_M"] = v
    encs[name] = DeviceEncoder(w, shape)
del os.environ["ICREC_SMALL_M"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n in (4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256):
    ids, cu = syn.synthetic_token_batch(n, seed=100 + n)
    i_d, c_d, mx = torch.from_numpy(ids).cuda(), torch.from_numpy(cu).cuda(), int(np.diff(cu).max())
    row = [f"{n:4d} seqs {int(cu[-1]):6d} tokens"]
    outs = []
    for name, enc in encs.items():
        for _ in range(3):
            out = enc.encode_packed(i_d, c_d, mx)
        ts = []
        for _ in range(20):
            e0.record(); out = enc.encode_packed(i_d, c_d, mx); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        outs.append(out.cpu().numpy())
        row.append(f"{name} {np.median(ts):7.3f} ms")
    row.append("bits equal" if np.array_equal(outs[0], outs[1]) else "BITS DIFFER")
    print("  ".join(row), flush=True)
