"""Yardstick, not product: how fast does the vendor library (hipBLASLt through torch.matmul) run the
encoder's GEMM shapes in plain f16 / bf16 / f32 on this GPU?  x3 does 3 f16 MFMAs per product, so
library_f16_time * 3 is the time a library-quality kernel would need for the same arithmetic.
Usage: python tools/lib_gemm_yardstick.py [M]"""
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131150
shapes = [("qkv", 384, 1152), ("attn_out", 384, 384), ("ffn_up", 384, 1536), ("ffn_down", 1536, 384)]
dev = torch.device("cuda:0")
for name, K, N in shapes:
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        a = torch.randn(M, K, device=dev, dtype=dt)
        w = torch.randn(N, K, device=dev, dtype=dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        for _ in range(5):
            torch.matmul(a, w.t(), out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.matmul(a, w.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:9s} M={M} K={K:5d} N={N:5d} {str(dt):15s} {us:8.1f} us  {2.0 * M * K * N / us / 1e6:8.1f} TFLOP/s")
