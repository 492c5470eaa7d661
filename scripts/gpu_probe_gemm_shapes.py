#!/usr/bin/env python3
"""hipBLASLt (torch.nn.functional.linear, bf16, bias) on the encoder's GEMM shapes: the vendor's rate for the same work."""
import torch

shapes = [("minilm QKV", 65536, 1152, 384), ("minilm out-proj", 65536, 384, 384), ("minilm FFN-up", 65536, 1536, 384),
          ("minilm FFN-down", 65536, 384, 1536), ("bge QKV", 65536, 2304, 768), ("bge out-proj", 65536, 768, 768),
          ("bge FFN-up", 65536, 3072, 768), ("bge FFN-down", 65536, 768, 3072)]
for name, T, N, K in shapes:
    x = torch.randn(T, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.nn.functional.linear(x, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.nn.functional.linear(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:18s} T={T} N={N} K={K}: {us:7.1f} us  {2.0 * T * N * K / us / 1e6:7.1f} TFLOP/s")
