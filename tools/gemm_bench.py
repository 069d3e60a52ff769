#!/usr/bin/env python3
"""Micro-benchmark of the igemm kernel family through the C ABI (GPU box only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib

ctx = _lib.Context(0)
d = ctx.device
def bench(fn, flops, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return dt * 1e3, flops / dt / 1e12

shapes = [("lin 64^2 proj", 524288, 192, 192), ("lin 32^2 qk", 131072, 768, 384), ("lin ff1 geglu-size", 131072, 3072, 384),
          ("lin big", 65536, 960, 960), ("lin 8192^2-ish", 8192, 7680, 960), ("lin sq 4096", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=d).bfloat16(); w = torch.randn(N, K, device=d).bfloat16()
    ms, tf = bench(lambda: ctx.op_linear(a, w), 2.0 * M * N * K)
    print(f"{name:24s} M={M} N={N} K={K}: {ms:8.3f} ms  {tf:8.1f} TF", flush=True)
convs = [("conv 64^2 192", 128, 64, 64, 192, 192), ("conv 32^2 384", 128, 32, 32, 384, 384), ("conv 16^2 576", 128, 16, 16, 576, 576),
         ("conv 8^2 960", 128, 8, 8, 960, 960), ("conv 8^2 1920->960", 128, 8, 8, 1920, 960)]
for name, B, H, W, C, N in convs:
    x = torch.randn(B, H, W, C, device=d).bfloat16(); w = torch.randn(N, 3, 3, C, device=d).bfloat16(); b = torch.zeros(N, device=d)
    ms, tf = bench(lambda: ctx.op_conv3x3(x, w, b), 2.0 * B * H * W * N * 9 * C)
    print(f"{name:24s} B={B} {H}x{W} C={C} N={N}: {ms:8.3f} ms  {tf:8.1f} TF", flush=True)
