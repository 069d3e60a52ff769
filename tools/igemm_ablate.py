#!/usr/bin/env python3
"""Time one linear GEMM shape through the C ABI (use with RDM_IGEMM_DBG ablation bits: 1 = no MFMA, 2 = no LDS-DMA requests)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib
ctx = _lib.Context(0); d = ctx.device
shapes = [(131072, 384, 1920), (131072, 768, 384), (131072, 384, 384), (32768, 576, 2880)]
for (M, N, K) in shapes:
    a = [torch.randn(M, K, device=d).bfloat16() for _ in range(3)]; w = torch.randn(N, K, device=d).bfloat16()
    out = [ctx.op_linear(a[i % 3], w) for i in range(3)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for i in range(n): ctx.op_linear(a[i % 3], w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K}: {dt*1e6:.1f} us  {2.0*M*N*K/dt/1e12:.0f} TF", flush=True)
