#!/usr/bin/env python3
"""Bulk neighbour search timing: B queries x N rows x 512, k neighbours (defaults: 10 000 x 1 000 003, k = 20)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_003
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = _lib.Context(0); d = ctx.device
g = torch.Generator(device=d).manual_seed(7)
db = torch.empty((N, 512), device=d, dtype=torch.float16)
for r0 in range(0, N, 1 << 20):
    r1 = min(N, r0 + (1 << 20)); db[r0:r1] = (torch.randn((r1 - r0, 512), device=d, generator=g) * 0.45).half()
ctx.db_load(db); del db; torch.cuda.empty_cache()
q = torch.randn((B, 512), device=d, generator=g) * 0.45
ctx.knn(q[:512], k); torch.cuda.synchronize()
t0 = time.perf_counter(); idx, sc = ctx.knn(q, k); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"bulk kNN B={B} N={N} k={k}: {dt*1e3:.2f} ms, {4.0*B*N*512/dt/1e12:.0f} TFLOP/s (hi+lo MFMA work), fallback={ctx.knn_last_fallback()}")
