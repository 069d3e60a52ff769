#!/usr/bin/env python3
"""Exact-kNN scan timing at the OpenImages DB size (20,927,907 x 512 fp16 = 21.4 GB streamed per 64 queries)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20_927_907
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = _lib.Context(0); d = ctx.device
g = torch.Generator(device=d).manual_seed(7)
db = torch.empty((N, 512), device=d, dtype=torch.float16)
for r0 in range(0, N, 1 << 20):
    r1 = min(N, r0 + (1 << 20)); db[r0:r1] = (torch.randn((r1 - r0, 512), device=d, generator=g) * 0.45).half()
ctx.db_load(db); del db; torch.cuda.empty_cache()
q = torch.randn((B, 512), device=d, generator=g) * 0.45
for _ in range(2): ctx.knn(q, k)
torch.cuda.synchronize()
n = 10; t0 = time.perf_counter()
for _ in range(n): idx, sc = ctx.knn(q, k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{os.environ.get('RDM_HIP_LIB','default')}: kNN N={N} B={B} k={k}: {dt*1e3:.3f} ms per call, {N*512*2*((B+63)//64)/dt/1e12:.2f} TB/s of database streamed")
