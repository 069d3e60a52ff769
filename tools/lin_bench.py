#!/usr/bin/env python3
"""Timing of single linear GEMM shapes through the C ABI (GPU box only).  usage: lin_bench.py M,N,K[,res] ...   env: RDM_NO_LIN4, RDM_L4_VAR, RDM_LIN4_PROF"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RDM_OP_FRAG_CACHE", "1")
import torch
import rdm_amd
from rdm_amd import _lib
ctx = _lib.Context(0); d = ctx.device
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(",")]
    M, N, K = v[:3]; res = len(v) > 3 and v[3]
    a = torch.randn(M, K, device=d).bfloat16(); w = (torch.randn(N, K, device=d) * K ** -0.5).bfloat16(); b = torch.randn(N, device=d)
    r = torch.randn(M, N, device=d).bfloat16() if res else None
    As = [a] + [a.clone() for _ in range(3)]; it = [0]
    def f():
        it[0] += 1
        ctx.op_linear(As[it[0] % 4], w, b, residual=r)
    t = bench(f)
    by = 2.0 * (M * K + M * N * (2 if res else 1) + N * K)
    print(f"M={M} N={N} K={K} res={int(bool(res))}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF  {by/t/1e12:5.2f} TB/s", flush=True)
