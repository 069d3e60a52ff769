#!/usr/bin/env python3
"""Per-shape timing of the UNet's linear-class GEMMs at B'=128 (GPU box only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib
ctx = _lib.Context(0); d = ctx.device
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
Bp = 128
rows = []
total = 0.0
for (hw, C, nst) in ((1024, 384, 5), (256, 576, 5), (64, 960, 6)):
    M = Bp * hw
    shapes = [("proj_in", C, C, True, False, 0), ("qk", 2 * C, C, False, False, 0), ("o1", C, C, True, True, 0), ("q2", C, C, False, False, 0),
              ("o2", C, C, True, True, 0), ("ff1", 8 * C, C, True, False, 1), ("ff2", C, 4 * C, True, True, 0), ("proj_out", C, C, True, True, 0)]
    for name, N, K, bias, res, act in shapes:
        a = torch.randn(M, K, device=d).bfloat16(); w = torch.randn(N, K, device=d).bfloat16()
        b = torch.randn(N, device=d) if bias else None
        No = N // 2 if act == 1 else N
        r = torch.randn(M, No, device=d).bfloat16() if res else None
        # rotate among 4 copies of A to defeat the 256 MB infinity cache
        As = [a] + [a.clone() for _ in range(3)]
        it = [0]
        def f():
            it[0] += 1
            ctx.op_linear(As[it[0] % 4], w, b, residual=r, act=act)
        t = bench(f)
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + M * No * (2 if res else 1) + N * K)
        print(f"hw={hw:5d} C={C:4d} {name:9s} M={M:7d} N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF  {by/t/1e12:5.2f} TB/s  (x{nst})", flush=True)
        total += t * nst
print(f"sum per forward: {total*1e3:.2f} ms")
