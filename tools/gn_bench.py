#!/usr/bin/env python3
"""GroupNorm(+SiLU) timing through the C ABI: the whole batch in one call against the same batch in sub-batches (does the second
pass of a sub-batch find its input in the memory-side cache?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib
ctx = _lib.Context(0); d = ctx.device
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (H, C) in ((64, 192), (64, 384), (32, 384), (32, 768), (16, 576), (8, 960)):
    B = 128
    xs = [torch.randn(B, H * H, C, device=d).bfloat16() for _ in range(3)]      # rotate inputs: no reuse across calls
    g = torch.ones(C, device=d); b = torch.zeros(C, device=d)
    it = [0]
    def whole():
        it[0] += 1; ctx.op_groupnorm(xs[it[0] % 3], g, b, 1e-5, True)
    res = [bench(whole)]
    for S in (64, 32, 16):
        def sub():
            it[0] += 1; x = xs[it[0] % 3]
            for s0 in range(0, B, S): ctx.op_groupnorm(x[s0:s0 + S], g, b, 1e-5, True)
        res.append(bench(sub))
    mb = B * H * H * C * 2 / 1e6
    print(f"{H}x{H}x{C}: {mb:.0f} MB  whole {res[0]:.1f} us ({3*mb/res[0]/1e6*1e6/1e6:.2f} TB/s of 3 passes) | sub64 {res[1]:.1f} | sub32 {res[2]:.1f} | sub16 {res[3]:.1f}", flush=True)
