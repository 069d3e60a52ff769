#!/usr/bin/env python3
"""The UNet's three Upsample convs (nearest 2x + conv3x3) at the benchmark batch: time per call.  RDM_NO_UPS_PHASE=1 times the fused-upsample
halo kernel (nine taps at output resolution) instead of the four-phase form (2 x 2 taps at source resolution).  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RDM_OP_FRAG_CACHE", "1")
import torch
import rdm_amd
from rdm_amd import _lib
ctx = _lib.Context(0); d = ctx.device
tot = 0.0
for (B, H, C) in ((128, 32, 384), (128, 16, 576), (128, 8, 960)):
    x = torch.randn(B, H, H, C, device=d).bfloat16(); w = (torch.randn(C, 3, 3, C, device=d) * (9 * C) ** -0.5).bfloat16(); b = torch.zeros(C, device=d)
    for _ in range(3): ctx.op_conv3x3(x, w, b, ups=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ctx.op_conv3x3(x, w, b, ups=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    tot += dt
    print(f"upsample conv B={B} {H}x{H} -> {2*H}x{2*H} C={C}: {dt*1e3:7.3f} ms  ({2.0*B*4*H*H*C*9*C/dt/1e12:7.1f} TFLOP/s of the nine-tap count)", flush=True)
print(f"sum {tot*1e3:.3f} ms per UNet forward")
