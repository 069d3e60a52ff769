#!/usr/bin/env python3
"""LayerNorm + Linear as two launches vs the folded kernel (lin4.hip <.., LN>), the UNet's six (M, N, K) sites (GPU box only).
env: RDM_LIN4_PROF=1 prints the per-block phase clocks of every lin4 launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RDM_OP_FRAG_CACHE", "1")
import torch
import rdm_amd
from rdm_amd import _lib
from rdm_amd.packing import _geglu_perm
ctx = _lib.Context(0); d = ctx.device
prof = bool(int(os.environ.get("RDM_LIN4_PROF", "0")))
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (M, C) in [(131072, 384), (32768, 576), (8192, 960)]:
    x = (torch.randn(M, C, device=d) * 1.3 + 0.2).bfloat16()
    g, b = torch.rand(C, device=d) + 0.5, torch.randn(C, device=d) * 0.1
    for name, N, act in (("qkv", 3 * C, 0), ("geglu", 8 * C, 1)):
        w = (torch.randn(N, C, device=d) * C ** -0.5).bfloat16(); bias = torch.randn(N, device=d) * 0.1 if act else None
        ln = torch.nn.functional.layer_norm(x.float(), (C,), g, b).bfloat16()
        t_lin = bench(lambda: ctx.op_linear(ln, w, bias, act=act), 3 if prof else 20)
        t_fold = bench(lambda: ctx.op_linear_ln(x, w, bias, g, b, act=act), 3 if prof else 20)
        print(f"M={M} C={C} {name}: linear alone {t_lin*1e6:7.1f} us, folded {t_fold*1e6:7.1f} us (+{(t_fold-t_lin)*1e6:6.1f}); the LayerNorm pass it replaces moves {4.0*M*C/1e6:.0f} MB", flush=True)
