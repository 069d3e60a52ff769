#!/usr/bin/env python3
"""Round 6, verdict item 2: the fused feed-forward kernel (csrc/ffn.hip, rdm_op_ffn_fused) against the two kernels it would replace (lin4 GEGLU +
lin4 ff.net.2 x proj_out) at the 32 x 32 level of a guided batch of 64: M = 131072 rows, C = 384.  GPU box only.  Prints parity (rel L2 against an
fp32 torch reference on the bf16-rounded operands) and microseconds per call.  usage: ffn_bench.py [M=131072]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RDM_OP_FRAG_CACHE", "1")
import torch, torch.nn.functional as F
import rdm_amd
from rdm_amd import _lib
from rdm_amd.packing import _geglu_perm
torch.set_grad_enabled(False)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
C = 384
ctx = _lib.Context(0); d = ctx.device
g = torch.Generator(device=d).manual_seed(1)
R = lambda *s, sc=1.0: (torch.randn(*s, device=d, generator=g) * sc)
bf = lambda t: t.to(torch.bfloat16)
l3, t2, xin = bf(R(M, C)), bf(R(M, C)), bf(R(M, C))
w1, b1 = bf(R(8 * C, C, sc=C ** -0.5)), R(8 * C, sc=0.3)
wf, bfb = bf(R(C, 5 * C, sc=(5 * C) ** -0.5)), R(C, sc=0.3)
perm = torch.as_tensor(_geglu_perm(8 * C), device=d)
w1p, b1p = w1[perm].contiguous(), b1[perm].contiguous()
# parity on the first 1024 rows
n = 1024
pp = l3[:n].float() @ w1.float().t() + b1
x, gate = pp.chunk(2, dim=-1)
ff = bf(x * F.gelu(gate)).float()
ref = bf(torch.cat([ff, t2[:n].float()], dim=1) @ wf.float().t() + bfb + xin[:n].float()).float()
out = ctx.op_ffn_fused(l3, t2, xin, w1p, b1p, wf, bfb)
torch.cuda.synchronize()
rel = lambda a, b: float((a.float() - b).norm() / b.norm())
print(f"fused kernel vs fp32 reference on bf16 operands, first {n} rows: rel L2 {rel(out[:n], ref):.3e}; finite: {bool(torch.isfinite(out.float()).all())}")
hid = ctx.op_linear(l3, w1p, b1p, act=_lib.ACT_GEGLU)
cat = torch.cat([hid, t2], dim=1).contiguous()
pair = ctx.op_linear(cat, wf, bfb, residual=xin)
torch.cuda.synchronize()
print(f"the two-kernel path vs the same reference: rel L2 {rel(pair[:n], ref):.3e}; fused vs two-kernel path (all rows): {rel(out, pair.float()):.3e}")
def bench(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
t_f = bench(lambda: ctx.op_ffn_fused(l3, t2, xin, w1p, b1p, wf, bfb))
t_g = bench(lambda: ctx.op_linear(l3, w1p, b1p, act=_lib.ACT_GEGLU))
t_o = bench(lambda: ctx.op_linear(cat, wf, bfb, residual=xin))
fl = 2.0 * M * (8 * C * C + 5 * C * C)
print(f"M = {M}, C = {C}: fused {t_f:.1f} us ({fl / t_f / 1e6:.0f} TFLOP/s) | GEGLU {t_g:.1f} us + ff2 x proj_out {t_o:.1f} us = {t_g + t_o:.1f} us ({fl / (t_g + t_o) / 1e6:.0f} TFLOP/s)", flush=True)
