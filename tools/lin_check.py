#!/usr/bin/env python3
"""Parity of the one-wave-per-SIMD linear kernel (lin4.hip) against an fp32 reference through the C ABI (GPU box only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib
ctx = _lib.Context(0); d = ctx.device
bad = 0
for (M, N, K, bias, res) in [(49152, 192, 64, 1, 0), (49152, 384, 384, 1, 0), (65536, 384, 384, 1, 1), (131072, 768, 384, 0, 0), (32768 * 2, 576, 576, 1, 1),
                             (49152 + 256, 192, 192, 1, 1), (65536, 384, 1536, 1, 1), (32768 + 128, 768, 128, 1, 1), (98304, 576, 192, 1, 0)]:
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16(); w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, generator=g) if bias else None
    r = torch.randn(M, N, generator=g).bfloat16() if res else None
    out = ctx.op_linear(a.to(d), w.to(d), None if b is None else b.to(d), residual=None if r is None else r.to(d)).float().cpu()
    ref = torch.empty(M, N)
    for s in range(0, M, 16384):
        x = a[s:s + 16384].float() @ w.float().t()
        if b is not None: x += b
        if r is not None: x += r[s:s + 16384].float()
        ref[s:s + 16384] = x
    err = float((out - ref).norm() / ref.norm()); mx = float((out - ref).abs().max())
    ok = err < 5e-3
    bad += not ok
    print(f"lin M={M} N={N} K={K} bias={bias} res={res}: rel l2 {err:.3e} max abs {mx:.3e} {'OK' if ok else 'FAIL'}", flush=True)
    if not ok:
        e = (out - ref).abs()
        rows = (e.amax(dim=1) > 0.1).nonzero().flatten(); cols = (e.amax(dim=0) > 0.1).nonzero().flatten()
        print("  bad rows", len(rows), rows[:16].tolist(), "bad cols", len(cols), cols[:32].tolist())
        nn = torch.isnan(out)
        nr = nn.any(dim=1).nonzero().flatten(); nc = nn.any(dim=0).nonzero().flatten()
        print("  nan entries", int(nn.sum()), "rows", len(nr), nr[:8].tolist(), (nr // 128).unique()[:24].tolist(), "cols", len(nc), nc[:24].tolist())
# GEGLU: W rows stored as [32 x | 32 gates] blocks (packing.py _geglu_perm)
for (M, N, K) in [(49152, 768, 192), (65536, 3072, 384)]:
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16(); w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16(); b = torch.randn(N, generator=g)
    out = ctx.op_linear(a.to(d), w.to(d), b.to(d), act=1).float().cpu()
    ref = torch.empty(M, N // 2)
    wf = w.float().reshape(N // 64, 2, 32, K); bf = b.reshape(N // 64, 2, 32)
    for s0 in range(0, M, 16384):
        y = torch.einsum("mk,qhrk->mqhr", a[s0:s0 + 16384].float(), wf) + bf
        ref[s0:s0 + 16384] = (y[:, :, 0] * torch.nn.functional.gelu(y[:, :, 1])).reshape(-1, N // 2)
    err = float((out - ref).norm() / ref.norm()); mx = float((out - ref).abs().max())
    ok = err < 5e-3
    bad += not ok
    print(f"geglu M={M} N={N} K={K}: rel l2 {err:.3e} max abs {mx:.3e} {'OK' if ok else 'FAIL'}", flush=True)
sys.exit(1 if bad else 0)
