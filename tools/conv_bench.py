#!/usr/bin/env python3
"""3x3-conv kernel bench over the UNet's conv shapes through the C ABI (GPU box only), plus multi-tile parity checks.
usage: conv_bench.py [check] [bench]      env: RDM_NO_HALO4=1 -> the 8-wave ping-pong kernel, RDM_HALO_PROF=1 -> cycle breakdown"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
what = sys.argv[1:] or ["check"]
if "bench" in what: os.environ.setdefault("RDM_OP_FRAG_CACHE", "1")     # constant weights: keep the fragment-ordered copy between calls
import torch
import torch.nn.functional as F
import rdm_amd
from rdm_amd import _lib

ctx = _lib.Context(0)
d = ctx.device

def ref_conv(x, w, b, ups=False):
    xx = x.float().permute(0, 3, 1, 2)
    if ups: xx = F.interpolate(xx, scale_factor=2, mode="nearest")
    return F.conv2d(xx, w.float().permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)

if "check" in what:
    # many tiles per block (persistent walk, cross-tile prefetch), every resolution, residual + time row, upsample, split-K
    for (B, H, C, N, ups) in [(40, 64, 64, 192, 0), (136, 32, 128, 192, 0), (72, 16, 192, 384, 0), (520, 8, 64, 192, 0), (36, 8, 320, 192, 0),
                              (24, 32, 64, 192, 1), (12, 64, 64, 128, 0), (64, 16, 64, 192, 1)]:
        g = torch.Generator().manual_seed(B * 131 + H)
        x = torch.randn(B, H, H, C, generator=g).bfloat16()
        w = (torch.randn(N, 3, 3, C, generator=g) * (9 * C) ** -0.5).bfloat16()
        b = torch.randn(N, generator=g) * 0.1
        Ho = H * 2 if ups else H
        temb = torch.randn(B, N, generator=g); res = torch.randn(B, Ho, Ho, N, generator=g).bfloat16()
        ref = ref_conv(x, w, b, bool(ups)) + temb[:, None, None, :] + res.float()
        out = ctx.op_conv3x3(x.to(d), w.to(d), b.to(d), rowvec=temb.to(d), residual=res.to(d), ups=ups).float().cpu()
        err = float((out - ref).norm() / ref.norm()); mx = float((out - ref).abs().max())
        print(f"check B={B} {H}x{H} C={C} N={N} ups={ups}: rel l2 {err:.3e} max abs {mx:.3e} {'OK' if err < 6e-3 else 'FAIL'}", flush=True)

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

if "bench" in what:
    # (B, H, Cin, N, residual+temb): the UNet's conv sites at the benchmark batch (B' = 128; 64 inside the shared guidance prefix)
    shapes = [(64, 64, 192, 192, 1), (128, 64, 192, 192, 1), (128, 64, 384, 192, 0), (128, 64, 576, 192, 0), (128, 64, 384, 384, 0),
              (128, 32, 384, 384, 1), (128, 32, 768, 384, 0), (128, 32, 960, 384, 0), (128, 32, 576, 576, 0),
              (128, 16, 576, 576, 1), (128, 16, 960, 576, 0), (128, 16, 1536, 576, 0), (128, 16, 960, 960, 0),
              (128, 8, 960, 960, 1), (128, 8, 1920, 960, 0), (128, 8, 1536, 960, 0)]
    tot_t = tot_f = 0.0
    for (B, H, C, N, r) in shapes:
        x = torch.randn(B, H, H, C, device=d).bfloat16(); w = (torch.randn(N, 3, 3, C, device=d) * (9 * C) ** -0.5).bfloat16()
        b = torch.zeros(N, device=d)
        temb = torch.randn(B, N, device=d) if r else None
        res = torch.randn(B, H, H, N, device=d).bfloat16() if r else None
        dt = bench(lambda: ctx.op_conv3x3(x, w, b, rowvec=temb, residual=res))
        fl = 2.0 * B * H * H * N * 9 * C
        tot_t += dt; tot_f += fl
        print(f"conv B={B} {H}x{H} C={C}->{N} res={r}: {dt * 1e3:8.3f} ms {fl / dt / 1e12:8.1f} TF", flush=True)
    print(f"TOTAL {tot_t * 1e3:.3f} ms  {tot_f / tot_t / 1e12:.1f} TF")
