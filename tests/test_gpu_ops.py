"""GPU parity of the individual HIP kernels (called through the C ABI) against plain PyTorch fp32
references of the same op on the same bf16-rounded inputs.

Tolerance (bf16 storage, fp32 accumulation): |out - ref| <= 2^-7 * max|ref| elementwise — one bf16
output rounding (2^-9 relative) plus accumulation-order noise; stated per test.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import bf16_round

pytestmark = pytest.mark.gpu
TOL = 2.0 ** -7


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _close(out, ref, tol=TOL, what=""):
    out, ref = out.float().cpu(), ref.float().cpu()
    err = (out - ref).abs().max().item()
    bound = tol * ref.abs().max().item() + 1e-6
    assert err <= bound, f"{what}: max err {err:.4e} > {bound:.4e}"


@pytest.mark.parametrize("M,N,K", [(300, 192, 192), (128, 384, 768), (1000, 128, 64), (2, 768, 192), (257, 960, 576),
                                   (512, 64, 128)])
def test_linear(ctx, M, N, K):
    a, w, b = bf16_round(_rand((M, K), 1)), bf16_round(_rand((N, K), 2, K ** -0.5)), _rand((N,), 3, 0.1)
    ref = a @ w.t() + b
    d = ctx.device
    out = ctx.op_linear(a.to(d, torch.bfloat16), w.to(d, torch.bfloat16), b.to(d))
    _close(out, ref, what="linear")
    outf = ctx.op_linear(a.to(d, torch.bfloat16), w.to(d, torch.bfloat16), b.to(d), out_f32=True)
    _close(outf, ref, tol=2 ** -10, what="linear f32 out")


def test_linear_residual_act_alpha(ctx):
    from rdm_amd import _lib
    M, N, K = 384, 192, 256
    d = ctx.device
    a, w, b, r = bf16_round(_rand((M, K), 4)), bf16_round(_rand((N, K), 5, K ** -0.5)), _rand((N,), 6, 0.1), bf16_round(_rand((M, N), 7))
    ab, wb = a.to(d, torch.bfloat16), w.to(d, torch.bfloat16)
    _close(ctx.op_linear(ab, wb, b.to(d), residual=r.to(d, torch.bfloat16)), a @ w.t() + b + r, what="residual")
    _close(ctx.op_linear(ab, wb, b.to(d), act=_lib.ACT_SILU), F.silu(a @ w.t() + b), what="silu")
    y = a @ w.t() + b
    _close(ctx.op_linear(ab, wb, b.to(d), act=_lib.ACT_QUICKGELU), y * torch.sigmoid(1.702 * y), what="quickgelu")
    _close(ctx.op_linear(ab, wb, None, alpha=0.125, out_f32=True), 0.125 * (a @ w.t()), tol=2 ** -10, what="alpha")


def test_linear_geglu(ctx):
    from rdm_amd import _lib
    from rdm_amd.packing import _geglu_perm
    M, C = 200, 128
    d = ctx.device
    a, w, b = bf16_round(_rand((M, C), 8)), bf16_round(_rand((8 * C, C), 9, C ** -0.5)), _rand((8 * C,), 10, 0.1)
    p = a @ w.t() + b
    x, g = p.chunk(2, dim=-1)
    ref = x * F.gelu(g)
    perm = _geglu_perm(8 * C)
    out = ctx.op_linear(a.to(d, torch.bfloat16), w[perm].contiguous().to(d, torch.bfloat16), b[perm].contiguous().to(d), act=_lib.ACT_GEGLU)
    assert out.shape == (M, 4 * C)
    _close(out, ref, what="geglu")


# Big-M projections: the one-wave-per-SIMD kernel (lin4.hip) takes them when M is a multiple of 128 / 256 and there are >= 192 tiles --
# both wave arrangements (N % 384 == 0: 1 x 4, else 2 x 2), one or two tiles per block (dead look-ahead cursors), a one-slice K,
# residual, no bias, and the GEGLU read-out.  Each case runs twice: the first version of the kernel failed intermittently.
@pytest.mark.parametrize("M,N,K,bias,res", [(49152, 192, 64, 1, 0), (49152, 384, 384, 1, 0), (65536, 384, 384, 1, 1), (32768, 768, 128, 0, 1),
                                            (49408, 192, 192, 1, 1), (49152, 576, 192, 1, 0), (8192, 960, 960, 1, 1)])
def test_linear_big_m(ctx, M, N, K, bias, res):
    d = ctx.device
    a, w = bf16_round(_rand((M, K), 21)), bf16_round(_rand((N, K), 22, K ** -0.5))
    b = _rand((N,), 23, 0.5) if bias else None
    r = bf16_round(_rand((M, N), 24)) if res else None
    ref = a @ w.t()
    if bias: ref = ref + b
    if res: ref = ref + r
    ab, wb = a.to(d, torch.bfloat16), w.to(d, torch.bfloat16)
    for rep in range(2):
        out = ctx.op_linear(ab, wb, None if b is None else b.to(d), residual=None if r is None else r.to(d, torch.bfloat16))
        assert torch.isfinite(out).all()
        _close(out, ref, what=f"big-M linear (run {rep})")


# A per-row-group additive vector (Ops::linear's rowvec; the executor: attn1.to_out over a guided batch, the unconditional half's start
# values carry attn2.to_out's bias): two groups through the one-wave-per-SIMD kernel (both wave arrangements, tiles on either side of
# the split, a split that leaves the second group shorter), several groups / unaligned groups through the generic GEMM.
@pytest.mark.parametrize("M,N,K,rows,res", [(65536, 384, 384, 32768, 1), (49152, 576, 576, 24576, 1), (49152, 384, 384, 32768, 0),
                                            (8192, 960, 960, 4096, 1), (1000, 192, 128, 100, 0), (4096, 384, 384, 1024, 1)])
def test_linear_rowvec(ctx, M, N, K, rows, res):
    d = ctx.device
    G = (M + rows - 1) // rows
    a, w = bf16_round(_rand((M, K), 61)), bf16_round(_rand((N, K), 62, K ** -0.5))
    b, rv = _rand((N,), 63, 0.5), _rand((G, N), 64, 0.7)
    r = bf16_round(_rand((M, N), 65)) if res else None
    ref = a @ w.t() + b + rv.repeat_interleave(rows, dim=0)[:M]
    if res: ref = ref + r
    for rep in range(2):
        out = ctx.op_linear_rowvec(a.to(d, torch.bfloat16), w.to(d, torch.bfloat16), b.to(d), rv.to(d).contiguous(), rows,
                                   residual=None if r is None else r.to(d, torch.bfloat16))
        assert torch.isfinite(out).all()
        _close(out, ref, what=f"linear with a row-group vector (run {rep})")


def test_linear_mid_size_gemm(tmp_path):
    """mgemm.hip (LDS-staged 64 x 64 / 128 x 64 tiles: the RARM decode step's plain projections from 1536 sequences on) on its own, through
    rdm_op_linear in a child process with RDM_MGEMM_ANY=64 (dev switch: plain ops of >= 64 rows take it): ragged row counts, fp32 and bf16
    outputs, bias, bf16 residual, SiLU / QuickGELU, both tile heights and two ring depths -- against fp32 products of the same bf16 operands
    (bound: the fp32-accumulation level for fp32 outputs, one bf16 rounding for bf16 outputs)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {os.path.dirname(here)!r}); sys.path.insert(0, {here!r})\n"
        "import rdm_amd\nfrom rdm_amd import _lib\nfrom _util import bf16_round\n"
        "torch.set_grad_enabled(False)\nctx = _lib.Context(0); d = ctx.device\n"
        "def rnd(shape, seed, scale=1.0):\n"
        "    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))\n"
        "worst = 0.0\n"
        "for (M, N, K, act, res, f32) in [(1064, 768, 768, 0, 1, 1), (2048, 2304, 768, 0, 0, 0), (1000, 192, 3072, 3, 1, 0), (130, 64, 128, 2, 0, 1), (4096, 768, 256, 0, 1, 0)]:\n"
        "    a, w, b = bf16_round(rnd((M, K), 1)), bf16_round(rnd((N, K), 2, K ** -0.5)), rnd((N,), 3, 0.3)\n"
        "    r = bf16_round(rnd((M, N), 4)) if res else None\n"
        "    y = a.double() @ w.double().t() + b.double()\n"
        "    if act == 3: y = y * torch.sigmoid(y)\n"
        "    if act == 2: y = y * torch.sigmoid(1.702 * y)\n"
        "    if res: y = y + r.double()\n"
        "    out = ctx.op_linear(a.to(d, torch.bfloat16), w.to(d, torch.bfloat16), b.to(d), residual=None if r is None else r.to(d, torch.bfloat16), act=act, out_f32=bool(f32))\n"
        "    assert out.shape == (M, N) and bool(torch.isfinite(out).all())\n"
        "    e = ((out.double().cpu() - y).abs().max() / y.abs().max()).item()\n"
        "    print('mgemm', M, N, K, 'act', act, 'res', res, 'f32' if f32 else 'bf16', 'max err / max |ref|: %.3e' % e)\n"
        "    assert e <= (2e-5 if f32 else 6e-3), (M, N, K, e)\n"
        "    worst = max(worst, e)\n"
        "print('OK', worst)\n")
    for extra in ({}, {"RDM_MGEMM_NS": "3"}, {"RDM_MGEMM_NS": "2"}, {"RDM_MGEMM_BM": "128", "RDM_MGEMM_NS": "3"}, {"RDM_MGEMM_BN": "128"}, {"RDM_MGEMM_BN": "128", "RDM_MGEMM_BM": "128"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RDM_MGEMM_ANY="64", **extra), capture_output=True, text=True, timeout=600)
        print(r.stdout[-1500:])
        assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


@pytest.mark.parametrize("M,C", [(49152, 192), (32768, 384)])
def test_linear_geglu_big_m(ctx, M, C):
    from rdm_amd import _lib
    from rdm_amd.packing import _geglu_perm
    d = ctx.device
    a, w, b = bf16_round(_rand((M, C), 25)), bf16_round(_rand((8 * C, C), 26, C ** -0.5)), _rand((8 * C,), 27, 0.3)
    x, g = (a @ w.t() + b).chunk(2, dim=-1)
    ref = x * F.gelu(g)
    perm = _geglu_perm(8 * C)
    ab, wp, bp = a.to(d, torch.bfloat16), w[perm].contiguous().to(d, torch.bfloat16), b[perm].contiguous().to(d)
    for rep in range(2):
        out = ctx.op_linear(ab, wp, bp, act=_lib.ACT_GEGLU)
        assert out.shape == (M, 4 * C) and torch.isfinite(out).all()
        _close(out, ref, what=f"big-M geglu (run {rep})")


# LayerNorm folded into the big-M projection (lin4.hip <.., LN>): norm1 -> q | k | v and norm3 -> GEGLU of BasicTransformerBlock
# (rdm/modules/attention.py:147-168).  The reference is LayerNorm in fp32 on the bf16-rounded rows followed by the fp32 GEMM; rows carry a
# per-row offset and scale so that mean and variance differ from row to row (a wrong row <-> statistic pairing cannot pass).  Both wave
# arrangements, one / several tiles per block, the UNet's three widths, a few-tile case (M = 8192) and a 2-slice K.
@pytest.mark.parametrize("M,N,K", [(32768, 1152, 384), (16384, 1728, 576), (8192, 2880, 960), (49152, 384, 128), (33024, 192, 192)])
def test_linear_layernorm_folded(ctx, M, N, K):
    d = ctx.device
    x = _rand((M, K), 31) * (0.5 + _rand((M, 1), 32).abs() * 2.0) + _rand((M, 1), 33) * 1.5
    x = bf16_round(x)
    w = bf16_round(_rand((N, K), 34, K ** -0.5))
    gamma, beta, b = 1.0 + _rand((K,), 35, 0.3), _rand((K,), 36, 0.3), _rand((N,), 37, 0.5)
    ref = torch.empty(M, N)
    for s0 in range(0, M, 8192):
        ref[s0:s0 + 8192] = F.layer_norm(x[s0:s0 + 8192], (K,), gamma, beta, 1e-5) @ w.t() + b
    xb, wb = x.to(d, torch.bfloat16), w.to(d, torch.bfloat16)
    for rep in range(2):
        out = ctx.op_linear_ln(xb, wb, b.to(d), gamma.to(d), beta.to(d))
        assert torch.isfinite(out).all()
        _close(out, ref, what=f"LayerNorm-folded linear (run {rep})")
    out = ctx.op_linear_ln(xb, wb, None, gamma.to(d), beta.to(d))       # q | k | v has no bias
    _close(out, ref - b, what="LayerNorm-folded linear, no bias")


@pytest.mark.parametrize("M,C", [(32768, 384), (16384, 576), (8192, 960)])
def test_linear_geglu_layernorm_folded(ctx, M, C):
    from rdm_amd import _lib
    from rdm_amd.packing import _geglu_perm
    d = ctx.device
    x = bf16_round(_rand((M, C), 41) * (0.5 + _rand((M, 1), 42).abs()) + _rand((M, 1), 43))
    w, b = bf16_round(_rand((8 * C, C), 44, C ** -0.5)), _rand((8 * C,), 45, 0.3)
    gamma, beta = 1.0 + _rand((C,), 46, 0.3), _rand((C,), 47, 0.3)
    ref = torch.empty(M, 4 * C)
    for s0 in range(0, M, 4096):
        h, g = (F.layer_norm(x[s0:s0 + 4096], (C,), gamma, beta, 1e-5) @ w.t() + b).chunk(2, dim=-1)
        ref[s0:s0 + 4096] = h * F.gelu(g)
    perm = _geglu_perm(8 * C)
    xb, wp, bp = x.to(d, torch.bfloat16), w[perm].contiguous().to(d, torch.bfloat16), b[perm].contiguous().to(d)
    for rep in range(2):
        out = ctx.op_linear_ln(xb, wp, bp, gamma.to(d), beta.to(d), act=_lib.ACT_GEGLU)
        assert out.shape == (M, 4 * C) and torch.isfinite(out).all()
        _close(out, ref, what=f"LayerNorm-folded geglu (run {rep})")


def test_linear_layernorm_folded_refuses_other_shapes(ctx):
    from rdm_amd._lib import RdmError
    d = ctx.device
    x, w = torch.zeros((100, 128), device=d, dtype=torch.bfloat16), torch.zeros((192, 128), device=d, dtype=torch.bfloat16)
    with pytest.raises(RdmError):
        ctx.op_linear_ln(x, w, None, torch.ones(128, device=d), torch.zeros(128, device=d))


@pytest.mark.parametrize("M,N,K,act", [(1, 768, 768, 0), (2, 2304, 768, 0), (33, 768, 3072, 0), (64, 16384, 768, 0), (128, 2304, 768, 0),
                                       (128, 768, 768, 3), (96, 512, 256, 2), (64, 6144, 768, 1), (128, 6144, 768, 1), (3, 1024, 256, 1)])
def test_skinny_linear(ctx, M, N, K, act):
    """Decode-sized batches (M <= 128, K % 256 == 0): the weight-streaming kernel of the RARM decode step (sgemm.hip): plain / SiLU /
    QuickGELU / GEGLU epilogues, bf16 and fp32 outputs, bf16 residual, every fragment count of M."""
    from rdm_amd import _lib
    from rdm_amd.packing import _geglu_perm
    d = ctx.device
    a, w, b = bf16_round(_rand((M, K), 21)), bf16_round(_rand((N, K), 22, K ** -0.5)), _rand((N,), 23, 0.1)
    y = a @ w.t() + b
    ab = a.to(d, torch.bfloat16)
    if act == 1:
        x, g = y.chunk(2, dim=-1)
        perm = _geglu_perm(N)
        out = ctx.op_linear(ab, w[perm].contiguous().to(d, torch.bfloat16), b[perm].contiguous().to(d), act=_lib.ACT_GEGLU)
        assert out.shape == (M, N // 2)
        _close(out, x * F.gelu(g), what="skinny geglu")
        return
    ref = {0: y, 2: y * torch.sigmoid(1.702 * y), 3: F.silu(y)}[act]
    wb = w.to(d, torch.bfloat16)
    _close(ctx.op_linear(ab, wb, b.to(d), act=act), ref, what="skinny linear")
    _close(ctx.op_linear(ab, wb, b.to(d), act=act, out_f32=True), ref, tol=2 ** -10, what="skinny linear f32 out")
    if act == 0:
        r = bf16_round(_rand((M, N), 24))
        _close(ctx.op_linear(ab, wb, None, residual=r.to(d, torch.bfloat16)), a @ w.t() + r, what="skinny residual")


def _conv_ref(x_nhwc, w, b, stride=1, ups=False):
    x = x_nhwc.permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x, w, b, stride=stride, padding=1)
    return y.permute(0, 2, 3, 1)


def _pack_conv(w):
    return w.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,W,C,N,stride,ups", [(2, 8, 8, 64, 192, 1, 0), (3, 16, 16, 128, 128, 1, 0), (2, 16, 16, 64, 64, 2, 0),
                                                  (2, 8, 8, 192, 192, 1, 1), (1, 32, 32, 64, 384, 1, 0), (5, 4, 4, 64, 64, 1, 0),
                                                  # fused nearest-2x upsample through the halo kernel (32x32 / 64x64 outputs) and the generic path (odd M)
                                                  (2, 16, 16, 128, 192, 1, 1), (1, 32, 32, 64, 192, 1, 1), (3, 4, 4, 64, 128, 1, 1),
                                                  # ... which now runs by output phase (four 2x2-tap convs on pre-summed weights): rectangular sources, tall / short tiles, N % 128 != 0
                                                  (4, 32, 16, 128, 384, 1, 1), (1, 2, 8, 64, 72, 1, 1), (9, 16, 16, 64, 576, 1, 1)])
def test_conv3x3(ctx, B, H, W, C, N, stride, ups):
    d = ctx.device
    x, w, b = bf16_round(_rand((B, H, W, C), 11)), bf16_round(_rand((N, C, 3, 3), 12, (9 * C) ** -0.5)), _rand((N,), 13, 0.1)
    ref = _conv_ref(x, w, b, stride, bool(ups))
    out = ctx.op_conv3x3(x.to(d, torch.bfloat16), _pack_conv(w).to(d, torch.bfloat16), b.to(d), stride=stride, ups=ups)
    assert tuple(out.shape) == tuple(ref.shape)
    _close(out, ref, what="conv3x3")


@pytest.mark.parametrize("B,H,W,C0,C1,N", [(4, 8, 8, 64, 0, 192), (8, 8, 8, 128, 64, 192), (1, 16, 16, 64, 64, 128), (2, 32, 32, 64, 0, 192),
                                            (1, 64, 64, 64, 0, 384), (1, 64, 64, 128, 0, 128), (3, 16, 16, 192, 0, 192),
                                            # few tiles -> K-split (2 or 3 fp32 partial planes + finisher): 8x8 level shapes, uneven slice split, dual source
                                            (8, 8, 8, 320, 0, 192), (16, 8, 8, 192, 128, 384), (4, 8, 8, 256, 0, 128), (2, 16, 16, 384, 0, 192)])
def test_conv3x3_halo_kernel(ctx, B, H, W, C0, C1, N):
    """Shapes the input-stationary halo kernel takes (B*H*W % 256 == 0, W <= 64): every tile geometry (4 samples per
    tile at 8x8, whole image at 16x16, 8 / 4 rows at 32 / 64 wide), dual source, time-embedding row and residual."""
    d = ctx.device
    C = C0 + C1
    x0 = bf16_round(_rand((B, H, W, C0), 40))
    x1 = bf16_round(_rand((B, H, W, C1), 41)) if C1 else None
    w, b = bf16_round(_rand((N, C, 3, 3), 42, (9 * C) ** -0.5)), _rand((N,), 43, 0.1)
    temb, res = _rand((B, N), 44), bf16_round(_rand((B, H, W, N), 45))
    xc = x0 if x1 is None else torch.cat([x0, x1], -1)
    ref = _conv_ref(xc, w, b) + temb[:, None, None, :] + res
    out = ctx.op_conv3x3(x0.to(d, torch.bfloat16), _pack_conv(w).to(d, torch.bfloat16), b.to(d),
                         x1=None if x1 is None else x1.to(d, torch.bfloat16), rowvec=temb.to(d), residual=res.to(d, torch.bfloat16))
    _close(out, ref, what="conv3x3 halo")
    out2 = ctx.op_conv3x3(x0.to(d, torch.bfloat16), _pack_conv(w).to(d, torch.bfloat16), b.to(d),
                          x1=None if x1 is None else x1.to(d, torch.bfloat16))
    _close(out2, _conv_ref(xc, w, b), what="conv3x3 halo plain")


# Images wider than 64 pixels: the one-wave-per-SIMD kernel on 64-column strips (conv_halo4.hip <.., STRIP>): the first-stage decoder's
# 128- and 256-pixel levels (plain, dual source, residual + per-sample row, fused nearest-2x upsample 64 -> 128 and 128 -> 256,
# non-square images, N = 128 / 256 / 512).  Random left / right neighbours make a strip that read zeros for its side halo fail.
@pytest.mark.parametrize("B,H,W,C0,C1,N,ups", [(2, 128, 128, 128, 0, 128, 0), (1, 128, 128, 256, 0, 256, 0), (1, 256, 256, 128, 0, 128, 0),
                                                (3, 64, 128, 64, 64, 256, 0), (1, 8, 192, 64, 0, 128, 0), (1, 4, 320, 64, 0, 128, 0),
                                                (2, 64, 64, 128, 0, 128, 1), (1, 128, 128, 64, 0, 256, 1), (1, 32, 96, 64, 0, 512, 1)])
def test_conv3x3_wide_images_as_strips(ctx, B, H, W, C0, C1, N, ups):
    d = ctx.device
    C = C0 + C1
    x0 = bf16_round(_rand((B, H, W, C0), 50))
    x1 = bf16_round(_rand((B, H, W, C1), 51)) if C1 else None
    w, b = bf16_round(_rand((N, C, 3, 3), 52, (9 * C) ** -0.5)), _rand((N,), 53, 0.1)
    xc = x0 if x1 is None else torch.cat([x0, x1], -1)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    ref = _conv_ref(xc, w, b, 1, bool(ups))
    out = ctx.op_conv3x3(x0.to(d, torch.bfloat16), _pack_conv(w).to(d, torch.bfloat16), b.to(d),
                         x1=None if x1 is None else x1.to(d, torch.bfloat16), ups=ups)
    assert tuple(out.shape) == (B, Ho, Wo, N)
    _close(out, ref, what="wide conv3x3")
    temb, res = _rand((B, N), 54), bf16_round(_rand((B, Ho, Wo, N), 55))
    out2 = ctx.op_conv3x3(x0.to(d, torch.bfloat16), _pack_conv(w).to(d, torch.bfloat16), b.to(d),
                          x1=None if x1 is None else x1.to(d, torch.bfloat16), rowvec=temb.to(d), residual=res.to(d, torch.bfloat16), ups=ups)
    _close(out2, ref + temb[:, None, None, :] + res, what="wide conv3x3 + row + residual")
    # bitwise the same as the generic implicit GEMM?  No (other summation order): but both must sit within the tolerance of the reference,
    # and the strip kernel must be the one that ran when it applies (env RDM_NO_HALO4_STRIP=1 switches it off for A/B)


def test_conv3x3_dual_source_rowvec_residual(ctx):
    d = ctx.device
    B, H, W, C0, C1, N = 2, 8, 8, 128, 64, 192
    x0, x1 = bf16_round(_rand((B, H, W, C0), 14)), bf16_round(_rand((B, H, W, C1), 15))
    w, b = bf16_round(_rand((N, C0 + C1, 3, 3), 16, (9 * (C0 + C1)) ** -0.5)), _rand((N,), 17, 0.1)
    temb, res = _rand((B, N + 7), 18), bf16_round(_rand((B, H, W, N), 19))
    # the time-embedding table is wider than N (row stride N + 7): the leading dimension is honoured, columns [0, N) are added
    out = ctx.op_conv3x3(x0.to(d, torch.bfloat16), _pack_conv(w).to(d, torch.bfloat16), b.to(d), x1=x1.to(d, torch.bfloat16),
                         rowvec=temb.to(d), residual=res.to(d, torch.bfloat16))
    ref = _conv_ref(torch.cat([x0, x1], -1), w, b) + temb[:, None, None, :N] + res
    _close(out, ref, what="conv dual+rowvec+res")


@pytest.mark.parametrize("B,HW,C0,C1,silu,eps", [(2, 64, 192, 0, 1, 1e-5), (3, 256, 128, 64, 1, 1e-5), (2, 16, 960, 0, 0, 1e-6),
                                                 (1, 1024, 576, 384, 1, 1e-5), (2, 4096, 64, 0, 1, 1e-6),
                                                 # the one-pass kernel's slices (round 5): every UNet shape of the 32 x 32 / 16 x 16 / 8 x 8 levels incl. the
                                                 # dual-source decoder inputs whose slices straddle the h | skip boundary, batches that do / do not take the
                                                 # XCD-grouped block order, 512- and 1024-thread blocks
                                                 (8, 1024, 384, 0, 1, 1e-5), (3, 1024, 384, 0, 0, 1e-6), (8, 1024, 576, 384, 1, 1e-5), (2, 1024, 384, 384, 1, 1e-5),
                                                 (8, 1024, 384, 192, 1, 1e-5), (16, 256, 576, 0, 1, 1e-5), (8, 256, 960, 576, 1, 1e-5), (5, 256, 576, 576, 1, 1e-5),
                                                 (8, 256, 576, 384, 1, 1e-5), (16, 64, 960, 0, 0, 1e-6), (8, 64, 960, 960, 1, 1e-5), (9, 64, 960, 576, 1, 1e-5),
                                                 (2, 4096, 192, 192, 1, 1e-5), (2, 4096, 384, 192, 1, 1e-5)])
def test_groupnorm(ctx, B, HW, C0, C1, silu, eps):
    d = ctx.device
    C = C0 + C1
    x0 = bf16_round(_rand((B, HW, C0), 20) * 2 + 0.5)
    x1 = bf16_round(_rand((B, HW, C1), 21)) if C1 else None
    g, b = 1 + 0.1 * _rand((C,), 22), 0.1 * _rand((C,), 23)
    xc = x0 if x1 is None else torch.cat([x0, x1], -1)
    ref = F.group_norm(xc.permute(0, 2, 1), 32, g, b, eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    out = ctx.op_groupnorm(x0.to(d, torch.bfloat16), g.to(d), b.to(d), eps, silu, None if x1 is None else x1.to(d, torch.bfloat16))
    _close(out, ref, what="groupnorm")


@pytest.mark.parametrize("M,C,f32", [(100, 384, False), (77, 512, True), (513, 960, False), (9, 768, True), (4, 128, False),
                                     (16384 + 13, 384, False), (16384 + 2, 960, False)])        # large ragged M
def test_layernorm(ctx, M, C, f32):
    d = ctx.device
    x = _rand((M, C), 24) * 1.5 + 0.3
    if not f32:
        x = bf16_round(x)
    g, b = 1 + 0.1 * _rand((C,), 25), 0.1 * _rand((C,), 26)
    ref = F.layer_norm(x, (C,), g, b, 1e-5)
    out = ctx.op_layernorm(x.to(d) if f32 else x.to(d, torch.bfloat16), g.to(d), b.to(d))
    _close(out, ref, what="layernorm")


@pytest.mark.parametrize("B,n,heads", [(2, 64, 2), (1, 256, 6), (2, 1024, 3), (3, 32, 4)])
def test_flash_self_attention(ctx, B, n, heads):
    d = ctx.device
    C = heads * 32
    q, k, v = (bf16_round(_rand((B, n, C), s)) for s in (27, 28, 29))
    k = k * 2.0   # wider logits
    sp = lambda t: t.reshape(B, n, heads, 32).permute(0, 2, 1, 3)
    att = (sp(q) @ sp(k).transpose(-1, -2) * 32 ** -0.5).softmax(-1)
    ref = (att @ sp(v)).permute(0, 2, 1, 3).reshape(B, n, C)
    qk = torch.cat([q, k], -1).contiguous()
    vt = v.permute(0, 2, 1).contiguous()                     # [B, C, n]
    out = ctx.op_self_attention(qk.to(d, torch.bfloat16), vt.to(d, torch.bfloat16), heads)
    _close(out, ref, tol=2 ** -6, what="flash attention")    # P is rounded to bf16 before the PV MFMA
    if n % 64 == 0:      # the fused-projection form: token-major V through the kernel's transpose reads; same arithmetic, same bits
        qkv = torch.cat([q, k, v], -1).contiguous().to(d, torch.bfloat16)
        out2 = ctx.op_self_attention_qkv(qkv, heads)
        assert torch.equal(out2, out), "token-major-V flash kernel differs from the V^T one"


@pytest.mark.parametrize("B,H,W,C,Cout,norm", [(2, 64, 64, 192, 3, True), (1, 32, 96, 128, 3, True), (3, 16, 32, 64, 4, False), (2, 34, 64, 224, 3, True),
                                               (70, 8, 32, 32, 1, True)])
def test_head_conv(ctx, B, H, W, C, Cout, norm):
    """GroupNorm + SiLU + 3x3 head conv in one kernel against fp32 torch (activations rounded to bf16 after the norm, as the two-kernel
    form rounds them; weights fp32: the kernel carries them as bf16 high + low parts)."""
    d = ctx.device
    x = bf16_round(_rand((B, H, W, C), 50) * 1.3 + 0.2)
    w = _rand((Cout, C, 3, 3), 51) / (3 * C ** 0.5)
    bias = _rand((Cout,), 52)
    g, be = 1 + 0.1 * _rand((C,), 53), 0.1 * _rand((C,), 54)
    xc = x.permute(0, 3, 1, 2)
    act = bf16_round(F.silu(F.group_norm(xc, 32, g, be, 1e-5))) if norm else xc
    ref = F.conv2d(act, w, bias, padding=1)
    out = ctx.op_head_conv(x.to(d, torch.bfloat16), w.to(d), bias.to(d), gn=(g.to(d), be.to(d), 1e-5) if norm else None)
    _close(out, ref, tol=2e-3, what="head conv")


@pytest.mark.parametrize("B,n,heads,k", [(3, 1024, 12, 4), (2, 256, 18, 4), (4, 64, 30, 4), (2, 64, 4, 2), (2, 32, 2, 1), (2, 96, 6, 4),
                                        (8, 64, 4, 4), (16, 256, 2, 2)])       # the last two: block counts that take the XCD-aware block order
def test_xattn_fused(ctx, B, n, heads, k):
    """Fused skinny cross-attention (scores GEMM + group softmax + output GEMM + bias + residual) against fp32 torch on the same bf16 operands."""
    d = ctx.device
    C, NP, ncols = heads * 32, 128, heads * k
    x, res = bf16_round(_rand((B, n, C), 40)), bf16_round(_rand((B, n, C), 41))
    G = torch.zeros(B, NP, C); U = torch.zeros(B, C, NP)
    G[:, :ncols] = _rand((B, ncols, C), 42) * (4.0 / C ** 0.5)       # logits of a few units
    U[:, :, :ncols] = _rand((B, C, ncols), 43)
    G, U = bf16_round(G), bf16_round(U)
    bias = _rand((C,), 44)
    sc = torch.einsum("bnc,bjc->bnj", x, G[:, :ncols])
    pr = bf16_round(sc.reshape(B, n, heads, k).softmax(-1).reshape(B, n, ncols))     # the kernel rounds the probabilities to bf16 for the second MFMA
    ref = torch.einsum("bnj,bcj->bnc", pr, U[:, :, :ncols]) + bias + res
    dev = lambda t: t.to(d, torch.bfloat16).contiguous()
    out = ctx.op_xattn_fused(dev(x), dev(G), dev(U), bias.to(d), dev(res), ncols, k)
    _close(out, ref, tol=2 ** -6, what="fused cross attention")
    out0 = ctx.op_xattn_fused(dev(x), dev(G), dev(U), None, None, ncols, k)
    _close(out0, ref - bias - res, tol=2 ** -6, what="fused cross attention (no bias / residual)")
    # LayerNorm in front and the residual folded in: raw rows with a mean and a scale, gamma / beta non-trivial
    raw = bf16_round(_rand((B, n, C), 45) * 1.7 + 0.4)
    g, be = 1 + 0.1 * _rand((C,), 46), 0.1 * _rand((C,), 47)
    xn = bf16_round(F.layer_norm(raw, (C,), g, be, 1e-5))
    sc = torch.einsum("bnc,bjc->bnj", xn, G[:, :ncols])
    pr = bf16_round(sc.reshape(B, n, heads, k).softmax(-1).reshape(B, n, ncols))
    ref_ln = torch.einsum("bnj,bcj->bnc", pr, U[:, :, :ncols]) + bias + raw
    out_ln = ctx.op_xattn_fused(dev(raw), dev(G), dev(U), bias.to(d), None, ncols, k, ln=(g.to(d), be.to(d), 1e-5))
    _close(out_ln, ref_ln, tol=2 ** -6, what="fused LayerNorm + cross attention + residual")


@pytest.mark.parametrize("B,n,heads,k", [(3, 1024, 12, 4), (2, 256, 18, 4), (4, 64, 30, 4), (2, 32, 2, 1), (16, 256, 2, 2)])
def test_xattn_fused_in_place_with_norm3(ctx, B, n, heads, k):
    """norm2 + attn2 + residual IN PLACE and norm3 of the finished rows in the same launch (what the executor runs on the conditional rows
    of a guided batch): x against fp32 torch on the bf16 operands, norm3 against F.layer_norm of the kernel's OWN bf16 rows (the
    statistics are taken on the rounded values, like the separate LayerNorm pass that reads them back)."""
    d = ctx.device
    C, NP, ncols = heads * 32, 128, heads * k
    G = torch.zeros(B, NP, C); U = torch.zeros(B, C, NP)
    G[:, :ncols] = _rand((B, ncols, C), 52) * (4.0 / C ** 0.5)
    U[:, :, :ncols] = _rand((B, C, ncols), 53)
    G, U = bf16_round(G), bf16_round(U)
    bias = _rand((C,), 54)
    raw = bf16_round(_rand((B, n, C), 55) * 1.7 + 0.4)
    g, be = 1 + 0.1 * _rand((C,), 56), 0.1 * _rand((C,), 57)
    g3, be3 = 1 + 0.2 * _rand((C,), 58), 0.2 * _rand((C,), 59)
    xn = bf16_round(F.layer_norm(raw, (C,), g, be, 1e-5))
    sc = torch.einsum("bnc,bjc->bnj", xn, G[:, :ncols])
    pr = bf16_round(sc.reshape(B, n, heads, k).softmax(-1).reshape(B, n, ncols))
    ref = torch.einsum("bnj,bcj->bnc", pr, U[:, :, :ncols]) + bias + raw
    dev = lambda t: t.to(d, torch.bfloat16).contiguous()
    x = dev(raw)
    l3 = ctx.op_xattn_fused_ln3(x, dev(G), dev(U), bias.to(d), ncols, k, ln=(g.to(d), be.to(d), 1e-5), ln3=(g3.to(d), be3.to(d)))
    _close(x, ref, tol=2 ** -6, what="in-place LayerNorm + cross attention + residual")
    # the separate-output form gives the same bits
    sep = ctx.op_xattn_fused(dev(raw), dev(G), dev(U), bias.to(d), None, ncols, k, ln=(g.to(d), be.to(d), 1e-5))
    assert torch.equal(sep, x)
    ref3 = F.layer_norm(x.float().cpu(), (C,), g3, be3, 1e-5)
    _close(l3, ref3, tol=2 ** -7, what="norm3 emitted by the cross-attention kernel")
    # ... and agrees with the library's own LayerNorm pass over the same rows to the last rounding
    own = ctx.op_layernorm(x.reshape(B * n, C), g3.to(d), be3.to(d), 1e-5).reshape(B, n, C)
    assert (own.float() - l3.float()).abs().max().item() <= 2 ** -7 * ref3.abs().max().item()


@pytest.mark.parametrize("B,nq,nkv,heads,D,causal", [(2, 64, 4, 4, 32, 0), (2, 77, 77, 2, 64, 1), (1, 50, 50, 3, 64, 0),
                                                     (2, 16, 16, 2, 32, 0), (1, 1024, 16, 12, 32, 0)])
def test_small_attention(ctx, B, nq, nkv, heads, D, causal):
    d = ctx.device
    C = heads * D
    q, k, v = bf16_round(_rand((B, nq, C), 30)), bf16_round(_rand((B, nkv, C), 31)), bf16_round(_rand((B, nkv, C), 32))
    sp = lambda t: t.reshape(B, t.shape[1], heads, D).permute(0, 2, 1, 3)
    s = sp(q) @ sp(k).transpose(-1, -2) * D ** -0.5
    if causal:
        s = s + torch.full((nq, nkv), float("-inf")).triu_(1)
    ref = (s.softmax(-1) @ sp(v)).permute(0, 2, 1, 3).reshape(B, nq, C)
    out = ctx.op_small_attention(q.to(d, torch.bfloat16), k.to(d, torch.bfloat16), v.to(d, torch.bfloat16), heads, D, causal, D ** -0.5)
    _close(out, ref, what="small attention")


@pytest.mark.parametrize("B,H,W", [(2, 96, 80), (1, 256, 256), (3, 224, 224), (1, 37, 501), (2, 1200, 900)])
def test_clip_preprocess_bicubic(ctx, B, H, W):
    """ClipImageRetriever.preprocess (rdm/modules/retrievers.py:83-91): bicubic resize with align_corners=True (kornia 0.6.2 ->
    F.interpolate; un-vendored, so the fp32 torch CPU op is the reference), (x+1)/2, CLIP mean/std.  Non-identity sizes, up- and
    down-scaling, and the identity size."""
    from oracle import clip as oclip, unet as ounet
    from rdm_amd import packing
    from _util import spec_to_clip_cfg
    spec = oclip.ClipSpec(embed_dim=64, image_resolution=224, vision_layers=1, vision_width=128, vision_patch_size=32,
                          context_length=77, vocab_size=1000, transformer_width=128, transformer_heads=2, transformer_layers=1)
    cfg = spec_to_clip_cfg(spec)
    sd = ounet.synth_state_dict(oclip.clip_param_shapes(spec), seed=3)
    ctx.load_clip(cfg, packing.pack("clip", cfg, sd))
    x = torch.from_numpy(np.random.default_rng(H * 7 + W).uniform(-1, 1, (B, 3, H, W)).astype(np.float32))
    got = ctx.clip_preprocess(x)
    torch.cuda.synchronize()
    # reference in fp64 (exact tap positions).  The kernel, like the reference's CUDA op, evaluates the taps at the fp32-rounded
    # source position oy*(H-1)/(R-1): a position error <= ulp(H)/2 times the local slope (<= 2 per pixel for this +-1 noise
    # image), times 1/(2 std) = 1.9 from the normalisation -- hence a bound that grows with the source size.
    ref = torch.nn.functional.interpolate(x.double(), size=(224, 224), mode="bicubic", align_corners=True)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], dtype=torch.float64)[None, :, None, None]
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], dtype=torch.float64)[None, :, None, None]
    ref = ((ref + 1.) / 2. - mean) / std
    err = float((got.cpu().double() - ref).abs().max())
    bound = 2e-5 + 1.9 * 2.0 * max(H, W) * 2.0 ** -23
    print(f"bicubic {H}x{W} -> 224x224: max |err| {err:.2e} (bound {bound:.2e})")
    assert got.shape == (B, 3, 224, 224) and err <= bound
    # a smooth image (slope ~1e-2 per pixel): the position rounding no longer matters, plain fp32 accuracy remains
    yy, xx = torch.meshgrid(torch.linspace(0, 3, H), torch.linspace(0, 2, W), indexing="ij")
    xs = (torch.sin(yy)[None, None] * torch.cos(xx)[None, None]).expand(1, 3, H, W).contiguous().float()
    gs = ctx.clip_preprocess(xs).cpu().double()
    rs = ((torch.nn.functional.interpolate(xs.double(), size=(224, 224), mode="bicubic", align_corners=True) + 1.) / 2. - mean) / std
    assert float((gs - rs).abs().max()) <= 2e-5
    # fused variant (resize feeds the patch-embedding GEMM directly) == tower on the materialised preprocess output
    a = ctx.clip_encode_image_raw(x)
    b = ctx.clip_encode_image(got)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_cross_attention_reference_golden(ctx):
    """rdm.modules.attention.CrossAttention (attention.py:20-74) on the golden generated from the reference class
    (tests/golden/attention.npz, tools/gen_golden.py): q/k/v projections, 4 heads of 32 over k = 4 neighbours, output projection —
    composed from the library's operator entry points."""
    from oracle import unet as ounet
    from _util import golden, rel_l2
    g = golden("attention.npz")
    C, heads = 128, 4
    shapes = {"to_q.weight": (C, C), "to_k.weight": (C, 512), "to_v.weight": (C, 512), "to_out.0.weight": (C, C), "to_out.0.bias": (C,)}
    sd = ounet.synth_state_dict(shapes, seed=int(g["seed"]) + 1)
    d = ctx.device
    bf = lambda t: t.to(d, torch.bfloat16).contiguous()
    x, cx = torch.from_numpy(g["ca_x"]), torch.from_numpy(g["ca_ctx"])
    q = ctx.op_linear(bf(x.reshape(-1, C)), bf(sd["to_q.weight"])).reshape(2, 64, C)
    k = ctx.op_linear(bf(cx.reshape(-1, 512)), bf(sd["to_k.weight"])).reshape(2, 4, C)
    v = ctx.op_linear(bf(cx.reshape(-1, 512)), bf(sd["to_v.weight"])).reshape(2, 4, C)
    a = ctx.op_small_attention(q, k, v, heads, 32, False, 32 ** -0.5)
    y = ctx.op_linear(a.reshape(-1, C), bf(sd["to_out.0.weight"]), sd["to_out.0.bias"].to(d), out_f32=True).reshape(2, 64, C)
    torch.cuda.synchronize()
    e = rel_l2(y, torch.from_numpy(g["ca_y"]))
    print("CrossAttention vs reference golden rel L2:", e)
    assert e <= 2e-2


def test_ffn_fused_matches_the_two_kernel_feed_forward(ctx):
    """Round 6 (verdict item 2): csrc/ffn.hip's fused GEGLU -> ff.net.2 x proj_out kernel (the hidden tensor never in HBM; a measurement vehicle, not used
    by the executors) against an fp32 torch reference on bf16-rounded operands -- [x | g] = l3 W1^T + b1, ff = bf16(x gelu(g)), out = bf16([ff | t2] Wf^T +
    bf + x_in) (rdm/modules/attention.py:77-96; ldm GEGLU) -- and against the two-kernel path of the executors (same roundings: flips only)."""
    from rdm_amd import _lib
    from rdm_amd.packing import _geglu_perm
    from _util import rel_l2
    d = ctx.device
    g = torch.Generator().manual_seed(9)
    M, C = 512, 384
    bf = lambda t: t.to(torch.bfloat16)
    R = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    l3, t2, xin = bf(R(M, C)), bf(R(M, C)), bf(R(M, C))
    w1, b1 = bf(R(8 * C, C, sc=C ** -0.5)), R(8 * C, sc=0.3)
    wf, bfb = bf(R(C, 5 * C, sc=(5 * C) ** -0.5)), R(C, sc=0.3)
    pp = l3.float() @ w1.float().t() + b1
    x, gate = pp.chunk(2, dim=-1)
    ff = bf(x * F.gelu(gate)).float()
    ref = bf(torch.cat([ff, t2.float()], dim=1) @ wf.float().t() + bfb + xin.float()).float()
    perm = torch.as_tensor(_geglu_perm(8 * C))
    dev = lambda t: t.to(d).contiguous()
    out = ctx.op_ffn_fused(dev(l3), dev(t2), dev(xin), dev(w1[perm]), dev(b1[perm]), dev(wf), dev(bfb))
    hid = ctx.op_linear(dev(l3), dev(w1[perm]), dev(b1[perm]), act=_lib.ACT_GEGLU)
    pair = ctx.op_linear(torch.cat([hid, dev(t2)], dim=1).contiguous(), dev(wf), dev(bfb), residual=dev(xin))
    torch.cuda.synchronize()
    e, e2 = rel_l2(out, ref), rel_l2(out, pair.float())
    print(f"fused feed-forward vs fp32 reference {e:.3e}, vs the two-kernel path {e2:.3e}")
    assert e <= 3e-3 and e2 <= 3e-3
    with pytest.raises(_lib.RdmError):
        ctx.op_ffn_fused(dev(l3[:100]), dev(t2[:100]), dev(xin[:100]), dev(w1[perm]), dev(b1[perm]), dev(wf), dev(bfb))      # M % 128
