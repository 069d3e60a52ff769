#!/usr/bin/env python3
"""Op-by-op distance between the library and oracle/unet_emul.py's formulas (GPU box): locates a storage point the restatement misses.
Every line should read at the flip level (<~ 5e-4); a line at ~2-4e-3 is one bf16 rounding that one side has and the other has not."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import rdm_amd
from rdm_amd import _lib, packing
from oracle import unet as ounet
from oracle.unet_emul import unet_forward_emulated
from _util import rel_l2, spec_to_unet_cfg
torch.set_grad_enabled(False)
ctx = _lib.Context(0); d = ctx.device
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
g = torch.Generator().manual_seed(1)
R = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
dev = lambda t: t.to(d, torch.bfloat16).contiguous()

# GroupNorm + SiLU
x = bf(R(2, 1024, 384) * 1.3 + 0.2); ga, be = 1 + 0.1 * R(384), 0.1 * R(384)
out = ctx.op_groupnorm(dev(x), ga.to(d), be.to(d), 1e-5, 1)
ref = bf(F.silu(F.group_norm(x.permute(0, 2, 1).reshape(2, 384, 32, 32), 32, ga, be, 1e-5))).reshape(2, 384, 1024).permute(0, 2, 1)
print("groupnorm+silu", rel_l2(out, ref))
# LayerNorm
out = ctx.op_layernorm(dev(x).reshape(2048, 384), ga.to(d), be.to(d)); ref = bf(F.layer_norm(x, (384,), ga, be, 1e-5)).reshape(2048, 384)
print("layernorm", rel_l2(out, ref))
# linear + bias + residual (lin4-sized and small)
for M, N, K in ((49152, 384, 384), (2048, 384, 384), (8192, 960, 960)):
    a, w, b, r = bf(R(M, K)), bf(R(N, K, sc=K ** -0.5)), R(N, sc=0.3), bf(R(M, N))
    out = ctx.op_linear(dev(a), dev(w), b.to(d), residual=dev(r)); ref = bf(a @ w.t() + b + r)
    print(f"linear {M}x{N}x{K}", rel_l2(out, ref))
# GEGLU
from rdm_amd.packing import _geglu_perm
M, C = 32768, 384
a, w, b = bf(R(M, C)), bf(R(8 * C, C, sc=C ** -0.5)), R(8 * C, sc=0.3)
xg, gg = (a @ w.t() + b).chunk(2, dim=-1); ref = bf(xg * F.gelu(gg))
perm = _geglu_perm(8 * C)
out = ctx.op_linear(dev(a), dev(w[perm]), b[perm].contiguous().to(d), act=_lib.ACT_GEGLU)
print("geglu", rel_l2(out, ref))
# conv3x3 + bias (+ rowvec) (+ residual), small and halo-kernel-sized batches
for B in (2, 16):
    H, C, N = 32, 384, 384
    xx, w, b, rv, r = bf(R(B, H, H, C)), bf(R(N, C, 3, 3, sc=(9 * C) ** -0.5)), R(N, sc=0.2), R(B, N, sc=0.3), bf(R(B, H, H, N))
    base = F.conv2d(xx.permute(0, 3, 1, 2), w, b, padding=1)
    for name, kw, extra in (("plain", {}, 0), ("+rowvec", dict(rowvec=rv.to(d).contiguous()), rv[:, :, None, None]),
                            ("+res", dict(residual=dev(r)), r.permute(0, 3, 1, 2)),
                            ("+rowvec+res", dict(rowvec=rv.to(d).contiguous(), residual=dev(r)), rv[:, :, None, None] + r.permute(0, 3, 1, 2))):
        out = ctx.op_conv3x3(dev(xx), dev(w.permute(0, 2, 3, 1)), b.to(d), **kw)
        print(f"conv3x3 B={B} {name}", rel_l2(out, bf(base + extra).permute(0, 2, 3, 1)))
# self attention from qkv
B, n, heads = 2, 1024, 12; C = heads * 32
qkv = bf(R(B, n, 3 * C))
out = ctx.op_self_attention_qkv(dev(qkv), heads)
q, k, v = qkv.split(C, dim=-1); sp = lambda t: t.reshape(B, n, heads, 32).permute(0, 2, 1, 3)
from oracle.unet_emul import flash_attention, _R
o = flash_attention(sp(q), sp(k), sp(v), 32 ** -0.5, _R(True))
print("self-attention", rel_l2(out, bf(o.permute(0, 2, 1, 3).reshape(B, n, C))))
# block by block through the debug tap, TEACHER-FORCED (the restatement's block i + 1 starts from the library's block i)
from oracle import diffusion as odiff
def lib_taps(run, shapes):
    out = []
    for bi, shp in enumerate(shapes):
        buf = torch.empty(shp, device=d, dtype=torch.bfloat16)
        ctx.debug_tap(buf, bi); run(); torch.cuda.synchronize()
        out.append(buf.float().cpu().permute(0, 3, 1, 2).contiguous())
    ctx.debug_tap(None, -1)
    return out
for name, spec in (("tiny", ounet.tiny_spec()), ("shipped", ounet.shipped_spec())):
    sd = ounet.synth_state_dict(ounet.param_shapes(spec), seed=1234)
    cfg = spec_to_unet_cfg(spec); ctx.load_unet(cfg, packing.pack("unet", cfg, sd))
    hw = 16 if name == "tiny" else 64
    x = R(2, 3, hw, hw); t = torch.tensor([981, 501]); c = R(2, 4, 512, sc=0.45)
    free = []
    unet_forward_emulated(sd, spec, x, t, c, taps=free)
    lt = lib_taps(lambda: ctx.unet_forward(x, t, c), [(2, tp.shape[2], tp.shape[3], tp.shape[1]) for tp in free])
    forced = []
    emu = unet_forward_emulated(sd, spec, x, t, c, taps=forced, forced=dict(enumerate(lt)))
    for bi in range(len(lt)):
        print(f"unet {name} block {bi:2d} {spec.blocks[bi][0]:18s} {[l[0] for l in spec.blocks[bi][1]]}: free {rel_l2(lt[bi], free[bi]):.3e} teacher-forced {rel_l2(lt[bi], forced[bi]):.3e}")
    eps = ctx.unet_forward(x, t, c)
    print(f"unet {name}: eps teacher-forced {rel_l2(eps, emu):.3e}")
# inside the sampler: guided batch [x | x], contexts [cond | 0]: shared prefix + zero-neighbour rows
sched = odiff.Schedule()
B = 2
x_T = R(B, 3, 64, 64); cond = R(B, 4, 512, sc=0.45); uncond = torch.zeros_like(cond)
first_st = next(i for i, (_, ls) in enumerate(spec.blocks) if any(l[0] == "st" for l in ls))
shapes = [((B if bi < first_st else 2 * B), tp.shape[2], tp.shape[3], tp.shape[1]) for bi, tp in enumerate(free)]
lt = lib_taps(lambda: ctx.ddim_sample(1, x_T, cond, uncond, sched.alphas_cumprod, eta=0.0, scale=2.0), shapes)
lt = [torch.cat([v, v]) if bi < first_st else v for bi, v in enumerate(lt)]
forced = []
tt = torch.full((2 * B,), 1, dtype=torch.long)
unet_forward_emulated(sd, spec, torch.cat([x_T, x_T]), tt, torch.cat([cond, uncond]), ctx_rows=B, taps=forced, forced=dict(enumerate(lt)))
for bi in range(len(lt)):
    print(f"sampler block {bi:2d}: teacher-forced {rel_l2(lt[bi], forced[bi]):.3e}   cond rows {rel_l2(lt[bi][:B], forced[bi][:B]):.3e} uncond rows {rel_l2(lt[bi][B:], forced[bi][B:]):.3e}")
