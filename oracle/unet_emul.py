"""The LIBRARY's arithmetic for one UNet forward, restated on the CPU (oracle — test infrastructure only).

Purpose (verdict round 4, item 4 "close the parity window once"): the fp32 oracle (oracle/unet.py, pinned bit-for-bit against the reference's
own classes) and the library differ by bf16 STORAGE rounding, ~1.2e-2 relative L2 per forward at the shipped size -- a window wide enough
to hide a mis-wired residual or a tanh-for-erf GELU.  This module walks the SAME graph with the same state dict in fp32 arithmetic and
rounds to bf16 at exactly the points where the library stores a bf16 tensor (csrc/model.hip: unet_body), with the library's algebraic
re-associations done the library's way (each cites its site):

  * weights of every GEMM / 3x3 conv in bf16; biases, norm affine, conv_in and the `out` conv in fp32;
  * time embedding: bf16 sinusoid, SiLU folded into the two MLP outputs (only SiLU(emb) is ever consumed), emb rows in fp32;
  * GroupNorm / LayerNorm statistics in fp32 on the bf16 values, outputs rounded once;
  * conv: fp32 accumulate + bias + emb row (+ residual), one rounding;
  * self-attention: fp32 scores and row sums, probabilities rounded to bf16 for the P.V product (flash_d32_lds_kernel);
  * cross-attention over the k neighbours re-associated per sample (unet_compute_xattn): G = bf16(bf16(K / sqrt d) W_q), U = bf16(W_o V^T),
    scores on bf16(LayerNorm-2), group softmax in fp32, bf16 probabilities, + bias + residual, one rounding;
    rows >= ctx_rows (all-zero neighbours: the unconditional half of a guided batch) get t2 = to_out1(...) + b_o1 + b_o2 + t0 in ONE rounding;
  * GEGLU: x * gelu_erf(gate) in fp32, one rounding;
  * ff.net.2 and proj_out as ONE linear map of [ff | t2] with the product weights bf16([W_out W_2 | W_out]) (packing.py `fuse_w`, fp64 product);
  * Upsample: nearest-2x + conv3x3 by output phase on pre-summed weights (fp32 sum of the bf16 taps, one rounding: conv_phase_weights_kernel);
  * `out`: bf16(SiLU(GroupNorm)) against fp32 weights, fp32 result.

What remains between this and the library is fp32 summation order (MFMA vs the CPU's) and the bf16 roundings those differences flip:
measured ~1e-3 relative L2 at the shipped size, so tests/test_gpu_emul.py holds the UNet forward, a guided DDIM step and the VQ decoder at
2e-3 -- six times tighter than the fp32 bound.  With `rounding=False` the same code path is exact algebra: tests/test_oracle_cpu.py pins it
to the fp32 oracle (<= 2e-5), so the re-associations above are verified independently of any GPU.

Follows (through oracle/unet.py): rdm/modules/diffusionmodules/openaimodel.py:335-371, rdm/modules/attention.py:16-196, ldm ResBlock /
GroupNorm32 / Upsample / Downsample / GEGLU."""
import torch
import torch.nn.functional as F

from .unet import UNetSpec, group_norm, timestep_embedding


class _R:
    """Rounding policy: bf() = one bf16 storage rounding (identity with rounding=False)."""

    def __init__(self, rounding=True):
        self.on = rounding

    def bf(self, t):
        return t.to(torch.bfloat16).to(torch.float32) if self.on else t


def flash_attention(q, k, v, scale, R, tile=32):
    """softmax(q k^T scale) v as flash_d32_lds_kernel forms it (csrc/attention.hip).  q, k, v [B, heads, n, d] (bf16 values).
    The probabilities the P.V product sees are rounded to bf16 at the scale of the RUNNING row maximum -- the maximum over the keys of
    all 32-key tiles up to and including the key's own -- not of the final one: p_j = bf16(2^(c s_j - m_t(j))), later rescaled in fp32 by
    2^(m_t - m_final).  (Rounding exp(s - m_final) instead is a different, equally valid set of 2^-9 errors: on zero-mean values the
    two outputs differ by ~2e-3, as much as the whole budget of tests/test_gpu_emul.py.)  Row sums use the unrounded p."""
    n = k.shape[2]
    c = torch.tensor(scale * 1.4426950408889634, dtype=torch.float32)
    s = torch.einsum("bhid,bhjd->bhij", q, k) * c                                  # log2-domain scores
    if not R.on:
        p = torch.exp2(s - s.amax(dim=-1, keepdim=True))
        return torch.einsum("bhij,bhjd->bhid", p, v) / p.sum(dim=-1, keepdim=True)
    nt = (n + tile - 1) // tile
    pad = nt * tile - n
    sp_ = F.pad(s, (0, pad), value=float("-inf")) if pad else s
    m_run = torch.cummax(sp_.reshape(*s.shape[:-1], nt, tile).amax(dim=-1), dim=-1).values          # [B, h, n_q, tiles]
    m_key = m_run.repeat_interleave(tile, dim=-1)[..., :n]
    p = torch.exp2(s - m_key)
    wgt = torch.exp2(m_key - m_run[..., -1:])
    o = torch.einsum("bhij,bhjd->bhid", R.bf(p) * wgt, v)
    return o / (p * wgt).sum(dim=-1, keepdim=True)


def _phase_weights(w, R):
    """[N, C, 3, 3] (bf16 values) -> four [N, C, 3, 3] kernels at SOURCE resolution, one per output phase (a, b): output pixel (2y + a, 2x + b)
    of conv3x3(nearest2x(src)) reads source rows {y - 1 + a, y + a} x columns {x - 1 + b, x + b}; the taps landing on one source pixel
    are summed in fp32 and rounded once (csrc/igemm.hip conv_phase_weights_kernel)."""
    def fold(t, a, dim):        # 3 taps along `dim` -> source offsets (-1, 0, +1)
        t0, t1, t2 = t.unbind(dim)
        z = torch.zeros_like(t0)
        parts = (t0, t1 + t2, z) if a == 0 else (z, t0 + t1, t2)
        return torch.stack(parts, dim)
    return {(a, b): R.bf(fold(fold(w, a, 2), b, 3)) for a in (0, 1) for b in (0, 1)}


def _upsample_conv(x, w, bias, R):
    B, C, H, W_ = x.shape
    out = torch.empty((B, w.shape[0], 2 * H, 2 * W_), dtype=torch.float32)
    for (a, b), wp in _phase_weights(w, R).items():
        out[:, :, a::2, b::2] = F.conv2d(x, wp, bias, padding=1)
    return out


# ---- Winograd F(2x2, 3x3) with the roundings a bf16-MFMA kernel would have (round 6: the accuracy half of verdict item 1; zero GPU)
_WG = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
_WBT = torch.tensor([[1.0, 0.0, -1.0, 0.0], [0.0, 1.0, 1.0, 0.0], [0.0, -1.0, 1.0, 0.0], [0.0, 1.0, 0.0, -1.0]])
_WAT = torch.tensor([[1.0, 1.0, 1.0, 0.0], [0.0, 1.0, -1.0, -1.0]])


def wino_conv3x3(x, w, R, two_stage=False):
    """conv2d(x, w, padding=1) for even H, W as Winograd F(2x2, 3x3) (Lavin & Gray) in the arithmetic a fused bf16-MFMA kernel would use:
    U = bf16(G g G^T) from the bf16 weights, V = bf16(B^T d B) formed in fp32 from the bf16 activations (two_stage: the row transform is
    rounded to bf16 too, as a packed-bf16 transform would), the 16 element-wise products summed over the input channels in fp32 (bf16 x
    bf16 products are exact in fp32: the MFMA's accumulation), Y = A^T M A in fp32.  No bias / rounding of Y: the caller adds and rounds
    once, like the direct conv's epilogue.  x [B, C, H, W] bf16 values, w [N, C, 3, 3] bf16 values -> [B, N, H, W] fp32."""
    B, C, H, W_ = x.shape
    U = R.bf(torch.einsum("ai,ncij,bj->ncab", _WG, w, _WG))                                 # [N, C, 4, 4]
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)                              # [B, C, H/2, W/2, 4, 4]
    if two_stage:
        V = R.bf(torch.einsum("ai,bcyxij->bcyxaj", _WBT, d))
        V = R.bf(torch.einsum("bcyxaj,ej->bcyxae", V, _WBT))
    else:
        V = R.bf(torch.einsum("ai,bcyxij,ej->bcyxae", _WBT, d, _WBT))
    M = torch.einsum("ncae,bcyxae->bnyxae", U, V)
    Y = torch.einsum("pa,bnyxae,qe->bnyxpq", _WAT, M, _WAT)                                 # [B, N, H/2, W/2, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, W_)


def unet_forward_emulated(sd, spec: UNetSpec, x, timesteps, context, ctx_rows=None, rounding=True, taps=None, forced=None, wino=None, wino_two_stage=False):
    """x [B,Cin,H,W] f32, timesteps [B] int64, context [B,k,context_dim] f32 -> eps [B,Cout,H,W] f32, in the library's arithmetic.
    ctx_rows: samples [ctx_rows, B) have all-zero neighbours and take the library's shortcut (default: none do).
    taps: a dict that receives every intermediate the library can show through rdm_debug_tap, keyed (block, sub) like the C ABI
    (sub 0 = the block's output, else 16 * layer + stage), each as [B, H*W, width] (the executor's NHWC rows).
    forced: {(block, sub): tensor}: TEACHER FORCING -- after a stage's own value went into `taps`, the given tensor (the library's value of
    that stage) replaces it for everything downstream, so that the NEXT stage's tap measures that stage alone.  (Free-running, the two
    sides decorrelate within a few layers: a discrepancy d ahead of a bf16 rounding comes out as ~sqrt(d * ulp) behind it -- 1e-6 -> 6e-5 ->
    5e-4 -> 1.4e-3 -> ... -> the rounding noise floor itself; tests/test_gpu_emul.py.)"""
    R = _R(rounding)
    bf = R.bf
    Wb = lambda k: bf(sd[k].float())                      # a bf16-stored weight
    # wino: set of spatial heights at which the ResBlocks' stride-1 3x3 convs run as Winograd F(2x2, 3x3) (tools/wino_accuracy.py)
    conv3 = lambda h, wk: wino_conv3x3(h, Wb(wk), R, wino_two_stage) if (wino and h.shape[2] in wino) else F.conv2d(h, Wb(wk), None, padding=1)
    Bn = x.shape[0]
    ctx_rows = Bn if ctx_rows is None else int(ctx_rows)
    mc = spec.model_channels
    to_tok = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], t.shape[2] * t.shape[3], t.shape[1])
    to_img = lambda t, hh, ww: t.reshape(t.shape[0], hh, ww, t.shape[2]).permute(0, 3, 1, 2)
    pos = [0, 0]                                          # (block, layer) being executed

    def stage(num, tok):                                  # tok [B, n, width]
        key = (pos[0], 16 * pos[1] + num if num else 0)
        if taps is not None:
            taps[key] = tok
        if forced is not None and key in forced:
            return forced[key].float().reshape(tok.shape)
        return tok

    def stage_img(num, img):
        return to_img(stage(num, to_tok(img)), img.shape[2], img.shape[3])

    # ---- time embedding (unet_body: temb -> e1 -> semb = SiLU(emb), all bf16; the 22 emb_layers rows in fp32)
    temb = bf(timestep_embedding(timesteps, mc))
    e1 = bf(F.silu(F.linear(temb, Wb("time_embed.0.weight"), sd["time_embed.0.bias"])))
    semb = bf(F.silu(F.linear(e1, Wb("time_embed.2.weight"), sd["time_embed.2.bias"])))
    ctx_b = bf(context.float())

    def resblock(pre, xin):
        h = stage_img(1, bf(F.silu(group_norm(xin, sd[pre + ".in_layers.0.weight"], sd[pre + ".in_layers.0.bias"], 1e-5))))
        e = F.linear(semb, Wb(pre + ".emb_layers.1.weight"), sd[pre + ".emb_layers.1.bias"])
        h = stage_img(2, bf(conv3(h, pre + ".in_layers.2.weight") + sd[pre + ".in_layers.2.bias"][None, :, None, None] + e[:, :, None, None]))
        h = stage_img(3, bf(F.silu(group_norm(h, sd[pre + ".out_layers.0.weight"], sd[pre + ".out_layers.0.bias"], 1e-5))))
        res = xin
        if (pre + ".skip_connection.weight") in sd:
            res = stage_img(4, bf(F.conv2d(xin, Wb(pre + ".skip_connection.weight"), sd[pre + ".skip_connection.bias"])))
        return stage_img(5, bf(conv3(h, pre + ".out_layers.3.weight") + sd[pre + ".out_layers.3.bias"][None, :, None, None] + res))

    def transformer(pre, xin, heads):
        b, c, hh, ww = xin.shape
        n, d = hh * ww, c // heads
        scale = d ** -0.5
        tb = pre + ".transformer_blocks.0"
        xn = stage(1, to_tok(bf(group_norm(xin, sd[pre + ".norm.weight"], sd[pre + ".norm.bias"], 1e-6))))
        t0 = stage(2, bf(F.linear(xn, Wb(pre + ".proj_in.weight").reshape(c, c), sd[pre + ".proj_in.bias"])))
        ln = lambda t, nm: bf(F.layer_norm(t, (c,), sd[f"{tb}.{nm}.weight"], sd[f"{tb}.{nm}.bias"], 1e-5))
        # --- attn1: q | k | v in one projection, flash attention with bf16 probabilities
        l1 = stage(3, ln(t0, "norm1"))
        qkv = stage(4, torch.cat([bf(F.linear(l1, Wb(f"{tb}.attn1.{nm}.weight"))) for nm in ("to_q", "to_k", "to_v")], dim=-1))
        q, k_, v = qkv.split(c, dim=-1)
        sp = lambda t: t.reshape(b, n, heads, d).permute(0, 2, 1, 3)
        ao = stage(5, bf(flash_attention(sp(q), sp(k_), sp(v), scale, R).permute(0, 2, 1, 3).reshape(b, n, c)))
        acc1 = F.linear(ao, Wb(f"{tb}.attn1.to_out.0.weight"))
        bo1, bo2 = sd[f"{tb}.attn1.to_out.0.bias"], sd[f"{tb}.attn2.to_out.0.bias"]
        t1 = bf(acc1 + bo1 + t0)
        if ctx_rows < b:                                   # zero-neighbour rows: attn2 == b_o2 exactly, folded into attn1.to_out's start values:
            t1 = torch.cat([t1[:ctx_rows], bf(acc1[ctx_rows:] + (bo1 + bo2) + t0[ctx_rows:])])       # those rows of the t1 buffer already hold t2
        t1 = stage(6, t1)
        # --- attn2 over the k neighbours, re-associated per sample
        kn = ctx_b.shape[1]
        K = bf(F.linear(ctx_b, Wb(f"{tb}.attn2.to_k.weight")))                    # [b, k, c]
        V = bf(F.linear(ctx_b, Wb(f"{tb}.attn2.to_v.weight")))
        Ks = bf(K * scale).reshape(b, kn, heads, d)
        Wq = Wb(f"{tb}.attn2.to_q.weight").reshape(heads, d, c)                   # rows d of head h
        G = bf(torch.einsum("bjhd,hdc->bhjc", Ks, Wq)).reshape(b, heads * kn, c)  # row h*k + j
        Wo = Wb(f"{tb}.attn2.to_out.0.weight").reshape(c, heads, d)
        U = bf(torch.einsum("chd,bjhd->bchj", Wo, V.reshape(b, kn, heads, d))).reshape(b, c, heads * kn)
        l2 = ln(t1, "norm2")
        sc = torch.einsum("bnc,bjc->bnj", l2, G).reshape(b, n, heads, kn)
        pr = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
        pr = bf(pr / pr.sum(dim=-1, keepdim=True)).reshape(b, n, heads * kn)
        t2 = bf(torch.einsum("bnj,bcj->bnc", pr, U) + bo2 + t1)
        if ctx_rows < b:
            t2 = torch.cat([t2[:ctx_rows], t1[ctx_rows:]])
        t2 = stage(7, t2)
        # --- GEGLU feed-forward, then ff.net.2 and proj_out as one map of [ff | t2]
        l3 = stage(8, ln(t2, "norm3"))
        pp = F.linear(l3, Wb(f"{tb}.ff.net.0.proj.weight"), sd[f"{tb}.ff.net.0.proj.bias"])
        a, gate = pp.chunk(2, dim=-1)
        ff = stage(9, bf(a * F.gelu(gate)))
        w2, b2 = sd[f"{tb}.ff.net.2.weight"].double(), sd[f"{tb}.ff.net.2.bias"].double()
        wo, bo = sd[pre + ".proj_out.weight"].reshape(c, c).double(), sd[pre + ".proj_out.bias"].double()
        fw = bf(torch.cat([wo @ w2, wo], dim=1).float())
        fb = (wo @ b2 + bo).float()
        out = stage(10, bf(F.linear(torch.cat([ff, t2], dim=-1), fw, fb) + to_tok(xin)))
        return to_img(out, hh, ww)

    def run(name, layers, h):
        for j, l in enumerate(layers):
            pre = f"{name}.{j}"
            pos[1] = j
            if l[0] == "conv_in":
                h = bf(F.conv2d(h, sd[pre + ".weight"], sd[pre + ".bias"], padding=1))             # fp32 weights (conv_in_kernel)
            elif l[0] == "res":
                h = resblock(pre, h)
            elif l[0] == "st":
                h = transformer(pre, h, l[2])
            elif l[0] == "down":
                h = bf(F.conv2d(h, Wb(pre + ".op.weight"), sd[pre + ".op.bias"], stride=2, padding=1))
            elif l[0] == "up":
                h = bf(_upsample_conv(h, Wb(pre + ".conv.weight"), sd[pre + ".conv.bias"], R))
        return h

    hs = []
    h = x.float()
    for bi, (name, layers) in enumerate(spec.blocks):
        pos[0] = bi
        if name.startswith("output_blocks"):
            h = torch.cat([h, hs.pop()], dim=1)
        h = stage_img(0, run(name, layers, h))
        if name.startswith("input_blocks"):
            hs.append(h)
    h = bf(F.silu(group_norm(h, sd["out.0.weight"], sd["out.0.bias"], 1e-5)))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)                            # fp32 weights as a (hi, lo) bf16 pair
