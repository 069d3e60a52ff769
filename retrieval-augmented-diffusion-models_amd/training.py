"""Backward of the UNet's ResBlock on the native path (the first backward pieces of the training
step; reference: rdm/models/diffusion/ddpm.py:390-443 shared_step -> ldm p_losses -> autograd through UNetModel, whose ResBlock is
ldm's `h = conv(silu(gn(x))); h += linear(silu(emb)); h = conv(silu(gn(h))); return skip(x) + h`).

Every arithmetic step is a C-ABI call into librdm_hip (include/rdm_hip.h "backward"): conv dgrad through the forward conv kernel with
the flipped / transposed filter, conv wgrad as a pixel-reduction GEMM on the MFMA kernel, GroupNorm+SiLU backward, deterministic column
sums, GEMMs for the time-embedding projection.  torch is used for device memory only (allocation, reshapes / views, zero padding).
The feed-forward sub-block of BasicTransformerBlock (`x + ff(norm3(x))`, ldm attention.py FeedForward / GEGLU) has its forward and
backward here too (ff_forward / ff_backward).  Attention (ldm CrossAttention: softmax(q k^T scale) v per head) has an unfused forward / backward here as well (attention_forward /
attention_backward: scores materialised per (sample, head), batched GEMMs on the MFMA kernel).  SpatialTransformer (GroupNorm -> proj_in -> block -> proj_out + x) and an AdamW step (rdm_op_adamw) close the loop for a small
ResBlock + SpatialTransformer stack (training_step_demo).  The whole-UNet graph in training form (down / up sampling, stem / head
convs, time-embedding MLP backward), LitEma and the optimisation step live in training_unet.py; the bucketed gradient all-reduce in
parallel.average_gradients (RCCL through torch.distributed; rdm_comm_all_reduce_f32 for callers of the C ABI); the reference-surface
entry (`MinimalRETRODiffusion.training_step`: first-stage encode -> q_sample -> conditioning dropout -> loss -> step) in
models/diffusion/ddpm.py."""
import torch

from . import _lib


def _pad_rows(t, mult=64):
    """[M, K] -> [ceil(M / mult) * mult, K] with zero rows (the GEMM's contraction length must be a multiple of 64)."""
    m = t.shape[0]
    mp = (m + mult - 1) // mult * mult
    if mp == m:
        return t
    out = torch.zeros((mp, t.shape[1]), device=t.device, dtype=t.dtype)
    out[:m] = t
    return out


_UNFUSED_ATTENTION_BWD = False       # tests flip this to reach the materialised-score path


def _pad_keys(t, mult=64):
    """[B, m, C] -> [B, ceil(m / mult) * mult, C] with zero rows."""
    m = t.shape[1]
    mp = (m + mult - 1) // mult * mult
    if mp == m:
        return t
    out = torch.zeros((t.shape[0], mp, t.shape[2]), device=t.device, dtype=t.dtype)
    out[:, :m] = t
    return out


def linear_backward(ctx, a, w, dy, bias=True, acc=None):
    """y = a w^T + b  (a [M,K], w [N,K], dy [M,N], all bf16)  ->  da bf16 [M,K] (+ acc when given: another path's gradient w.r.t. the same
    input joins in the GEMM's read-out), dw f32 [N,K], db f32 [N] (None for bias=False)."""
    da = ctx.op_linear(dy, ctx.op_transpose(w), residual=acc)                          # dy [M,N] . (w^T)^T
    dw = ctx.op_linear_wgrad(dy, a)                                                    # dy^T a, K-split over the M rows
    return da, dw, (ctx.op_colsum(dy) if bias else None)


def resblock_forward(ctx, p, x, semb):
    """x bf16 [B,H,W,Cin], semb = silu(time embedding) bf16 [B,E].  p: gn1_g/gn1_b/gn2_g/gn2_b f32, w1 [Cout,3,3,Cin] / w2 bf16, b1 / b2 f32,
    emb_w bf16 [Cout,E], emb_b f32, optional skip_w bf16 [Cout,Cin] / skip_b.  -> (out, saved activations)"""
    B, H, W, Cin = x.shape
    n1 = ctx.op_groupnorm(x.reshape(B, H * W, Cin), p["gn1_g"], p["gn1_b"], 1e-5, 1).reshape(B, H, W, Cin)
    emb_out = ctx.op_linear(semb, p["emb_w"], p["emb_b"], out_f32=True)                # [B, Cout] f32 (the conv adds it per sample)
    h1 = ctx.op_conv3x3(n1, p["w1"], p["b1"], rowvec=emb_out)
    Cout = h1.shape[3]
    n2 = ctx.op_groupnorm(h1.reshape(B, H * W, Cout), p["gn2_g"], p["gn2_b"], 1e-5, 1).reshape(B, H, W, Cout)
    if "skip_w" in p:
        res = ctx.op_linear(x.reshape(B * H * W, Cin), p["skip_w"], p["skip_b"]).reshape(B, H, W, Cout)
    else:
        res = x
    out = ctx.op_conv3x3(n2, p["w2"], p["b2"], residual=res)
    return out, {"n1": n1, "h1": h1, "n2": n2}


def resblock_backward(ctx, p, x, semb, saved, dout):
    """Gradients of `resblock_forward` given dout (bf16 [B,H,W,Cout]): -> dict with dx, dsemb (bf16) and fp32 parameter gradients."""
    B, H, W, Cin = x.shape
    Cout = dout.shape[3]
    HW, M = H * W, B * H * W
    g = {}
    dflat = dout.reshape(M, Cout)
    # out = conv2(n2) + b2 + res
    g["w2"] = ctx.op_conv3x3_wgrad(saved["n2"], dout)
    g["b2"] = ctx.op_colsum(dflat)
    dn2 = ctx.op_conv3x3_dgrad(dout, p["w2"])
    # n2 = silu(gn2(h1))
    dh1, g["gn2_g"], g["gn2_b"] = ctx.op_groupnorm_bwd(saved["h1"].reshape(B, HW, Cout), dn2.reshape(B, HW, Cout), p["gn2_g"], p["gn2_b"], 1e-5, 1)
    dh1 = dh1.reshape(B, H, W, Cout)
    # h1 = conv1(n1) + b1 + emb_out[b]
    g["w1"] = ctx.op_conv3x3_wgrad(saved["n1"], dh1)
    g["b1"] = ctx.op_colsum(dh1.reshape(M, Cout))
    demb = ctx.op_colsum_samples(dh1.reshape(B, HW, Cout))                             # [B, Cout]: sum over a sample's pixels, one launch
    g["dsemb"], g["emb_w"], g["emb_b"] = linear_backward(ctx, semb, p["emb_w"], demb)
    dn1 = ctx.op_conv3x3_dgrad(dh1, p["w1"])
    # skip path first: its gradient joins the GroupNorm gradient inside the GroupNorm backward kernel
    if "skip_w" in p:
        dxs, g["skip_w"], g["skip_b"] = linear_backward(ctx, x.reshape(M, Cin), p["skip_w"], dflat)
    else:
        dxs = dflat
    # n1 = silu(gn1(x))
    dx, g["gn1_g"], g["gn1_b"] = ctx.op_groupnorm_bwd(x.reshape(B, HW, Cin), dn1.reshape(B, HW, Cin), p["gn1_g"], p["gn1_b"], 1e-5, 1,
                                                      residual=dxs.reshape(B, HW, Cin).contiguous())
    g["dx"] = dx.reshape(B, H, W, Cin)
    return g


def ff_forward(ctx, p, x):
    """BasicTransformerBlock's feed-forward residual branch: out = x + W2 (a * gelu(g)) + b2 with [a | g] = W1 LayerNorm(x) + b1
    (ldm attention.py: `x = self.ff(self.norm3(x)) + x`, FeedForward(glu=True) = GEGLU -> Dropout(0) -> Linear).
    x bf16 [M, C]; p: ln_g / ln_b f32 [C], w1 bf16 [2F, C] (rows [x | gate], UNPERMUTED), b1 f32 [2F], w2 bf16 [C, F], b2 f32 [C].
    -> (out bf16 [M, C], saved activations)"""
    ln = ctx.op_layernorm(x, p["ln_g"], p["ln_b"])
    pre = ctx.op_linear(ln, p["w1"], p["b1"])                                         # [M, 2F]
    h = ctx.op_geglu(pre)                                                               # [M, F]
    out = ctx.op_linear(h, p["w2"], p["b2"], residual=x)
    return out, {"ln": ln, "pre": pre, "h": h}


def ff_backward(ctx, p, x, saved, dout):
    """Gradients of `ff_forward` given dout (bf16 [M, C]) -> dx bf16 and fp32 parameter gradients (w1, b1, w2, b2, ln_g, ln_b)."""
    g = {}
    dh, g["w2"], g["b2"] = linear_backward(ctx, saved["h"], p["w2"], dout)              # out = h w2^T + b2 (+ x)
    dpre = ctx.op_geglu(saved["pre"], dh)                                               # [da | dg]
    dln, g["w1"], g["b1"] = linear_backward(ctx, saved["ln"], p["w1"], dpre)
    g["x"], g["ln_g"], g["ln_b"] = ctx.op_layernorm_bwd(x, dln, p["ln_g"], residual=dout.contiguous())      # + the residual path, in the kernel
    return g


def attention_forward(ctx, q, k, v, heads):
    """ldm CrossAttention core (attention.py:52-72): per head softmax(q k^T / sqrt(d)) v.  q [B, n, C], k / v [B, m, C] bf16 (already
    projected), C = heads * d, d <= 64, n and m multiples of 4 -> (out [B, n, C] bf16, saved)."""
    B, n, C = q.shape
    d = C // heads
    scale = d ** -0.5
    m = k.shape[1]
    if d == 32 and m == n and n % 64 == 0 and not _UNFUSED_ATTENTION_BWD:
        # self-attention at the UNet's head width: the sampling path's flash kernel on [q | k | v] (V token-major, transposed inside
        # the kernel's LDS reads); the fused backward needs only q, k, v, the output and its gradient
        out = ctx.op_self_attention_qkv(torch.cat([q, k, v], dim=-1), heads)
        return out, {"o": out}
    if d == 32 and m <= 32 and not _UNFUSED_ATTENTION_BWD:
        # cross-attention on a handful of conditioning tokens: one thread per query row, K / V in LDS (the sampling path's kernel); the
        # backward recomputes the probabilities from q and K
        out = ctx.op_small_attention(q.contiguous(), k.contiguous(), v.contiguous(), heads, 32, False, scale)
        return out, {"small": True}
    k, v = _pad_keys(k), _pad_keys(v)                                                   # key count -> multiple of 64 (a GEMM K unit); padding gets probability 0
    qp, kp = ctx.op_heads(q, heads, d, 0), ctx.op_heads(k, heads, d, 0)                # [BH, n|mp, 64]
    vt = ctx.op_heads(v, heads, d, 1)                                                   # [BH, 64, mp]
    s = ctx.op_bmm(qp, kp, alpha=scale, out_f32=True)                                   # [BH, n, mp] fp32
    p = ctx.op_softmax(s, n_valid=m)
    o = ctx.op_bmm(p, vt)                                                               # [BH, n, 64]
    out = ctx.op_heads(o, heads, d, 2)
    return out, {"p": p, "kpad": k, "vpad": v, "m": m, "o": out}


def attention_backward(ctx, q, k, v, heads, saved, dout):
    """Gradients of `attention_forward`: dV = P^T dO, dP = dO V^T, dS = P (dP - rowsum(P dP)), dQ = dS K scale, dK = dS^T Q scale."""
    B, n, C = q.shape
    d = C // heads
    scale = d ** -0.5
    if d == 32 and n % 32 == 0 and k.shape[1] % 32 == 0 and "o" in saved and not _UNFUSED_ATTENTION_BWD:
        dq, dk, dv = ctx.op_attention_bwd(q.contiguous(), k.contiguous(), v.contiguous(), saved["o"], dout.contiguous(), heads)     # fused: no score matrix
        return {"q": dq, "k": dk, "v": dv}
    if saved.get("small"):
        dq, dk, dv = ctx.op_small_attention_bwd(q.contiguous(), k.contiguous(), v.contiguous(), dout.contiguous(), heads, scale)
        return {"q": dq, "k": dk, "v": dv}
    p, k, v, m = saved["p"], saved["kpad"], saved["vpad"], saved["m"]
    dop, dot_ = ctx.op_heads(dout, heads, d, 0), ctx.op_heads(dout, heads, d, 1)        # [BH, n, 64], [BH, 64, n]
    vp = ctx.op_heads(v, heads, d, 0)                                                   # [BH, m, 64]
    dv = ctx.op_bmm(ctx.op_transpose_batched(p), dot_)                                  # [BH, m, 64]
    dp = ctx.op_bmm(dop, vp, out_f32=True)                                              # [BH, n, m]
    ds = ctx.op_softmax_bwd(p, dp)
    kt, qt = ctx.op_heads(k, heads, d, 1), ctx.op_heads(q, heads, d, 1)                 # [BH, 64, m], [BH, 64, n]
    dq = ctx.op_bmm(ds, kt, alpha=scale)                                                # [BH, n, 64]
    dk = ctx.op_bmm(ctx.op_transpose_batched(ds), qt, alpha=scale)                      # [BH, m, 64]
    return {"q": ctx.op_heads(dq, heads, d, 2), "k": ctx.op_heads(dk, heads, d, 2)[:, :m].contiguous(), "v": ctx.op_heads(dv, heads, d, 2)[:, :m].contiguous()}


def attn_block_forward(ctx, p, x, context=None):
    """One attention residual branch of BasicTransformerBlock: out = x + to_out(attention(to_q(norm(x)), to_k(c), to_v(c))) with
    c = norm(x) (self-attention, attn1) or the conditioning `context` [B, m, Cc] (cross-attention, attn2); attention.py:84-96, 52-72.
    x bf16 [B, n, C]; p: ln_g / ln_b f32, wq [C, C], wk / wv [C, Cc] bf16 (no bias), wo [C, C] bf16, bo f32, heads."""
    B, n, C = x.shape
    ln = ctx.op_layernorm(x.reshape(B * n, C), p["ln_g"], p["ln_b"])
    c = ln if context is None else context.reshape(-1, context.shape[-1])
    m = n if context is None else context.shape[1]
    q = ctx.op_linear(ln, p["wq"]).reshape(B, n, C)
    k = ctx.op_linear(c, p["wk"]).reshape(B, m, C)
    v = ctx.op_linear(c, p["wv"]).reshape(B, m, C)
    att, saved = attention_forward(ctx, q, k, v, p["heads"])
    out = ctx.op_linear(att.reshape(B * n, C), p["wo"], p["bo"], residual=x.reshape(B * n, C)).reshape(B, n, C)
    saved.update({"ln": ln, "q": q, "k": k, "v": v, "att": att})
    return out, saved


def attn_block_backward(ctx, p, x, saved, dout, context=None):
    """-> dict: x (bf16), context (bf16, cross-attention only), fp32 wq / wk / wv / wo / bo / ln_g / ln_b."""
    B, n, C = x.shape
    g = {}
    dflat = dout.reshape(B * n, C)
    datt, g["wo"], g["bo"] = linear_backward(ctx, saved["att"].reshape(B * n, C), p["wo"], dflat)
    d = attention_backward(ctx, saved["q"], saved["k"], saved["v"], p["heads"], saved, datt.reshape(B, n, C))
    c = saved["ln"] if context is None else context.reshape(-1, context.shape[-1])
    m = saved["k"].shape[1]
    # the three projections' input gradients accumulate through the GEMMs' residual input (self-attention: all three read norm(x))
    dln, g["wq"], _ = linear_backward(ctx, saved["ln"], p["wq"], d["q"].reshape(B * n, C), bias=False)
    dck, g["wk"], _ = linear_backward(ctx, c, p["wk"], d["k"].reshape(B * m, C), bias=False, acc=dln if context is None else None)
    dc, g["wv"], _ = linear_backward(ctx, c, p["wv"], d["v"].reshape(B * m, C), bias=False, acc=dck)
    if context is None:
        dln = dc
    else:
        g["context"] = dc.reshape(context.shape)
    dx, g["ln_g"], g["ln_b"] = ctx.op_layernorm_bwd(x.reshape(B * n, C), dln, p["ln_g"], residual=dflat.contiguous())
    g["x"] = dx.reshape(B, n, C)
    return g


def transformer_block_forward(ctx, p, x, context):
    """BasicTransformerBlock.forward (attention.py:88-96): x = attn1(norm1(x)) + x; x = attn2(norm2(x), context) + x; x = ff(norm3(x)) + x.
    p = {"attn1": ..., "attn2": ..., "ff": ...} (see attn_block_forward / ff_forward)."""
    B, n, C = x.shape
    x1, s1 = attn_block_forward(ctx, p["attn1"], x)
    x2, s2 = attn_block_forward(ctx, p["attn2"], x1, context)
    x3, s3 = ff_forward(ctx, p["ff"], x2.reshape(B * n, C))
    return x3.reshape(B, n, C), {"x1": x1, "x2": x2, "attn1": s1, "attn2": s2, "ff": s3}


def transformer_block_backward(ctx, p, x, context, saved, dout):
    B, n, C = x.shape
    gff = ff_backward(ctx, p["ff"], saved["x2"].reshape(B * n, C), saved["ff"], dout.reshape(B * n, C))
    g2 = attn_block_backward(ctx, p["attn2"], saved["x1"], saved["attn2"], gff["x"].reshape(B, n, C), context)
    g1 = attn_block_backward(ctx, p["attn1"], x, saved["attn1"], g2["x"])
    return {"x": g1["x"], "context": g2["context"], "attn1": g1, "attn2": g2, "ff": gff}


def spatial_transformer_forward(ctx, p, x, context):
    """SpatialTransformer.forward (attention.py:147-183, depth 1): h = proj_in(norm(x)) -> BasicTransformerBlock(h, context) -> proj_out(h) + x.
    x bf16 [B, H, W, C] (NHWC); p: gn_g / gn_b f32, win / wout bf16 [C, C] (the 1x1 convs), bin / bout f32, block = transformer-block params."""
    B, H, W, C = x.shape
    n = H * W
    xn = ctx.op_groupnorm(x.reshape(B, n, C), p["gn_g"], p["gn_b"], 1e-6, 0)                       # Normalize: GroupNorm(32, eps 1e-6), no SiLU
    h = ctx.op_linear(xn.reshape(B * n, C), p["win"], p["bin"]).reshape(B, n, C)
    hb, sb = transformer_block_forward(ctx, p["block"], h, context)
    out = ctx.op_linear(hb.reshape(B * n, C), p["wout"], p["bout"], residual=x.reshape(B * n, C)).reshape(B, H, W, C)
    return out, {"xn": xn, "h": h, "hb": hb, "block": sb}


def spatial_transformer_backward(ctx, p, x, context, saved, dout):
    B, H, W, C = x.shape
    n = H * W
    g = {}
    dflat = dout.reshape(B * n, C)
    dhb, g["wout"], g["bout"] = linear_backward(ctx, saved["hb"].reshape(B * n, C), p["wout"], dflat)
    gb = transformer_block_backward(ctx, p["block"], saved["h"], context, saved["block"], dhb.reshape(B, n, C))
    dxn, g["win"], g["bin"] = linear_backward(ctx, saved["xn"].reshape(B * n, C), p["win"], gb["x"].reshape(B * n, C))
    dx_gn, g["gn_g"], g["gn_b"] = ctx.op_groupnorm_bwd(x.reshape(B, n, C), dxn.reshape(B, n, C), p["gn_g"], p["gn_b"], 1e-6, 0,
                                                        residual=dflat.reshape(B, n, C).contiguous())
    g["x"] = dx_gn.reshape(B, H, W, C)
    g["context"] = gb["context"]
    g["block"] = gb
    return g


def flatten_params(p, prefix=""):
    """nested dict of tensors -> {dotted name: tensor} (ints such as `heads` skipped)."""
    out = {}
    for k, v in p.items():
        if isinstance(v, dict):
            out.update(flatten_params(v, prefix + k + "."))
        elif torch.is_tensor(v):
            out[prefix + k] = v
    return out


def training_step_demo(ctx, master, state, x, semb, context, target, step, lr=1e-3, weight_decay=1e-2):
    """One optimisation step of `ResBlock -> SpatialTransformer` under an MSE loss against `target` (the shape of ldm p_losses'
    `loss_simple` for a two-block network): forward, loss gradient, backward through both blocks, AdamW on the fp32 master weights.
    master: {"res": ..., "st": ...} fp32 tensors; state: {"m": {...}, "v": {...}} flat fp32 moments (zeros at step 1).  Weights the
    kernels read in bf16 are re-cast from the masters.  -> (loss value, flat fp32 gradients)"""
    def working(p):
        return {k: (working(v) if isinstance(v, dict) else (v.to(torch.bfloat16) if torch.is_tensor(v) and v.dim() >= 2 else v)) for k, v in p.items()}
    w = working(master)
    h, s_res = resblock_forward(ctx, w["res"], x, semb)
    y, s_st = spatial_transformer_forward(ctx, w["st"], h, context)
    diff = y.float() - target.float()
    loss = float((diff * diff).mean())
    dy = (diff * (2.0 / diff.numel())).to(torch.bfloat16)
    g_st = spatial_transformer_backward(ctx, w["st"], h, context, s_st, dy)
    g_res = resblock_backward(ctx, w["res"], x, semb, s_res, g_st["x"])
    grads = {**{"res." + k: v for k, v in flatten_params(g_res).items()}, **{"st." + k: v for k, v in flatten_params(g_st).items()}}
    flat = flatten_params(master)
    for name, pm in flat.items():
        if name not in grads:
            continue
        gfl = grads[name].float().reshape(pm.shape).contiguous()
        ctx.op_adamw(pm, gfl, state["m"][name], state["v"][name], step, lr=lr, weight_decay=weight_decay)
    return loss, grads
