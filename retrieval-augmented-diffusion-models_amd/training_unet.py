"""The whole UNet in training form on the native ops: forward with saved activations, backward to every
parameter, for the reference's UNetModel topology (rdm/modules/diffusionmodules/openaimodel.py:144-371 as configured by the shipped
RDM configs: ResBlocks, SpatialTransformers of depth 1, strided-conv Downsample, nearest + conv Upsample, skip concatenations, the
time-embedding MLP, the GroupNorm + SiLU + conv head).  This is what autograd does under `MinimalRETRODiffusion.shared_step` ->
ldm `p_losses` in `main.py`'s training loop.

Every arithmetic step is a C-ABI call (rdm_amd._lib / rdm_amd.training); torch holds device memory and does layout plumbing only
(channel zero-padding of the 3-channel stem / head, concatenation / slicing along channels).  Since round 4 the sinusoidal timestep
table, the zero insertion that turns a stride-2 gradient into a stride-1 one, the nearest-neighbour copy in front of Upsample's weight
gradient, the per-sample bias gradient of the time-embedding rows and the loss with its gradient are HIP ops too, and the bf16
operands the kernels read are WORKING COPIES kept beside the fp32 masters (`TrainState.work`, refreshed by the optimiser kernel's bf16
output) instead of being re-cast on every forward.

Parameters live in a dict keyed by the reference's state-dict names, fp32 "master" tensors in the NATIVE layouts
(`params_from_state_dict`: 3x3 conv weights [Cout, 3, 3, Cin], 1x1 convs [Cout, Cin]); `grads_to_state_dict_layout` maps gradients back."""
import math

import torch

from . import parallel
from . import training as T


class TrainSpec:
    """Layer plan of one UNetModel, derived from an `rdm_unet_cfg` (`_lib.make_unet_cfg`).  `blocks` lists the top-level modules in
    state-dict order, each (name, [layers]) with layers ("conv_in", cin, cout) | ("res", cin, cout) | ("st", channels, heads) |
    ("down", channels) | ("up", channels): the order the reference's constructor appends them in (openaimodel.py:144-305: one
    ResBlock (+ SpatialTransformer where the downsampling factor is an attention resolution) per input block, a Downsample between
    levels, res-st-res in the middle, num_res_blocks + 1 output blocks per level consuming the skip stack in reverse, the last of a
    level carrying the Upsample)."""

    def __init__(self, cfg):
        self.in_channels, self.out_channels = int(cfg.in_channels), int(cfg.out_channels)
        self.model_channels, self.context_dim = int(cfg.model_channels), int(cfg.context_dim)
        mults = [int(cfg.channel_mult[i]) for i in range(cfg.n_channel_mult)]
        attn = {int(cfg.attention_resolutions[i]) for i in range(cfg.n_attention_resolutions)}
        nres, hc, mc = int(cfg.num_res_blocks), int(cfg.num_head_channels), self.model_channels

        def level_layers(cin, cout, factor):
            layers = [("res", cin, cout)]
            if factor in attn:
                layers.append(("st", cout, cout // hc))
            return layers

        blocks = [("input_blocks.0", [("conv_in", self.in_channels, mc)])]
        skips, width, factor = [mc], mc, 1
        for lvl, m in enumerate(mults):
            for _ in range(nres):
                blocks.append((f"input_blocks.{len(blocks)}", level_layers(width, m * mc, factor)))
                width = m * mc
                skips.append(width)
            if lvl + 1 < len(mults):
                blocks.append((f"input_blocks.{len(blocks)}", [("down", width)]))
                skips.append(width)
                factor *= 2
        blocks.append(("middle_block", [("res", width, width), ("st", width, width // hc), ("res", width, width)]))
        n_out = 0
        for lvl in range(len(mults) - 1, -1, -1):
            for i in range(nres + 1):
                layers = level_layers(width + skips.pop(), mults[lvl] * mc, factor)
                width = mults[lvl] * mc
                if lvl > 0 and i == nres:
                    layers.append(("up", width))
                    factor //= 2
                blocks.append((f"output_blocks.{n_out}", layers))
                n_out += 1
        self.blocks = blocks


def timestep_embedding(t, dim, max_period=10000.0):
    """ldm `timestep_embedding` (openaimodel util): [cos | sin] of t * exp(-log(max_period) * i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def params_from_state_dict(sd, device):
    """reference state dict (fp32, PyTorch layouts) -> native-layout fp32 masters on `device`."""
    out = {}
    for k, v in sd.items():
        v = v.detach().float()
        if v.dim() == 4 and v.shape[2:] == (3, 3):
            v = v.permute(0, 2, 3, 1)
        elif v.dim() == 4 and v.shape[2:] == (1, 1):
            v = v.reshape(v.shape[0], v.shape[1])
        out[k] = v.contiguous().to(device)
    return out


def grads_to_state_dict_layout(grads, sd):
    out = {}
    for k, g in grads.items():
        ref = sd[k]
        g = g.float()
        if ref.dim() == 4 and ref.shape[2:] == (3, 3):
            g = g.reshape(ref.shape[0], 3, 3, ref.shape[1]).permute(0, 3, 1, 2)
        out[k] = g.reshape(ref.shape)
    return out


def _bf(t):
    return t.to(torch.bfloat16)


class _Params:
    """Parameter lookup of the training graph: `P` the fp32 masters, `W` (optional) bf16 working copies of the >= 2-D tensors.  w(k) is
    what a kernel reads for a weight matrix (the working copy when there is one, else a cast of the master); 1-D tensors (biases, norm
    affines) are read in fp32 straight from the masters."""

    def __init__(self, P, W=None):
        self.P, self.W = P, W

    def __contains__(self, k): return k in self.P
    def __getitem__(self, k): return self.P[k]

    def w(self, k):
        if self.W is not None and k in self.W:
            return self.W[k]
        return _bf(self.P[k])


def _as_params(P):
    return P if isinstance(P, _Params) else _Params(P)


def _pad_channels(x, C):
    """[..., c] -> [..., C] with zero channels."""
    if x.shape[-1] == C:
        return x
    out = torch.zeros(x.shape[:-1] + (C,), device=x.device, dtype=x.dtype)
    out[..., :x.shape[-1]] = x
    return out


def _res_params(P, pre):
    P = _as_params(P)
    p = {"gn1_g": P[pre + ".in_layers.0.weight"], "gn1_b": P[pre + ".in_layers.0.bias"], "w1": P.w(pre + ".in_layers.2.weight"),
         "b1": P[pre + ".in_layers.2.bias"], "emb_w": P.w(pre + ".emb_layers.1.weight"), "emb_b": P[pre + ".emb_layers.1.bias"],
         "gn2_g": P[pre + ".out_layers.0.weight"], "gn2_b": P[pre + ".out_layers.0.bias"], "w2": P.w(pre + ".out_layers.3.weight"),
         "b2": P[pre + ".out_layers.3.bias"]}
    if pre + ".skip_connection.weight" in P:
        p["skip_w"] = P.w(pre + ".skip_connection.weight"); p["skip_b"] = P[pre + ".skip_connection.bias"]
    return p


_RES_NAMES = {"gn1_g": ".in_layers.0.weight", "gn1_b": ".in_layers.0.bias", "w1": ".in_layers.2.weight", "b1": ".in_layers.2.bias",
              "emb_w": ".emb_layers.1.weight", "emb_b": ".emb_layers.1.bias", "gn2_g": ".out_layers.0.weight", "gn2_b": ".out_layers.0.bias",
              "w2": ".out_layers.3.weight", "b2": ".out_layers.3.bias", "skip_w": ".skip_connection.weight", "skip_b": ".skip_connection.bias"}


def _attn_names(tb, a, n):
    return {"ln_g": f"{tb}.norm{n}.weight", "ln_b": f"{tb}.norm{n}.bias", "wq": f"{tb}.{a}.to_q.weight", "wk": f"{tb}.{a}.to_k.weight",
            "wv": f"{tb}.{a}.to_v.weight", "wo": f"{tb}.{a}.to_out.0.weight", "bo": f"{tb}.{a}.to_out.0.bias"}


def _st_names(pre):
    tb = pre + ".transformer_blocks.0"
    return {"gn_g": pre + ".norm.weight", "gn_b": pre + ".norm.bias", "win": pre + ".proj_in.weight", "bin": pre + ".proj_in.bias",
            "wout": pre + ".proj_out.weight", "bout": pre + ".proj_out.bias",
            "block": {"attn1": _attn_names(tb, "attn1", 1), "attn2": _attn_names(tb, "attn2", 2),
                      "ff": {"ln_g": tb + ".norm3.weight", "ln_b": tb + ".norm3.bias", "w1": tb + ".ff.net.0.proj.weight", "b1": tb + ".ff.net.0.proj.bias",
                             "w2": tb + ".ff.net.2.weight", "b2": tb + ".ff.net.2.bias"}}}


def _gather(P, names, heads=None):
    P = _as_params(P)
    out = {}
    for k, v in names.items():
        if isinstance(v, dict):
            out[k] = _gather(P, v, heads)
        else:
            t = P[v]
            out[k] = P.w(v) if t.dim() >= 2 else t
    if heads is not None and "wq" in out:
        out["heads"] = heads
    return out


def _scatter(grads, g, names):
    for k, v in names.items():
        if isinstance(v, dict):
            _scatter(grads, g[k], v)
        else:
            grads[v] = g[k]


def unet_train_forward(ctx, P, spec, x, timesteps, context):
    """x bf16 [B, H, W, in_channels] (NHWC), timesteps int64 [B], context bf16 [B, k, context_dim] -> (eps bf16 [B, H, W, out_channels], tape)."""
    P = _as_params(P)
    mc = spec.model_channels
    tape = {"layers": [], "x": x, "context": context}
    t_emb = ctx.op_timestep_embedding(timesteps.to(x.device).long().contiguous(), mc)          # ldm timestep_embedding, [cos | sin]
    e1 = ctx.op_linear(t_emb, P.w("time_embed.0.weight"), P["time_embed.0.bias"], out_f32=True)
    s1 = ctx.op_silu(e1)
    emb = ctx.op_linear(s1, P.w("time_embed.2.weight"), P["time_embed.2.bias"], out_f32=True)
    semb = ctx.op_silu(emb)
    tape.update({"t_emb": t_emb, "e1": e1, "s1": s1, "emb": emb, "semb": semb})

    def run(name, layers, h):
        for j, l in enumerate(layers):
            pre = f"{name}.{j}"
            if l[0] == "conv_in":
                xp = _pad_channels(h, 64)
                w = _bf(_pad_channels(P[pre + ".weight"], 64))
                tape["layers"].append(("conv_in", pre, xp, None))
                h = ctx.op_conv3x3(xp, w, P[pre + ".bias"])
            elif l[0] == "res":
                p = _res_params(P, pre)
                out, saved = T.resblock_forward(ctx, p, h, semb)
                tape["layers"].append(("res", pre, h, (p, saved)))
                h = out
            elif l[0] == "st":
                p = _gather(P, _st_names(pre), heads=l[2])
                out, saved = T.spatial_transformer_forward(ctx, p, h, context)
                tape["layers"].append(("st", pre, h, (p, saved)))
                h = out
            elif l[0] == "down":
                w = P.w(pre + ".op.weight")
                tape["layers"].append(("down", pre, h, w))
                h = ctx.op_conv3x3(h, w, P[pre + ".op.bias"], stride=2)
            elif l[0] == "up":
                w = P.w(pre + ".conv.weight")
                tape["layers"].append(("up", pre, h, w))
                h = ctx.op_conv3x3(h, w, P[pre + ".conv.bias"], ups=1)
        return h

    hs = []
    h = x
    for name, layers in spec.blocks:
        if name.startswith("input_blocks"):
            h = run(name, layers, h)
            tape["layers"].append(("push", name, None, len(hs)))
            hs.append(h)
        elif name == "middle_block":
            h = run(name, layers, h)
        else:
            skip = hs.pop()
            tape["layers"].append(("cat", name, None, (h.shape[-1], len(hs))))
            h = torch.cat([h, skip], dim=-1)
            h = run(name, layers, h)
    B, H, W, C = h.shape
    n = ctx.op_groupnorm(h.reshape(B, H * W, C), P["out.0.weight"], P["out.0.bias"], 1e-5, 1).reshape(B, H, W, C)
    wo = _bf(_pad_channels(P["out.2.weight"].permute(1, 2, 3, 0), 64).permute(3, 0, 1, 2).contiguous())       # pad the OUTPUT channels to 64
    bo = _pad_channels(P["out.2.bias"], 64)
    y = ctx.op_conv3x3(n, wo, bo)
    tape.update({"h_out": h, "n_out": n, "wo": wo})
    return y[..., :spec.out_channels].contiguous(), tape


def unet_train_backward(ctx, P, spec, tape, deps):
    """deps bf16 [B, H, W, out_channels] -> {state-dict name: fp32 gradient in the native layout}."""
    P = _as_params(P)
    grads = {}
    semb = tape["semb"]
    dsemb = None
    # head
    h, n, wo = tape["h_out"], tape["n_out"], tape["wo"]
    B, H, W, C = h.shape
    dyp = _pad_channels(deps, 64)
    grads["out.2.weight"] = ctx.op_conv3x3_wgrad(n, dyp)[:spec.out_channels].contiguous()
    grads["out.2.bias"] = ctx.op_colsum(dyp.reshape(-1, 64))[:spec.out_channels].contiguous()
    dn = ctx.op_conv3x3_dgrad(dyp, wo)
    dh, grads["out.0.weight"], grads["out.0.bias"] = ctx.op_groupnorm_bwd(h.reshape(B, H * W, C), dn.reshape(B, H * W, C), P["out.0.weight"], P["out.0.bias"], 1e-5, 1)
    d = dh.reshape(B, H, W, C)
    dskips = {}
    for kind, pre, xin, aux in reversed(tape["layers"]):
        if kind == "cat":
            c0, idx = aux
            dskips[idx] = d[..., c0:].contiguous()
            d = d[..., :c0].contiguous()
        elif kind == "push":
            if aux in dskips:
                d = ctx.op_add(d, dskips.pop(aux))
        elif kind == "res":
            p, saved = aux
            g = T.resblock_backward(ctx, p, xin, semb, saved, d)
            for k, suffix in _RES_NAMES.items():
                if k in g:
                    grads[pre + suffix] = g[k]
            dsemb = g["dsemb"].float() if dsemb is None else dsemb + g["dsemb"].float()     # [B, 4 mc]: a few hundred values per block
            d = g["dx"]
        elif kind == "st":
            p, saved = aux
            g = T.spatial_transformer_backward(ctx, p, xin, tape_context(tape), saved, d)
            _scatter(grads, g_named(g), _st_names(pre))
            d = g["x"]
        elif kind == "down":
            Bz, Hh, Wh, N = d.shape
            z = ctx.op_expand2(d, 0)                                                       # stride-2 gradient as a stride-1 one (zero insertion)
            grads[pre + ".op.weight"] = ctx.op_conv3x3_wgrad(xin, z)
            grads[pre + ".op.bias"] = ctx.op_colsum(d.reshape(-1, N))
            d = ctx.op_conv3x3_dgrad(z, aux)
        elif kind == "up":
            N = d.shape[-1]
            xu = ctx.op_expand2(xin, 1)                                                     # what the fused conv read (nearest 2x)
            grads[pre + ".conv.weight"] = ctx.op_conv3x3_wgrad(xu, d)
            grads[pre + ".conv.bias"] = ctx.op_colsum(d.reshape(-1, N))
            d = ctx.op_sumpool2(ctx.op_conv3x3_dgrad(d, aux))
        elif kind == "conv_in":
            N = d.shape[-1]
            grads[pre + ".weight"] = ctx.op_conv3x3_wgrad(xin, d)[..., :spec.in_channels].contiguous()
            grads[pre + ".bias"] = ctx.op_colsum(d.reshape(-1, N))
    # time-embedding MLP: emb = W2 silu(W0 t + b0) + b2; every ResBlock read silu(emb)
    demb = ctx.op_silu(tape["emb"], dy=dsemb.contiguous())
    ds1, grads["time_embed.2.weight"], grads["time_embed.2.bias"] = T.linear_backward(ctx, tape["s1"], P.w("time_embed.2.weight"), _bf(demb))
    de1 = ctx.op_silu(tape["e1"], dy=ds1.float().contiguous())
    _, grads["time_embed.0.weight"], grads["time_embed.0.bias"] = T.linear_backward(ctx, tape["t_emb"], P.w("time_embed.0.weight"), _bf(de1))
    return grads


def tape_context(tape):
    return tape["context"]


def g_named(g):
    """spatial_transformer_backward's gradient dict -> the nesting of _st_names (block.attn1 / attn2 / ff)."""
    blk = g["block"]
    return {"gn_g": g["gn_g"], "gn_b": g["gn_b"], "win": g["win"], "bin": g["bin"], "wout": g["wout"], "bout": g["bout"],
            "block": {"attn1": blk["attn1"], "attn2": blk["attn2"], "ff": blk["ff"]}}


def unet_loss_and_grads(ctx, P, spec, x, timesteps, context, target, coef=None):
    """ldm p_losses' `loss_simple` (mean squared error of the predicted noise) and its gradient w.r.t. every UNet parameter.
    target: bf16 / f32 NHWC [B,H,W,C] (as x) or f32 NCHW [B,C,H,W].  coef f32 [B] (optional): per-sample weight of (eps - target) in
    the gradient -- default 2 / (B C H W), the gradient of the plain batch mean.  -> (loss, grads, se) with se the per-sample means."""
    eps, tape = unet_train_forward(ctx, P, spec, x, timesteps, context)
    B, H, W, C = eps.shape
    if target.dim() == 4 and target.shape[1] == C and target.shape[-1] != C:
        tn = target.float().contiguous()
    else:
        tn = target.float().permute(0, 3, 1, 2).contiguous()
    if coef is None:
        coef = torch.full((B,), 2.0 / (B * C * H * W), device=eps.device, dtype=torch.float32)
    se, deps = ctx.op_mse_loss(eps, tn, coef)
    return float(se.mean()), unet_train_backward(ctx, P, spec, tape, deps), se


class TrainState:
    """What one optimisation step needs beside the batch: fp32 masters `P` (native layouts, `params_from_state_dict`), AdamW moments,
    the bf16 working copies `work` of every >= 2-D tensor (written by the optimiser kernel: rdm_op_adamw's p_bf16), the step counter
    and (optionally) LitEma's shadow weights."""

    def __init__(self, P, ema_decay=None):
        self.P = P
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}
        self.work = {k: v.to(torch.bfloat16) for k, v in P.items() if v.dim() >= 2}
        self.step = 0
        self.ema = Ema(P, ema_decay) if ema_decay else None

    def params(self):
        return _Params(self.P, self.work)


def unet_training_step(ctx, P, state, spec, x, timesteps, context, target, step=None, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                       coef=None):
    """One optimisation step of the UNet (what `trainer.fit` does per batch in the reference's main.py with ldm's AdamW): forward, MSE
    loss, backward, bucketed gradient all-reduce, AdamW on the fp32 masters in place.  `state`: a TrainState (bf16 working copies, no
    per-forward casts), or the round-3 form {"m": {...}, "v": {...}} with P the masters and `step` given.  -> loss before the update."""
    if isinstance(state, TrainState):
        loss, grads, _ = unet_loss_and_grads(ctx, state.params(), spec, x, timesteps, context, target, coef)
        apply_gradients(ctx, state, grads, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        return loss
    loss, grads, _ = unet_loss_and_grads(ctx, P, spec, x, timesteps, context, target, coef)
    grads = parallel.average_gradients(grads)
    for k, p in P.items():
        ctx.op_adamw(p, grads[k].float().reshape(p.shape).contiguous(), state["m"][k], state["v"][k], step, lr=lr, betas=betas, eps=eps,
                     weight_decay=weight_decay)
    return loss


def apply_gradients(ctx, state, grads, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, group=None):
    """The optimiser half of a step on a TrainState: bucketed gradient all-reduce over the data-parallel group (no-op on one rank),
    AdamW on the fp32 masters in place with the bf16 working copies refreshed by the same kernel, LitEma update."""
    state.step += 1
    grads = parallel.average_gradients(grads, group=group) if group is not None else parallel.average_gradients(grads)
    keys = list(state.P)
    gs = [grads[k].float().reshape(state.P[k].shape).contiguous() for k in keys]
    ctx.op_adamw_multi([state.P[k] for k in keys], gs, [state.m[k] for k in keys], [state.v[k] for k in keys], state.step, lr=lr, betas=betas, eps=eps,
                       weight_decay=weight_decay, p_bf16s=[state.work.get(k) for k in keys])          # 688 tensors in 15 launches
    if state.ema is not None:
        state.ema.update(ctx, state.P)


def state_dict_from_params(P, like):
    """native-layout masters -> tensors in the reference's state-dict layouts (`like`: name -> tensor or shape), on the host."""
    out = {}
    for k, v in P.items():
        shape = tuple(like[k].shape) if hasattr(like[k], "shape") else tuple(like[k])
        v = v.detach().float().cpu()
        if len(shape) == 4 and shape[2:] == (3, 3):
            v = v.reshape(shape[0], 3, 3, shape[1]).permute(0, 3, 1, 2)
        out[k] = v.reshape(shape).contiguous()
    return out


class Ema:
    """ldm LitEma (ldm/modules/ema.py): exponential moving average of the parameters with the warm-up decay
    min(decay, (1 + n) / (10 + n)); `MinimalRETRODiffusion` (like LatentDiffusion) samples from these weights (`use_ema`)."""

    def __init__(self, P, decay=0.9999):
        self.decay, self.num_updates = decay, 0
        self.shadow = {k: v.clone() for k, v in P.items()}

    def resume(self, shadow, num_updates):
        """Continue from a checkpoint's LitEma buffers (`model_ema.*`, `model_ema.num_updates`): shadows in the native layouts of P."""
        missing = set(self.shadow) - set(shadow)
        if missing:
            raise KeyError(f"Ema.resume: checkpoint shadows lack {sorted(missing)[:4]}...")
        for k in self.shadow:
            self.shadow[k].copy_(shadow[k])
        self.num_updates = int(num_updates)

    def update(self, ctx, P):
        self.num_updates += 1
        decay = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        keys = list(P)
        ctx.op_ema_multi([self.shadow[k] for k in keys], [P[k] for k in keys], 1.0 - decay)
