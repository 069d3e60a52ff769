"""Seeded random weights of the shipped architectures, keyed like the reference's state_dicts.

No checkpoint is reachable in the build/bench environment, so `bench.py` and `scripts/rdm_sample.py --synthetic`
run the real graphs on weights drawn from `numpy.random.default_rng(seed)` (N(0,1)/sqrt(fan_in), norm
gamma 1 + 0.1 n, beta / biases 0.1 n; no zero-init, or the network degenerates to the identity).  Shapes follow the
reference constructors:
    UNetModel.__init__                       rdm/modules/diffusionmodules/openaimodel.py:66-317
    SpatialTransformer / BasicTransformerBlock / CrossAttention   rdm/modules/attention.py:20-196
    CLIP.__init__                            rdm/modules/custom_clip/model.py:238-302
    ldm Decoder / VQModelInterface           (un-vendored; names per SURVEY appendix A.3, config models/rdm/imagenet/config.yaml:60-80)
and take the library's cfg structs (`_lib.make_unet_cfg()` etc., defaults = the shipped configs).
"""
import math
from typing import Dict

import numpy as np
import torch

_NORM_KEYS = ("norm", "in_layers.0", "out_layers.0", "out.0", "ln_")


def synth_state_dict(shapes: Dict[str, tuple], seed: int) -> Dict[str, torch.Tensor]:
    rng = np.random.default_rng(seed)
    sd = {}
    for k in sorted(shapes):
        shp = shapes[k]
        if len(shp) == 1 and any(t in k for t in _NORM_KEYS):
            v = rng.standard_normal(shp) * 0.1 + (1.0 if k.endswith("weight") else 0.0)
        elif len(shp) == 1:
            v = rng.standard_normal(shp) * 0.1
        else:
            v = rng.standard_normal(shp) / math.sqrt(int(np.prod(shp[1:])))
        sd[k] = torch.from_numpy(v.astype(np.float32))
    return sd


def unet_param_shapes(cfg) -> Dict[str, tuple]:
    mc, ted, cd = cfg.model_channels, cfg.model_channels * 4, cfg.context_dim
    mults = [cfg.channel_mult[i] for i in range(cfg.n_channel_mult)]
    attn = {cfg.attention_resolutions[i] for i in range(cfg.n_attention_resolutions)}
    p: Dict[str, tuple] = {"time_embed.0.weight": (ted, mc), "time_embed.0.bias": (ted,),
                           "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,)}

    def res(pre, cin, cout):
        p[pre + ".in_layers.0.weight"] = (cin,); p[pre + ".in_layers.0.bias"] = (cin,)
        p[pre + ".in_layers.2.weight"] = (cout, cin, 3, 3); p[pre + ".in_layers.2.bias"] = (cout,)
        p[pre + ".emb_layers.1.weight"] = (cout, ted); p[pre + ".emb_layers.1.bias"] = (cout,)
        p[pre + ".out_layers.0.weight"] = (cout,); p[pre + ".out_layers.0.bias"] = (cout,)
        p[pre + ".out_layers.3.weight"] = (cout, cout, 3, 3); p[pre + ".out_layers.3.bias"] = (cout,)
        if cin != cout:
            p[pre + ".skip_connection.weight"] = (cout, cin, 1, 1); p[pre + ".skip_connection.bias"] = (cout,)

    def st(pre, c):
        p[pre + ".norm.weight"] = (c,); p[pre + ".norm.bias"] = (c,)
        p[pre + ".proj_in.weight"] = (c, c, 1, 1); p[pre + ".proj_in.bias"] = (c,)
        tb = pre + ".transformer_blocks.0"
        for a, d in (("attn1", c), ("attn2", cd)):
            p[f"{tb}.{a}.to_q.weight"] = (c, c); p[f"{tb}.{a}.to_k.weight"] = (c, d); p[f"{tb}.{a}.to_v.weight"] = (c, d)
            p[f"{tb}.{a}.to_out.0.weight"] = (c, c); p[f"{tb}.{a}.to_out.0.bias"] = (c,)
        p[f"{tb}.ff.net.0.proj.weight"] = (8 * c, c); p[f"{tb}.ff.net.0.proj.bias"] = (8 * c,)
        p[f"{tb}.ff.net.2.weight"] = (c, 4 * c); p[f"{tb}.ff.net.2.bias"] = (c,)
        for n in ("norm1", "norm2", "norm3"):
            p[f"{tb}.{n}.weight"] = (c,); p[f"{tb}.{n}.bias"] = (c,)
        p[pre + ".proj_out.weight"] = (c, c, 1, 1); p[pre + ".proj_out.bias"] = (c,)

    p["input_blocks.0.0.weight"] = (mc, cfg.in_channels, 3, 3); p["input_blocks.0.0.bias"] = (mc,)
    chans, ch, ds, idx = [mc], mc, 1, 1
    for level, mult in enumerate(mults):
        for _ in range(cfg.num_res_blocks):
            res(f"input_blocks.{idx}.0", ch, mult * mc); ch = mult * mc
            if ds in attn:
                st(f"input_blocks.{idx}.1", ch)
            idx += 1; chans.append(ch)
        if level != len(mults) - 1:
            p[f"input_blocks.{idx}.0.op.weight"] = (ch, ch, 3, 3); p[f"input_blocks.{idx}.0.op.bias"] = (ch,)
            idx += 1; chans.append(ch); ds *= 2
    res("middle_block.0", ch, ch); st("middle_block.1", ch); res("middle_block.2", ch, ch)
    o = 0
    for level in reversed(range(len(mults))):
        for i in range(cfg.num_res_blocks + 1):
            res(f"output_blocks.{o}.0", ch + chans.pop(), mc * mults[level]); ch = mc * mults[level]
            j = 1
            if ds in attn:
                st(f"output_blocks.{o}.{j}", ch); j += 1
            if level and i == cfg.num_res_blocks:
                p[f"output_blocks.{o}.{j}.conv.weight"] = (ch, ch, 3, 3); p[f"output_blocks.{o}.{j}.conv.bias"] = (ch,)
                ds //= 2
            o += 1
    p["out.0.weight"] = (mc,); p["out.0.bias"] = (mc,)
    p["out.2.weight"] = (cfg.out_channels, mc, 3, 3); p["out.2.bias"] = (cfg.out_channels,)
    return p


def vq_param_shapes(cfg) -> Dict[str, tuple]:
    mults = [cfg.ch_mult[i] for i in range(cfg.n_ch_mult)]
    p: Dict[str, tuple] = {}
    if not cfg.kl:
        p["quantize.embedding.weight"] = (cfg.n_embed, cfg.embed_dim)
    p["post_quant_conv.weight"] = (cfg.z_channels, cfg.embed_dim, 1, 1); p["post_quant_conv.bias"] = (cfg.z_channels,)

    def res(pre, cin, cout):
        p[pre + ".norm1.weight"] = (cin,); p[pre + ".norm1.bias"] = (cin,)
        p[pre + ".conv1.weight"] = (cout, cin, 3, 3); p[pre + ".conv1.bias"] = (cout,)
        p[pre + ".norm2.weight"] = (cout,); p[pre + ".norm2.bias"] = (cout,)
        p[pre + ".conv2.weight"] = (cout, cout, 3, 3); p[pre + ".conv2.bias"] = (cout,)
        if cin != cout:
            p[pre + ".nin_shortcut.weight"] = (cout, cin, 1, 1); p[pre + ".nin_shortcut.bias"] = (cout,)

    def attn(a, ch):
        p[a + ".norm.weight"] = (ch,); p[a + ".norm.bias"] = (ch,)
        for n in ("q", "k", "v", "proj_out"):
            p[f"{a}.{n}.weight"] = (ch, ch, 1, 1); p[f"{a}.{n}.bias"] = (ch,)

    bin_ = cfg.ch * mults[-1]
    p["decoder.conv_in.weight"] = (bin_, cfg.z_channels, 3, 3); p["decoder.conv_in.bias"] = (bin_,)
    res("decoder.mid.block_1", bin_, bin_)
    if cfg.mid_attn:
        attn("decoder.mid.attn_1", bin_)
    res("decoder.mid.block_2", bin_, bin_)
    attn_res = {cfg.attn_resolutions[i] for i in range(cfg.n_attn_resolutions)}
    curr_res = cfg.resolution >> (len(mults) - 1)
    for lvl in reversed(range(len(mults))):
        bout = cfg.ch * mults[lvl]
        for i in range(cfg.num_res_blocks + 1):
            res(f"decoder.up.{lvl}.block.{i}", bin_, bout); bin_ = bout
            if curr_res in attn_res:
                attn(f"decoder.up.{lvl}.attn.{i}", bin_)
        if lvl != 0:
            p[f"decoder.up.{lvl}.upsample.conv.weight"] = (bin_, bin_, 3, 3); p[f"decoder.up.{lvl}.upsample.conv.bias"] = (bin_,)
            curr_res *= 2
    p["decoder.norm_out.weight"] = (bin_,); p["decoder.norm_out.bias"] = (bin_,)
    p["decoder.conv_out.weight"] = (cfg.out_ch, bin_, 3, 3); p["decoder.conv_out.bias"] = (cfg.out_ch,)
    return p


def clip_param_shapes(cfg) -> Dict[str, tuple]:
    p: Dict[str, tuple] = {}

    def tower(pre, w, layers):
        for i in range(layers):
            b = f"{pre}.resblocks.{i}"
            p[b + ".attn.in_proj_weight"] = (3 * w, w); p[b + ".attn.in_proj_bias"] = (3 * w,)
            p[b + ".attn.out_proj.weight"] = (w, w); p[b + ".attn.out_proj.bias"] = (w,)
            p[b + ".ln_1.weight"] = (w,); p[b + ".ln_1.bias"] = (w,)
            p[b + ".mlp.c_fc.weight"] = (4 * w, w); p[b + ".mlp.c_fc.bias"] = (4 * w,)
            p[b + ".mlp.c_proj.weight"] = (w, 4 * w); p[b + ".mlp.c_proj.bias"] = (w,)
            p[b + ".ln_2.weight"] = (w,); p[b + ".ln_2.bias"] = (w,)

    vw, ps, g = cfg.vision_width, cfg.vision_patch_size, cfg.image_resolution // cfg.vision_patch_size
    p["visual.conv1.weight"] = (vw, 3, ps, ps)
    p["visual.class_embedding"] = (vw,)
    p["visual.positional_embedding"] = (g * g + 1, vw)
    p["visual.ln_pre.weight"] = (vw,); p["visual.ln_pre.bias"] = (vw,)
    tower("visual.transformer", vw, cfg.vision_layers)
    p["visual.ln_post.weight"] = (vw,); p["visual.ln_post.bias"] = (vw,)
    p["visual.proj"] = (vw, cfg.embed_dim)
    tw = cfg.transformer_width
    tower("transformer", tw, cfg.transformer_layers)
    p["token_embedding.weight"] = (cfg.vocab_size, tw)
    p["positional_embedding"] = (cfg.context_length, tw)
    p["ln_final.weight"] = (tw,); p["ln_final.bias"] = (tw,)
    p["text_projection"] = (tw, cfg.embed_dim)
    return p


def rarm_param_shapes(cfg) -> Dict[str, tuple]:
    """RetrievalPatchTransformer(continuous=False, positional_encodings=True, cross_attend=True) — rdm/modules/attention.py:206-249."""
    C = cfg.n_heads * cfg.d_head
    p: Dict[str, tuple] = {"proj_in.weight": (cfg.vocab_in, C), "positional_encoding": (C, cfg.sequence_length),
                           "proj_out.weight": (cfg.vocab_out, C, 1), "proj_out.bias": (cfg.vocab_out,)}
    for i in range(cfg.depth):
        tb = f"transformer_blocks.{i}"
        for a, d in (("attn1", C), ("attn2", cfg.context_dim)):
            p[f"{tb}.{a}.to_q.weight"] = (C, C); p[f"{tb}.{a}.to_k.weight"] = (C, d); p[f"{tb}.{a}.to_v.weight"] = (C, d)
            p[f"{tb}.{a}.to_out.0.weight"] = (C, C); p[f"{tb}.{a}.to_out.0.bias"] = (C,)
        p[f"{tb}.ff.net.0.proj.weight"] = (8 * C, C); p[f"{tb}.ff.net.0.proj.bias"] = (8 * C,)
        p[f"{tb}.ff.net.2.weight"] = (C, 4 * C); p[f"{tb}.ff.net.2.bias"] = (C,)
        for n in ("norm1", "norm2", "norm3"):
            p[f"{tb}.{n}.weight"] = (C,); p[f"{tb}.{n}.bias"] = (C,)
    return p


UNET_SEED, VQ_SEED, CLIP_SEED, RARM_SEED, VQGAN_SEED = 1234, 4321, 99, 777, 888   # seeds of the committed golden fixtures


def rarm_state_dict(cfg, seed=RARM_SEED):
    return synth_state_dict(rarm_param_shapes(cfg), seed)


def unet_state_dict(cfg, seed=UNET_SEED):
    return synth_state_dict(unet_param_shapes(cfg), seed)


def vq_state_dict(cfg, seed=VQ_SEED):
    return synth_state_dict(vq_param_shapes(cfg), seed)


def clip_state_dict(cfg, seed=CLIP_SEED):
    return synth_state_dict(clip_param_shapes(cfg), seed)


def clip_like_rows(n, dim=512, seed=7, dtype=np.float16, scale=0.45):
    """Synthetic CLIP-like embedding rows (row norm ~ 10)."""
    return (np.random.default_rng(seed).standard_normal((n, dim)) * scale).astype(dtype)
