"""fp32 CPU restatement of the reference UNet (oracle — test infrastructure only).

Functional (state-dict driven) so that the same key names drive the reference
classes, this oracle, and the HIP weight packer.

Follows:
  rdm/modules/diffusionmodules/openaimodel.py:17-33   TimestepEmbedSequential routing
  rdm/modules/diffusionmodules/openaimodel.py:66-317  UNetModel constructor (block table)
  rdm/modules/diffusionmodules/openaimodel.py:335-371 UNetModel.forward
  rdm/modules/attention.py:16-17, 20-74, 77-96, 122-196  Normalize / CrossAttention /
                                                       BasicTransformerBlock / SpatialTransformer
  [ldm, un-vendored, parity unpinned — SURVEY.md appendix A.1]
      timestep_embedding, GroupNorm32 (eps 1e-5), ResBlock, Downsample, Upsample,
      FeedForward/GEGLU (exact erf GELU)
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- spec
@dataclass
class UNetSpec:
    """Shapes of one UNetModel instance (openaimodel.py:66-129 arguments that matter
    for sampling with the shipped configs: use_spatial_transformer=True,
    transformer_depth=1, use_scale_shift_norm=False, resblock_updown=False)."""
    in_channels: int = 3
    out_channels: int = 3
    model_channels: int = 192
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (8, 4, 2)
    channel_mult: Tuple[int, ...] = (1, 2, 3, 5)
    num_head_channels: int = 32
    context_dim: int = 512
    # derived: list of (prefix, kind, cin, cout, heads)
    blocks: List[tuple] = field(default_factory=list)

    def __post_init__(self):
        self.blocks = _build_block_table(self)

    @property
    def time_embed_dim(self):
        return self.model_channels * 4


def shipped_spec() -> UNetSpec:
    """models/rdm/imagenet/config.yaml:36-59."""
    return UNetSpec()


def tiny_spec() -> UNetSpec:
    """Reduced config for fast CPU parity (same topology, fewer channels)."""
    return UNetSpec(model_channels=64, num_res_blocks=1, attention_resolutions=(2, 4),
                    channel_mult=(1, 2, 3), num_head_channels=32, context_dim=512)


def _build_block_table(s: UNetSpec):
    """Mirror of the constructor loops, openaimodel.py:144-305.

    Returns a list of top-level blocks; each is (name, [layers]) with layers =
    ("conv_in", cin, cout) | ("res", cin, cout) | ("st", ch, heads) |
    ("down", ch) | ("up", ch).
    """
    mc = s.model_channels
    out = []
    out.append(("input_blocks.0", [("conv_in", s.in_channels, mc)]))
    chans = [mc]
    ch, ds, idx = mc, 1, 1
    for level, mult in enumerate(s.channel_mult):
        for _ in range(s.num_res_blocks):
            layers = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in s.attention_resolutions:
                layers.append(("st", ch, ch // s.num_head_channels))
            out.append((f"input_blocks.{idx}", layers)); idx += 1
            chans.append(ch)
        if level != len(s.channel_mult) - 1:
            out.append((f"input_blocks.{idx}", [("down", ch)])); idx += 1
            chans.append(ch)
            ds *= 2
    out.append(("middle_block", [("res", ch, ch), ("st", ch, ch // s.num_head_channels), ("res", ch, ch)]))
    oidx = 0
    for level, mult in list(enumerate(s.channel_mult))[::-1]:
        for i in range(s.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in s.attention_resolutions:
                layers.append(("st", ch, ch // s.num_head_channels))
            if level and i == s.num_res_blocks:
                layers.append(("up", ch))
                ds //= 2
            out.append((f"output_blocks.{oidx}", layers)); oidx += 1
    return out


def param_shapes(s: UNetSpec) -> Dict[str, tuple]:
    """Every state_dict key of the reference UNetModel and its shape."""
    mc, ted = s.model_channels, s.time_embed_dim
    p: Dict[str, tuple] = {}
    p["time_embed.0.weight"] = (ted, mc); p["time_embed.0.bias"] = (ted,)
    p["time_embed.2.weight"] = (ted, ted); p["time_embed.2.bias"] = (ted,)
    for name, layers in s.blocks:
        for j, l in enumerate(layers):
            pre = f"{name}.{j}"
            if l[0] == "conv_in":
                p[pre + ".weight"] = (l[2], l[1], 3, 3); p[pre + ".bias"] = (l[2],)
            elif l[0] == "res":
                cin, cout = l[1], l[2]
                p[pre + ".in_layers.0.weight"] = (cin,); p[pre + ".in_layers.0.bias"] = (cin,)
                p[pre + ".in_layers.2.weight"] = (cout, cin, 3, 3); p[pre + ".in_layers.2.bias"] = (cout,)
                p[pre + ".emb_layers.1.weight"] = (cout, ted); p[pre + ".emb_layers.1.bias"] = (cout,)
                p[pre + ".out_layers.0.weight"] = (cout,); p[pre + ".out_layers.0.bias"] = (cout,)
                p[pre + ".out_layers.3.weight"] = (cout, cout, 3, 3); p[pre + ".out_layers.3.bias"] = (cout,)
                if cin != cout:
                    p[pre + ".skip_connection.weight"] = (cout, cin, 1, 1)
                    p[pre + ".skip_connection.bias"] = (cout,)
            elif l[0] == "st":
                c = l[1]
                p[pre + ".norm.weight"] = (c,); p[pre + ".norm.bias"] = (c,)
                p[pre + ".proj_in.weight"] = (c, c, 1, 1); p[pre + ".proj_in.bias"] = (c,)
                tb = pre + ".transformer_blocks.0"
                for a, cd in (("attn1", c), ("attn2", s.context_dim)):
                    p[f"{tb}.{a}.to_q.weight"] = (c, c)
                    p[f"{tb}.{a}.to_k.weight"] = (c, cd)
                    p[f"{tb}.{a}.to_v.weight"] = (c, cd)
                    p[f"{tb}.{a}.to_out.0.weight"] = (c, c); p[f"{tb}.{a}.to_out.0.bias"] = (c,)
                p[f"{tb}.ff.net.0.proj.weight"] = (8 * c, c); p[f"{tb}.ff.net.0.proj.bias"] = (8 * c,)
                p[f"{tb}.ff.net.2.weight"] = (c, 4 * c); p[f"{tb}.ff.net.2.bias"] = (c,)
                for n in ("norm1", "norm2", "norm3"):
                    p[f"{tb}.{n}.weight"] = (c,); p[f"{tb}.{n}.bias"] = (c,)
                p[pre + ".proj_out.weight"] = (c, c, 1, 1); p[pre + ".proj_out.bias"] = (c,)
            elif l[0] == "down":
                p[pre + ".op.weight"] = (l[1], l[1], 3, 3); p[pre + ".op.bias"] = (l[1],)
            elif l[0] == "up":
                p[pre + ".conv.weight"] = (l[1], l[1], 3, 3); p[pre + ".conv.bias"] = (l[1],)
    p["out.0.weight"] = (mc,); p["out.0.bias"] = (mc,)
    p["out.2.weight"] = (s.out_channels, mc, 3, 3); p["out.2.bias"] = (s.out_channels,)
    return p


def synth_state_dict(shapes: Dict[str, tuple], seed: int = 1234, norm_keys=("norm", "in_layers.0", "out_layers.0", "out.0", "ln_")):
    """SURVEY.md §8(d) synthetic weights: N(0,1)/sqrt(fan_in); norm gamma=1+0.1n, beta=0.1n;
    biases 0.1 n.  (No zero-init on proj_out / out convs, or the net degenerates.)"""
    rng = np.random.default_rng(seed)
    sd = {}
    for k in sorted(shapes):
        shp = shapes[k]
        is_norm = any(t in k for t in norm_keys) and len(shp) == 1
        if is_norm:
            v = rng.standard_normal(shp) * 0.1 + (1.0 if k.endswith("weight") else 0.0)
        elif len(shp) == 1:
            v = rng.standard_normal(shp) * 0.1
        else:
            fan_in = int(np.prod(shp[1:]))
            v = rng.standard_normal(shp) / math.sqrt(fan_in)
        sd[k] = torch.from_numpy(v.astype(np.float32))
    return sd


# ----------------------------------------------------------------------------- ops
def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """[ldm] util.timestep_embedding: cos first, then sin."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm(x, w, b, eps, groups=32):
    return F.group_norm(x.float(), groups, w, b, eps)


def resblock(sd, pre, x, emb):
    """[ldm] ResBlock._forward, use_scale_shift_norm=False, no up/down."""
    h = F.silu(group_norm(x, sd[pre + ".in_layers.0.weight"], sd[pre + ".in_layers.0.bias"], 1e-5))
    h = F.conv2d(h, sd[pre + ".in_layers.2.weight"], sd[pre + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[pre + ".emb_layers.1.weight"], sd[pre + ".emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = F.silu(group_norm(h, sd[pre + ".out_layers.0.weight"], sd[pre + ".out_layers.0.bias"], 1e-5))
    h = F.conv2d(h, sd[pre + ".out_layers.3.weight"], sd[pre + ".out_layers.3.bias"], padding=1)
    if (pre + ".skip_connection.weight") in sd:
        x = F.conv2d(x, sd[pre + ".skip_connection.weight"], sd[pre + ".skip_connection.bias"])
    return x + h


def cross_attention(sd, pre, x, context, heads):
    """rdm/modules/attention.py:42-74 (no mask, not causal)."""
    q = F.linear(x, sd[pre + ".to_q.weight"])
    ctx = x if context is None else context
    k = F.linear(ctx, sd[pre + ".to_k.weight"])
    v = F.linear(ctx, sd[pre + ".to_v.weight"])
    B, n, C = q.shape
    d = C // heads
    scale = d ** -0.5
    def split(t):
        return t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.permute(0, 2, 1, 3).reshape(B, n, C)
    return F.linear(out, sd[pre + ".to_out.0.weight"], sd[pre + ".to_out.0.bias"])


def feed_forward(sd, pre, x):
    """[ldm] FeedForward(glu=True): GEGLU -> Linear; exact erf GELU."""
    p = F.linear(x, sd[pre + ".net.0.proj.weight"], sd[pre + ".net.0.proj.bias"])
    a, gate = p.chunk(2, dim=-1)
    return F.linear(a * F.gelu(gate), sd[pre + ".net.2.weight"], sd[pre + ".net.2.bias"])


def basic_transformer_block(sd, pre, x, context, heads):
    """rdm/modules/attention.py:92-96."""
    x = cross_attention(sd, pre + ".attn1", F.layer_norm(x, x.shape[-1:], sd[pre + ".norm1.weight"], sd[pre + ".norm1.bias"]), None, heads) + x
    x = cross_attention(sd, pre + ".attn2", F.layer_norm(x, x.shape[-1:], sd[pre + ".norm2.weight"], sd[pre + ".norm2.bias"]), context, heads) + x
    x = feed_forward(sd, pre + ".ff", F.layer_norm(x, x.shape[-1:], sd[pre + ".norm3.weight"], sd[pre + ".norm3.bias"])) + x
    return x


def spatial_transformer(sd, pre, x, context, heads):
    """rdm/modules/attention.py:170-196 (dims=2, depth=1); Normalize eps 1e-6 (:16-17)."""
    b, c, h, w = x.shape
    x_in = x
    x = group_norm(x, sd[pre + ".norm.weight"], sd[pre + ".norm.bias"], 1e-6)
    x = F.conv2d(x, sd[pre + ".proj_in.weight"], sd[pre + ".proj_in.bias"])
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    x = basic_transformer_block(sd, pre + ".transformer_blocks.0", x, context, heads)
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    x = F.conv2d(x, sd[pre + ".proj_out.weight"], sd[pre + ".proj_out.bias"])
    return x + x_in


def unet_forward(sd, spec: UNetSpec, x, timesteps, context):
    """rdm/modules/diffusionmodules/openaimodel.py:335-371.

    x [B,Cin,H,W] f32, timesteps [B] int64, context [B,k,context_dim] f32 -> eps [B,Cout,H,W].
    """
    t_emb = timestep_embedding(timesteps, spec.model_channels)
    emb = F.linear(t_emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])

    def run(name, layers, h):
        for j, l in enumerate(layers):
            pre = f"{name}.{j}"
            if l[0] == "conv_in":
                h = F.conv2d(h, sd[pre + ".weight"], sd[pre + ".bias"], padding=1)
            elif l[0] == "res":
                h = resblock(sd, pre, h, emb)
            elif l[0] == "st":
                h = spatial_transformer(sd, pre, h, context, l[2])
            elif l[0] == "down":
                h = F.conv2d(h, sd[pre + ".op.weight"], sd[pre + ".op.bias"], stride=2, padding=1)
            elif l[0] == "up":
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = F.conv2d(h, sd[pre + ".conv.weight"], sd[pre + ".conv.bias"], padding=1)
        return h

    hs = []
    h = x.float()
    for name, layers in spec.blocks:
        if name.startswith("input_blocks"):
            h = run(name, layers, h)
            hs.append(h)
        elif name == "middle_block":
            h = run(name, layers, h)
        else:
            h = torch.cat([h, hs.pop()], dim=1)
            h = run(name, layers, h)
    h = F.silu(group_norm(h, sd["out.0.weight"], sd["out.0.bias"], 1e-5))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)
