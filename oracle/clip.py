"""fp32 CPU restatement of the CLIP ViT-B/32 text and image towers
(oracle — test infrastructure only).

Follows rdm/modules/custom_clip/model.py:
  :152-163  LayerNorm (fp32), QuickGELU  x*sigmoid(1.702x)
  :166-187  ResidualAttentionBlock  (nn.MultiheadAttention: packed in_proj, scale d^-1/2 on q,
            additive -inf causal mask for text)
  :201-235  VisualTransformer.forward
  :307-320  CLIP.encode_text (EOT row = argmax of token ids, @ text_projection)
PINNED by golden vectors generated from the in-tree reference class (tools/gen_golden.py).
"""
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F


@dataclass
class ClipSpec:
    embed_dim: int = 512
    image_resolution: int = 224
    vision_layers: int = 12
    vision_width: int = 768
    vision_patch_size: int = 32
    context_length: int = 77
    vocab_size: int = 49408
    transformer_width: int = 512
    transformer_heads: int = 8
    transformer_layers: int = 12

    @property
    def vision_heads(self):
        return self.vision_width // 64


def vitb32_spec():
    return ClipSpec()


def tiny_clip_spec():
    return ClipSpec(embed_dim=64, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=32,
                    context_length=77, vocab_size=1000, transformer_width=128, transformer_heads=2,
                    transformer_layers=2)


def _tower_shapes(p, pre, width, layers):
    for i in range(layers):
        b = f"{pre}.resblocks.{i}"
        p[b + ".attn.in_proj_weight"] = (3 * width, width); p[b + ".attn.in_proj_bias"] = (3 * width,)
        p[b + ".attn.out_proj.weight"] = (width, width); p[b + ".attn.out_proj.bias"] = (width,)
        p[b + ".ln_1.weight"] = (width,); p[b + ".ln_1.bias"] = (width,)
        p[b + ".mlp.c_fc.weight"] = (4 * width, width); p[b + ".mlp.c_fc.bias"] = (4 * width,)
        p[b + ".mlp.c_proj.weight"] = (width, 4 * width); p[b + ".mlp.c_proj.bias"] = (width,)
        p[b + ".ln_2.weight"] = (width,); p[b + ".ln_2.bias"] = (width,)


def clip_param_shapes(s: ClipSpec) -> Dict[str, tuple]:
    p: Dict[str, tuple] = {}
    vw, g = s.vision_width, s.image_resolution // s.vision_patch_size
    p["visual.conv1.weight"] = (vw, 3, s.vision_patch_size, s.vision_patch_size)
    p["visual.class_embedding"] = (vw,)
    p["visual.positional_embedding"] = (g * g + 1, vw)
    p["visual.ln_pre.weight"] = (vw,); p["visual.ln_pre.bias"] = (vw,)
    _tower_shapes(p, "visual.transformer", vw, s.vision_layers)
    p["visual.ln_post.weight"] = (vw,); p["visual.ln_post.bias"] = (vw,)
    p["visual.proj"] = (vw, s.embed_dim)
    tw = s.transformer_width
    _tower_shapes(p, "transformer", tw, s.transformer_layers)
    p["token_embedding.weight"] = (s.vocab_size, tw)
    p["positional_embedding"] = (s.context_length, tw)
    p["ln_final.weight"] = (tw,); p["ln_final.bias"] = (tw,)
    p["text_projection"] = (tw, s.embed_dim)
    return p


def _ln(x, sd, pre):
    return F.layer_norm(x.float(), x.shape[-1:], sd[pre + ".weight"], sd[pre + ".bias"], 1e-5)


def _mha(sd, pre, x, heads, causal):
    """x [B,L,D]; torch nn.MultiheadAttention math."""
    B, L, D = x.shape
    d = D // heads
    qkv = F.linear(x, sd[pre + ".in_proj_weight"], sd[pre + ".in_proj_bias"])
    q, k, v = qkv.split(D, dim=-1)
    sp = lambda t: t.reshape(B, L, heads, d).permute(0, 2, 1, 3)
    q, k, v = sp(q) * (d ** -0.5), sp(k), sp(v)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    a = s.softmax(-1)
    o = (a @ v).permute(0, 2, 1, 3).reshape(B, L, D)
    return F.linear(o, sd[pre + ".out_proj.weight"], sd[pre + ".out_proj.bias"])


def _block(sd, pre, x, heads, causal):
    x = x + _mha(sd, pre + ".attn", _ln(x, sd, pre + ".ln_1"), heads, causal)
    h = F.linear(_ln(x, sd, pre + ".ln_2"), sd[pre + ".mlp.c_fc.weight"], sd[pre + ".mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)
    return x + F.linear(h, sd[pre + ".mlp.c_proj.weight"], sd[pre + ".mlp.c_proj.bias"])


def encode_text(sd, spec: ClipSpec, tokens: torch.Tensor) -> torch.Tensor:
    """model.py:307-320. tokens int64 [B,77] -> [B,embed_dim] (un-normalised)."""
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"]
    for i in range(spec.transformer_layers):
        x = _block(sd, f"transformer.resblocks.{i}", x, spec.transformer_heads, True)
    x = _ln(x, sd, "ln_final")
    x = x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ sd["text_projection"]
    return x


def encode_image(sd, spec: ClipSpec, image: torch.Tensor) -> torch.Tensor:
    """model.py:216-235. image [B,3,R,R] (already CLIP-normalised) -> [B,embed_dim]."""
    x = F.conv2d(image, sd["visual.conv1.weight"], stride=spec.vision_patch_size)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd["visual.class_embedding"] + torch.zeros(x.shape[0], 1, x.shape[-1])
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    x = _ln(x, sd, "visual.ln_pre")
    for i in range(spec.vision_layers):
        x = _block(sd, f"visual.transformer.resblocks.{i}", x, spec.vision_heads, False)
    x = _ln(x[:, 0, :], sd, "visual.ln_post")
    return x @ sd["visual.proj"]
