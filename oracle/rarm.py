"""fp32 CPU restatement of the RARM sampling path (oracle — test infrastructure only).

Follows:
  rdm/modules/attention.py:199-272   RetrievalPatchTransformer.forward (continuous=False: nn.Embedding proj_in,
                                     positional_encoding [inner_dim, seq] added per position, blocks, Conv1d(k=1) proj_out)
  rdm/modules/attention.py:77-96     BasicTransformerBlock (attn1 causal self-attention, attn2 cross-attention on the
                                     neighbours, GEGLU feed-forward)           — CrossAttention causal mask :58-65
  rdm/models/autoregression/transformer.py:224-294   LatentImageRETRO.sample (prefix re-run each step, CFG on logits with
                                     zero neighbours, temperature, top-k, softmax, multinomial)
  [taming, un-vendored]              Net2NetTransformer.top_k_logits: v, _ = topk(logits, k); logits[logits < v[..., -1:]] = -inf
PINNED by golden vectors generated from the in-tree RetrievalPatchTransformer class (tools/gen_golden.py, rarm_*.npz).
The multinomial draw is DEFINED here (and in the HIP path) as the inverse CDF in vocabulary order at an explicit uniform:
torch.multinomial's random stream is device-specific and has no portable definition.
"""
from dataclasses import dataclass
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .unet import cross_attention, feed_forward


@dataclass
class RarmSpec:
    """models/rarm/imagenet/dogs/config.yaml:14-27."""
    vocab_in: int = 16386
    vocab_out: int = 16384
    n_heads: int = 12
    d_head: int = 64
    depth: int = 18
    context_dim: int = 512
    sequence_length: int = 256

    @property
    def inner_dim(self):
        return self.n_heads * self.d_head


def shipped_rarm_spec():
    return RarmSpec()


def tiny_rarm_spec():
    return RarmSpec(vocab_in=1002, vocab_out=1000, n_heads=2, d_head=64, depth=2, context_dim=512, sequence_length=24)


def rarm_param_shapes(s: RarmSpec) -> Dict[str, tuple]:
    C = s.inner_dim
    p = {"proj_in.weight": (s.vocab_in, C), "positional_encoding": (C, s.sequence_length),
         "proj_out.weight": (s.vocab_out, C, 1), "proj_out.bias": (s.vocab_out,)}
    for i in range(s.depth):
        tb = f"transformer_blocks.{i}"
        for a, d in (("attn1", C), ("attn2", s.context_dim)):
            p[f"{tb}.{a}.to_q.weight"] = (C, C); p[f"{tb}.{a}.to_k.weight"] = (C, d); p[f"{tb}.{a}.to_v.weight"] = (C, d)
            p[f"{tb}.{a}.to_out.0.weight"] = (C, C); p[f"{tb}.{a}.to_out.0.bias"] = (C,)
        p[f"{tb}.ff.net.0.proj.weight"] = (8 * C, C); p[f"{tb}.ff.net.0.proj.bias"] = (8 * C,)
        p[f"{tb}.ff.net.2.weight"] = (C, 4 * C); p[f"{tb}.ff.net.2.bias"] = (C,)
        for n in ("norm1", "norm2", "norm3"):
            p[f"{tb}.{n}.weight"] = (C,); p[f"{tb}.{n}.bias"] = (C,)
    return p


def causal_self_attention(sd, pre, x, heads):
    """CrossAttention.forward with context=None, causal=True (attention.py:42-74)."""
    q = F.linear(x, sd[pre + ".to_q.weight"]); k = F.linear(x, sd[pre + ".to_k.weight"]); v = F.linear(x, sd[pre + ".to_v.weight"])
    B, n, C = q.shape
    d = C // heads
    sp = lambda t: t.reshape(B, n, heads, d).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5
    mask = torch.ones(n, n, dtype=torch.bool).triu(1)
    sim = sim.masked_fill(mask, -torch.finfo(sim.dtype).max)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v).permute(0, 2, 1, 3).reshape(B, n, C)
    return F.linear(out, sd[pre + ".to_out.0.weight"], sd[pre + ".to_out.0.bias"])


def rarm_forward(sd, spec: RarmSpec, tokens: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
    """tokens int64 [b,t], context f32 [b,k,ctx] -> logits f32 [b,t,vocab_out]."""
    t = tokens.shape[1]
    x = F.embedding(tokens, sd["proj_in.weight"])                              # [b,t,C]
    x = x + sd["positional_encoding"][:, :t].t()[None]
    for i in range(spec.depth):
        tb = f"transformer_blocks.{i}"
        ln = lambda n, y: F.layer_norm(y, y.shape[-1:], sd[f"{tb}.{n}.weight"], sd[f"{tb}.{n}.bias"])
        x = causal_self_attention(sd, tb + ".attn1", ln("norm1", x), spec.n_heads) + x
        x = cross_attention(sd, tb + ".attn2", ln("norm2", x), context, spec.n_heads) + x
        x = feed_forward(sd, tb + ".ff", ln("norm3", x)) + x
    return F.linear(x, sd["proj_out.weight"][:, :, 0], sd["proj_out.bias"])


def top_k_logits(logits, k):
    v, _ = torch.topk(logits, k)
    out = logits.clone()
    out[out < v[..., [-1]]] = -float("inf")
    return out


def draw(probs: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """Inverse CDF in vocabulary order: smallest index whose cumulative probability exceeds u * total."""
    c = probs.double().cumsum(dim=-1)
    idx = (c > (u.double() * c[..., -1])[..., None]).float().argmax(dim=-1)
    return idx


def rarm_sample(sd, spec: RarmSpec, cond_tokens, context, steps, uniforms, temperature=1.0, top_k=None, guidance_scale=1.0,
                forward=None):
    """LatentImageRETRO.sample (sample=True).  uniforms [steps,b].  `forward(tokens, context)` may be the reference class."""
    fwd = forward or (lambda tok, ctx: rarm_forward(sd, spec, tok, ctx))
    x = cond_tokens.clone()
    bs = x.shape[0]
    r = context
    if guidance_scale > 1.0:
        r = torch.cat((r, torch.zeros_like(r)), dim=0)
    all_logits = []
    for s in range(steps):
        xin = torch.cat((x, x), dim=0) if guidance_scale > 1.0 else x
        logits = fwd(xin, r)
        if guidance_scale > 1.0:
            logits = logits[bs:] + guidance_scale * (logits[:bs] - logits[bs:])
        logits = logits[:, -1, :] / temperature
        all_logits.append(logits)
        if top_k is not None:
            logits = top_k_logits(logits, top_k)
        probs = F.softmax(logits, dim=-1)
        ix = draw(probs, uniforms[s])
        x = torch.cat((x, ix[:, None]), dim=1)
    return x[:, cond_tokens.shape[1]:], torch.stack(all_logits, dim=1)
