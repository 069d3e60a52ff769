"""Shared helpers for the parity tests (oracle side = test infrastructure)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def spec_to_unet_cfg(spec):
    from rdm_amd import _lib
    return _lib.make_unet_cfg(in_channels=spec.in_channels, out_channels=spec.out_channels,
                              model_channels=spec.model_channels, num_res_blocks=spec.num_res_blocks,
                              attention_resolutions=spec.attention_resolutions, channel_mult=spec.channel_mult,
                              num_head_channels=spec.num_head_channels, context_dim=spec.context_dim)


def spec_to_vq_cfg(spec):
    from rdm_amd import _lib
    return _lib.make_vq_cfg(embed_dim=spec.embed_dim, n_embed=spec.n_embed, z_channels=spec.z_channels, ch=spec.ch,
                            ch_mult=spec.ch_mult, num_res_blocks=spec.num_res_blocks, out_ch=spec.out_ch,
                            resolution=spec.resolution, mid_attn=spec.mid_attn)


def spec_to_clip_cfg(spec):
    from rdm_amd import _lib
    return _lib.make_clip_cfg(**{k: getattr(spec, k) for k in (
        "embed_dim", "image_resolution", "vision_layers", "vision_width", "vision_patch_size", "context_length",
        "vocab_size", "transformer_width", "transformer_heads", "transformer_layers")})
