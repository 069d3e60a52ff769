"""fp32 CPU restatement of the VQ-f4 first-stage decode (oracle — test infrastructure only).

PARITY UNPINNED: the source (ldm VQModelInterface / Decoder / ResnetBlock / AttnBlock,
taming VectorQuantizer2) is not vendored in the reference; restated from SURVEY.md
appendix A.3.  Call sites in the reference: rdm/models/diffusion/ddpm.py:840, 981;
configuration models/rdm/imagenet/config.yaml:60-80.
"""
import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class VQSpec:
    embed_dim: int = 3
    n_embed: int = 8192
    z_channels: int = 3
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 4)
    num_res_blocks: int = 2
    out_ch: int = 3
    resolution: int = 256
    mid_attn: bool = True
    attn_resolutions: Tuple[int, ...] = ()

    @property
    def z_res(self):
        return self.resolution // 2 ** (len(self.ch_mult) - 1)


def shipped_vq_spec():
    return VQSpec()


def vqgan_f16_spec():
    """taming VQModel of the RARM models (models/rarm/imagenet/dogs/config.yaml:28-51)."""
    return VQSpec(embed_dim=256, n_embed=16384, z_channels=256, ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, resolution=256,
                  attn_resolutions=(16,))


def tiny_vqgan_spec():
    return VQSpec(embed_dim=64, n_embed=512, z_channels=64, ch=64, ch_mult=(1, 2, 2), num_res_blocks=1, resolution=32, attn_resolutions=(8,))


def tiny_vq_spec():
    return VQSpec(n_embed=512, ch=64, ch_mult=(1, 2, 4), num_res_blocks=1, resolution=64)


def _res_shapes(p, pre, cin, cout):
    p[pre + ".norm1.weight"] = (cin,); p[pre + ".norm1.bias"] = (cin,)
    p[pre + ".conv1.weight"] = (cout, cin, 3, 3); p[pre + ".conv1.bias"] = (cout,)
    p[pre + ".norm2.weight"] = (cout,); p[pre + ".norm2.bias"] = (cout,)
    p[pre + ".conv2.weight"] = (cout, cout, 3, 3); p[pre + ".conv2.bias"] = (cout,)
    if cin != cout:
        p[pre + ".nin_shortcut.weight"] = (cout, cin, 1, 1); p[pre + ".nin_shortcut.bias"] = (cout,)


def vq_param_shapes(s: VQSpec) -> Dict[str, tuple]:
    """State-dict names per SURVEY A.3 (prefix `first_stage_model.` stripped)."""
    p: Dict[str, tuple] = {}
    p["quantize.embedding.weight"] = (s.n_embed, s.embed_dim)
    p["post_quant_conv.weight"] = (s.z_channels, s.embed_dim, 1, 1); p["post_quant_conv.bias"] = (s.z_channels,)
    nl = len(s.ch_mult)
    block_in = s.ch * s.ch_mult[-1]
    p["decoder.conv_in.weight"] = (block_in, s.z_channels, 3, 3); p["decoder.conv_in.bias"] = (block_in,)
    _res_shapes(p, "decoder.mid.block_1", block_in, block_in)
    if s.mid_attn:
        a = "decoder.mid.attn_1"
        p[a + ".norm.weight"] = (block_in,); p[a + ".norm.bias"] = (block_in,)
        for n in ("q", "k", "v", "proj_out"):
            p[f"{a}.{n}.weight"] = (block_in, block_in, 1, 1); p[f"{a}.{n}.bias"] = (block_in,)
    _res_shapes(p, "decoder.mid.block_2", block_in, block_in)
    curr_res = s.z_res
    for lvl in reversed(range(nl)):
        block_out = s.ch * s.ch_mult[lvl]
        for i in range(s.num_res_blocks + 1):
            _res_shapes(p, f"decoder.up.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
            if curr_res in s.attn_resolutions:
                a = f"decoder.up.{lvl}.attn.{i}"
                p[a + ".norm.weight"] = (block_in,); p[a + ".norm.bias"] = (block_in,)
                for n in ("q", "k", "v", "proj_out"):
                    p[f"{a}.{n}.weight"] = (block_in, block_in, 1, 1); p[f"{a}.{n}.bias"] = (block_in,)
        if lvl != 0:
            p[f"decoder.up.{lvl}.upsample.conv.weight"] = (block_in, block_in, 3, 3)
            p[f"decoder.up.{lvl}.upsample.conv.bias"] = (block_in,)
            curr_res *= 2
    p["decoder.norm_out.weight"] = (block_in,); p["decoder.norm_out.bias"] = (block_in,)
    p["decoder.conv_out.weight"] = (s.out_ch, block_in, 3, 3); p["decoder.conv_out.bias"] = (s.out_ch,)
    return p


def _gn(x, sd, pre):
    return F.group_norm(x, 32, sd[pre + ".weight"], sd[pre + ".bias"], 1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _resnet(sd, pre, x):
    h = F.conv2d(_swish(_gn(x, sd, pre + ".norm1")), sd[pre + ".conv1.weight"], sd[pre + ".conv1.bias"], padding=1)
    h = F.conv2d(_swish(_gn(h, sd, pre + ".norm2")), sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], padding=1)
    if (pre + ".nin_shortcut.weight") in sd:
        x = F.conv2d(x, sd[pre + ".nin_shortcut.weight"], sd[pre + ".nin_shortcut.bias"])
    return x + h


def _attn(sd, pre, x):
    """[ldm] AttnBlock: single head over h*w tokens, scale C^-0.5."""
    h_ = _gn(x, sd, pre + ".norm")
    q = F.conv2d(h_, sd[pre + ".q.weight"], sd[pre + ".q.bias"])
    k = F.conv2d(h_, sd[pre + ".k.weight"], sd[pre + ".k.bias"])
    v = F.conv2d(h_, sd[pre + ".v.weight"], sd[pre + ".v.bias"])
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k) * (int(c) ** -0.5)
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    h_ = F.conv2d(h_, sd[pre + ".proj_out.weight"], sd[pre + ".proj_out.bias"])
    return x + h_


def vq_quantize(sd, z):
    """[taming] VectorQuantizer2.forward (legacy ordering irrelevant for inference):
    argmin of ||z||^2 + ||e||^2 - 2 z.e ; first minimum on ties. Returns (z_q, indices)."""
    e = sd["quantize.embedding.weight"]
    zf = z.permute(0, 2, 3, 1).contiguous()
    flat = zf.reshape(-1, e.shape[1])
    d = (flat ** 2).sum(1, keepdim=True) + (e ** 2).sum(1) - 2 * flat @ e.t()
    idx = torch.argmin(d, dim=1)
    zq = e[idx].reshape(zf.shape)
    zq = zf + (zq - zf)  # straight-through form, reproduced literally (A.3)
    return zq.permute(0, 3, 1, 2).contiguous(), idx


def vq_decode(sd, spec: VQSpec, z, scale_factor=1.0, force_not_quantize=False, return_indices=False):
    """decode_first_stage -> VQModelInterface.decode (A.3)."""
    z = z / scale_factor
    idx = None
    if not force_not_quantize:
        z, idx = vq_quantize(sd, z)
    out = _decode_quantized(sd, spec, z)
    return (out, idx) if return_indices else out


def vq_decode_indices(sd, spec: VQSpec, indices):
    """[taming] Net2NetTransformer.decode_to_img: quantize.get_codebook_entry(index, (b,h,w,c)) -> VQModel.decode
    (post_quant_conv + Decoder); reached from rdm/models/autoregression/transformer.py:296-312.  indices int64 [b, h*w]."""
    b = indices.shape[0]
    zq = sd["quantize.embedding.weight"][indices.reshape(-1)].reshape(b, spec.z_res, spec.z_res, spec.embed_dim).permute(0, 3, 1, 2).contiguous()
    return _decode_quantized(sd, spec, zq)


def _decode_quantized(sd, spec: VQSpec, z):
    h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _resnet(sd, "decoder.mid.block_1", h)
    if spec.mid_attn:
        h = _attn(sd, "decoder.mid.attn_1", h)
    h = _resnet(sd, "decoder.mid.block_2", h)
    for lvl in reversed(range(len(spec.ch_mult))):
        for i in range(spec.num_res_blocks + 1):
            h = _resnet(sd, f"decoder.up.{lvl}.block.{i}", h)
            if f"decoder.up.{lvl}.attn.{i}.q.weight" in sd:
                h = _attn(sd, f"decoder.up.{lvl}.attn.{i}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"decoder.up.{lvl}.upsample.conv.weight"], sd[f"decoder.up.{lvl}.upsample.conv.bias"], padding=1)
    h = _swish(_gn(h, sd, "decoder.norm_out"))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


# ------------------------------------------------------------------------------------------------ encoder side (round 4)
# PARITY UNPINNED like the decoder: ldm Encoder / Downsample / VQModelInterface.encode are not vendored in the reference; restated from
# the published ldm code (ldm/modules/diffusionmodules/model.py class Encoder, ldm/models/autoencoder.py VQModelInterface.encode:
# `h = self.encoder(x); h = self.quant_conv(h); return h`).  Call site in the reference: MinimalRETRODiffusion.shared_step ->
# get_input -> encode_first_stage (rdm/models/diffusion/ddpm.py:390-391), under torch.no_grad().

def vq_encoder_param_shapes(s: VQSpec) -> Dict[str, tuple]:
    """`encoder.*` and `quant_conv.*` of the first-stage state dict (in_channels = out_ch, double_z = False)."""
    p: Dict[str, tuple] = {}
    p["encoder.conv_in.weight"] = (s.ch, s.out_ch, 3, 3); p["encoder.conv_in.bias"] = (s.ch,)
    block_in, curr_res = s.ch, s.resolution
    for lvl in range(len(s.ch_mult)):
        block_out = s.ch * s.ch_mult[lvl]
        for i in range(s.num_res_blocks):
            _res_shapes(p, f"encoder.down.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
            if curr_res in s.attn_resolutions:
                a = f"encoder.down.{lvl}.attn.{i}"
                p[a + ".norm.weight"] = (block_in,); p[a + ".norm.bias"] = (block_in,)
                for n in ("q", "k", "v", "proj_out"):
                    p[f"{a}.{n}.weight"] = (block_in, block_in, 1, 1); p[f"{a}.{n}.bias"] = (block_in,)
        if lvl != len(s.ch_mult) - 1:
            p[f"encoder.down.{lvl}.downsample.conv.weight"] = (block_in, block_in, 3, 3)
            p[f"encoder.down.{lvl}.downsample.conv.bias"] = (block_in,)
            curr_res //= 2
    _res_shapes(p, "encoder.mid.block_1", block_in, block_in)
    if s.mid_attn:
        a = "encoder.mid.attn_1"
        p[a + ".norm.weight"] = (block_in,); p[a + ".norm.bias"] = (block_in,)
        for n in ("q", "k", "v", "proj_out"):
            p[f"{a}.{n}.weight"] = (block_in, block_in, 1, 1); p[f"{a}.{n}.bias"] = (block_in,)
    _res_shapes(p, "encoder.mid.block_2", block_in, block_in)
    p["encoder.norm_out.weight"] = (block_in,); p["encoder.norm_out.bias"] = (block_in,)
    p["encoder.conv_out.weight"] = (s.z_channels, block_in, 3, 3); p["encoder.conv_out.bias"] = (s.z_channels,)
    p["quant_conv.weight"] = (s.embed_dim, s.z_channels, 1, 1); p["quant_conv.bias"] = (s.embed_dim,)
    return p


def vq_encode(sd, spec: VQSpec, x, scale_factor=1.0):
    """encode_first_stage + get_first_stage_encoding for a VQModelInterface first stage: scale_factor * quant_conv(encoder(x)).
    [ldm] Downsample(with_conv): F.pad(x, (0, 1, 0, 1)) then Conv2d(kernel 3, stride 2, padding 0)."""
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for lvl in range(len(spec.ch_mult)):
        for i in range(spec.num_res_blocks):
            h = _resnet(sd, f"encoder.down.{lvl}.block.{i}", h)
            if f"encoder.down.{lvl}.attn.{i}.q.weight" in sd:
                h = _attn(sd, f"encoder.down.{lvl}.attn.{i}", h)
        if lvl != len(spec.ch_mult) - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, sd[f"encoder.down.{lvl}.downsample.conv.weight"], sd[f"encoder.down.{lvl}.downsample.conv.bias"], stride=2, padding=0)
    h = _resnet(sd, "encoder.mid.block_1", h)
    if spec.mid_attn:
        h = _attn(sd, "encoder.mid.attn_1", h)
    h = _resnet(sd, "encoder.mid.block_2", h)
    h = F.conv2d(_swish(_gn(h, sd, "encoder.norm_out")), sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return scale_factor * F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
