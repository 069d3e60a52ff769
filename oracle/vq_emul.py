"""The LIBRARY's arithmetic for the VQ-f4 first-stage decode, restated on the CPU (oracle — test infrastructure only; the companion of
oracle/unet_emul.py, same purpose and method: fp32 math, one bf16 rounding wherever csrc/model.hip `vq_body` / `vq_trunk` store a bf16 tensor).

  * quantisation + post_quant_conv and conv_in in fp32 (vq_quantize_kernel, conv_in_kernel), conv_in's output rounded;
  * ResnetBlock: bf16(swish(GroupNorm eps 1e-6)), bf16 3x3 weights, bias / shortcut added in fp32, one rounding per conv;
  * AttnBlock: bf16 q, k and V^T (b_v added after P.V: rows of P sum to 1), fp32 scores, NORMALISED probabilities rounded to bf16
    (softmax_rows_kernel), bf16(P V + b_v), proj_out + residual in one rounding;
  * Upsample convs by output phase on pre-summed weights (oracle/unet_emul.py::_phase_weights);
  * norm_out + swish rounded, conv_out against fp32 weights, fp32 image.
With rounding=False it is exact algebra and must equal oracle/vqdecoder.py (tests/test_oracle_cpu.py).

Follows (through oracle/vqdecoder.py): ldm VQModelInterface.decode / Decoder / ResnetBlock / AttnBlock / Upsample, taming
VectorQuantizer2, reached from rdm/models/diffusion/ddpm.py:840."""
import torch
import torch.nn.functional as F

from .unet_emul import _R, _upsample_conv
from .vqdecoder import VQSpec, vq_quantize


def vq_decode_emulated(sd, spec: VQSpec, z, force_not_quantize=False, rounding=True, taps=None, forced=None):
    """taps: list receiving every layer's output [B, H*W, C] in execution order (conv_in, mid.block_1, mid.attn_1, mid.block_2, then per
    level its blocks (+ attn) and the upsample conv) -- the library shows the same tensors as rdm_debug_tap blocks 1000, 1001, ...;
    forced: {index: tensor} teacher forcing as in oracle/unet_emul.py."""
    R = _R(rounding)
    cnt = [0]

    def layer(img):
        tok = img.permute(0, 2, 3, 1).reshape(img.shape[0], img.shape[2] * img.shape[3], img.shape[1])
        i = cnt[0]; cnt[0] += 1
        if taps is not None:
            taps.append(tok)
        if forced is not None and i in forced:
            return forced[i].float().reshape(img.shape[0], img.shape[2], img.shape[3], img.shape[1]).permute(0, 3, 1, 2)
        return img

    bf = R.bf
    Wb = lambda k: bf(sd[k].float())
    gn = lambda x, pre: F.group_norm(x, 32, sd[pre + ".weight"], sd[pre + ".bias"], 1e-6)
    swish = lambda x: x * torch.sigmoid(x)

    def resnet(pre, x):
        h = bf(swish(gn(x, pre + ".norm1")))
        h = bf(F.conv2d(h, Wb(pre + ".conv1.weight"), sd[pre + ".conv1.bias"], padding=1))
        h = bf(swish(gn(h, pre + ".norm2")))
        res = x
        if (pre + ".nin_shortcut.weight") in sd:
            res = bf(F.conv2d(x, Wb(pre + ".nin_shortcut.weight"), sd[pre + ".nin_shortcut.bias"]))
        return bf(F.conv2d(h, Wb(pre + ".conv2.weight"), sd[pre + ".conv2.bias"], padding=1) + res)

    def attn(pre, x):
        b, c, hh, ww = x.shape
        n = hh * ww
        hn = bf(gn(x, pre + ".norm")).permute(0, 2, 3, 1).reshape(b, n, c)
        lin = lambda nm, bias=True: F.linear(hn, Wb(f"{pre}.{nm}.weight").reshape(c, c), sd[f"{pre}.{nm}.bias"] if bias else None)
        q, k, v = bf(lin("q")), bf(lin("k")), bf(lin("v", bias=False))
        s = torch.bmm(q, k.transpose(1, 2)) * (float(c) ** -0.5)
        p = bf(F.softmax(s, dim=2))
        ao = bf(torch.bmm(p, v) + sd[pre + ".v.bias"])
        out = F.linear(ao, Wb(pre + ".proj_out.weight").reshape(c, c), sd[pre + ".proj_out.bias"])
        return bf(out.reshape(b, hh, ww, c).permute(0, 3, 1, 2) + x)

    if not force_not_quantize:
        z, _ = vq_quantize(sd, z)
    h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])                      # fp32 (vq_quantize_kernel)
    h = layer(bf(F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)))        # fp32 weights (conv_in_kernel)
    h = layer(resnet("decoder.mid.block_1", h))
    if spec.mid_attn:
        h = layer(attn("decoder.mid.attn_1", h))
    h = layer(resnet("decoder.mid.block_2", h))
    for lvl in reversed(range(len(spec.ch_mult))):
        for i in range(spec.num_res_blocks + 1):
            h = layer(resnet(f"decoder.up.{lvl}.block.{i}", h))
            if f"decoder.up.{lvl}.attn.{i}.q.weight" in sd:
                h = layer(attn(f"decoder.up.{lvl}.attn.{i}", h))
        if lvl != 0:
            h = layer(bf(_upsample_conv(h, Wb(f"decoder.up.{lvl}.upsample.conv.weight"), sd[f"decoder.up.{lvl}.upsample.conv.bias"], R)))
    h = bf(swish(gn(h, "decoder.norm_out")))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
