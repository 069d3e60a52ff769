"""Weight packer: reference state_dict -> the flat blob layout librdm_hip defines.

The library owns the layout (rdm_*_manifest, include/rdm_hip.h); this module only executes it.  Keys are the
reference's own state_dict names, so a checkpoint loaded with
`torch.load(ckpt)["state_dict"]` (scripts/rdm_sample.py:163) packs directly after prefix stripping.
"""
from typing import Dict

import numpy as np
import torch

from . import _lib


def _bf16_bytes(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().to(torch.bfloat16).view(torch.int16).numpy().view(np.uint8).reshape(-1)


def _f32_bytes(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().to(torch.float32).numpy().view(np.uint8).reshape(-1)


def _geglu_perm(n8c: int) -> torch.Tensor:
    """Row order that puts each 32-wide block of GEGLU `x` columns next to its 32 gate columns, so one MFMA
    wave tile holds both halves (igemm.hip GEGLU epilogue)."""
    half = n8c // 2
    q = torch.arange(half // 32)
    x_rows = (q[:, None] * 32 + torch.arange(32)[None]).reshape(-1, 32)
    g_rows = x_rows + half
    return torch.cat([x_rows, g_rows], dim=1).reshape(-1)


def pack(kind: str, cfg, sd: Dict[str, torch.Tensor]) -> np.ndarray:
    entries, blob_bytes = _lib.manifest(kind, cfg)
    blob = np.zeros(blob_bytes, dtype=np.uint8)
    for off, nbytes, kd, srcs in entries:
        ts = []
        for s in srcs:
            if s not in sd:
                raise KeyError(f"state_dict is missing '{s}' (needed by the {kind} manifest)")
            ts.append(sd[s].detach().to("cpu", torch.float32))
        if kd == "f32":
            data = _f32_bytes(torch.cat([t.reshape(-1) for t in ts]))
        elif kd == "bf16":
            data = _bf16_bytes(torch.cat([t.reshape(t.shape[0], -1) for t in ts], dim=0))
        elif kd == "bf16_t":
            assert len(ts) == 1 and ts[0].ndim == 2
            data = _bf16_bytes(ts[0].t())
        elif kd == "f32_t":
            assert len(ts) == 1 and ts[0].ndim == 2
            data = _f32_bytes(ts[0].t())
        elif kd == "conv3":
            assert len(ts) == 1 and ts[0].ndim == 4 and ts[0].shape[2:] == (3, 3)
            data = _bf16_bytes(ts[0].permute(0, 2, 3, 1))            # [N][ky][kx][C]  (K = tap*C + c)
        elif kd == "geglu_w":
            data = _bf16_bytes(ts[0][_geglu_perm(ts[0].shape[0])])
        elif kd == "geglu_b":
            data = _f32_bytes(ts[0][_geglu_perm(ts[0].shape[0])])
        elif kd == "fuse_w":      # [W_out W_2 | W_out]: ff.net.2 then proj_out as ONE linear map of [ff | t2] (model.hip transformer())
            w2, wo = ts[0].double(), ts[1].reshape(ts[1].shape[0], -1).double()
            data = _bf16_bytes(torch.cat([wo @ w2, wo], dim=1).float())
        elif kd == "fuse_b":      # W_out b_2 + b_out
            b2, wo, bo = ts[0].double(), ts[1].reshape(ts[1].shape[0], -1).double(), ts[2].double()
            data = _f32_bytes((wo @ b2 + bo).float())
        else:
            raise ValueError(f"unknown manifest kind {kd}")
        if data.nbytes != nbytes:
            raise ValueError(f"manifest entry {srcs} ({kd}): expected {nbytes} bytes, tensors give {data.nbytes}")
        blob[off:off + nbytes] = data
    return blob


def strip_prefix(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def ema_unet_state_dict(ckpt_sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Sampling runs under ema_scope (rdm/models/diffusion/ddpm.py:836, 977): the weights that matter are the
    LitEma copies, stored as `model_ema.<param name with dots removed>`.  Returns UNet weights
    keyed like `model.diffusion_model.*` with the EMA values substituted where present."""
    live = strip_prefix(ckpt_sd, "model.diffusion_model.")
    out = {}
    for k, v in live.items():
        ema_key = "model_ema." + ("diffusion_model." + k).replace(".", "")
        out[k] = ckpt_sd.get(ema_key, v)
    return out
