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


def _parse_kind(kd: str):
    """'kind|R=a>b,c>d|C=e>f' -> (kind, row segments, channel segments): the library's channel-padding recipe (model.hip build_unet):
    a dimension of sum(logical) entries becomes sum(padded) entries, every segment followed by its zero tail."""
    parts = kd.split("|")
    rows = cols = None
    for sp in parts[1:]:
        segs = [tuple(int(v) for v in g.split(">")) for g in sp[2:].split(",")]
        if sp.startswith("R="):
            rows = segs
        elif sp.startswith("C="):
            cols = segs
        else:
            raise ValueError(f"unknown manifest recipe '{sp}'")
    return parts[0], rows, cols


def _pad_dim(t: torch.Tensor, dim: int, segs) -> torch.Tensor:
    """Pad dimension `dim` segment by segment (logical n -> padded p, zeros behind each segment)."""
    if segs is None:
        return t
    if sum(n for n, _ in segs) != t.shape[dim]:
        raise ValueError(f"padding recipe {segs} does not cover a dimension of {t.shape[dim]}")
    out, pos = [], 0
    for n, p_ in segs:
        piece = t.narrow(dim, pos, n); pos += n
        if p_ > n:
            shape = list(t.shape); shape[dim] = p_ - n
            piece = torch.cat([piece, torch.zeros(shape, dtype=t.dtype)], dim=dim)
        out.append(piece)
    return torch.cat(out, dim=dim)


def pack(kind: str, cfg, sd: Dict[str, torch.Tensor]) -> np.ndarray:
    entries, blob_bytes = _lib.manifest(kind, cfg)
    blob = np.zeros(blob_bytes, dtype=np.uint8)
    for off, nbytes, kd_full, srcs in entries:
        kd, rseg, cseg = _parse_kind(kd_full)
        ts = []
        for s in srcs:
            if s not in sd:
                raise KeyError(f"state_dict is missing '{s}' (needed by the {kind} manifest)")
            ts.append(sd[s].detach().to("cpu", torch.float32))
        if kd == "f32":
            if rseg is not None and ts[0].ndim >= 2:          # [rows, ...] (conv_in weights): pad the rows
                assert len(ts) == 1
                data = _f32_bytes(_pad_dim(ts[0].reshape(ts[0].shape[0], -1), 0, rseg))
            else:
                data = _f32_bytes(_pad_dim(torch.cat([t.reshape(-1) for t in ts]), 0, rseg))
        elif kd == "f32_cin":                                 # [Cout][Cin][3][3] fp32 (conv_out_kernel): the Cin axis is padded
            assert len(ts) == 1 and ts[0].ndim == 4
            data = _f32_bytes(_pad_dim(ts[0], 1, cseg))
        elif kd == "bf16":
            data = _bf16_bytes(_pad_dim(_pad_dim(torch.cat([t.reshape(t.shape[0], -1) for t in ts], dim=0), 0, rseg), 1, cseg))
        elif kd == "bf16_t":
            assert len(ts) == 1 and ts[0].ndim == 2
            data = _bf16_bytes(ts[0].t())
        elif kd == "f32_t":
            assert len(ts) == 1 and ts[0].ndim == 2
            data = _f32_bytes(ts[0].t())
        elif kd == "conv3":
            assert len(ts) == 1 and ts[0].ndim == 4 and ts[0].shape[2:] == (3, 3)
            data = _bf16_bytes(_pad_dim(_pad_dim(ts[0].permute(0, 2, 3, 1), 0, rseg), 3, cseg))     # [N][ky][kx][C]  (K = tap*C + c)
        elif kd == "geglu_w":
            data = _bf16_bytes(_pad_dim(ts[0][_geglu_perm(ts[0].shape[0])], 1, cseg))
        elif kd == "geglu_b":
            data = _f32_bytes(ts[0][_geglu_perm(ts[0].shape[0])])
        elif kd == "fuse_w":      # [W_out W_2 | W_out]: ff.net.2 then proj_out as ONE linear map of [ff | t2] (model.hip transformer())
            w2, wo = ts[0].double(), ts[1].reshape(ts[1].shape[0], -1).double()
            data = _bf16_bytes(_pad_dim(_pad_dim(torch.cat([wo @ w2, wo], dim=1).float(), 0, rseg), 1, cseg))
        elif kd == "fuse_b":      # W_out b_2 + b_out
            b2, wo, bo = ts[0].double(), ts[1].reshape(ts[1].shape[0], -1).double(), ts[2].double()
            data = _f32_bytes(_pad_dim((wo @ b2 + bo).float(), 0, rseg))
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
