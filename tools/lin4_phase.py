#!/usr/bin/env python3
"""lin4 per-block phase clocks (RDM_LIN4_PROF=1 prints main-loop vs read-out cycles per block) for the UNet's big-M GEMM shapes."""
import os, sys
os.environ["RDM_LIN4_PROF"] = "1"
os.environ["RDM_OP_FRAG_CACHE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib
from rdm_amd.packing import _geglu_perm
ctx = _lib.Context(0); d = ctx.device
for name, M, N, K, act, res in (("geglu 32x32", 131072, 3072, 384, 1, 0), ("geglu 16x16", 32768, 4608, 576, 1, 0), ("qkv 32x32", 131072, 1152, 384, 0, 0),
                                ("proj_in 32x32", 131072, 384, 384, 0, 0), ("to_out 32x32", 131072, 384, 384, 0, 1), ("ff2*proj_out 32x32", 131072, 384, 1920, 0, 1)):
    a = torch.randn(M, K, device=d).bfloat16(); w = torch.randn(N, K, device=d).bfloat16() * K ** -0.5
    b = torch.randn(N, device=d)
    r = torch.randn(M, N, device=d).bfloat16() if res else None
    print(name, flush=True)
    for _ in range(2):
        ctx.op_linear(a, w, b, residual=r, act=act)
    torch.cuda.synchronize()
