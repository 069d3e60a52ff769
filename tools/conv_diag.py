#!/usr/bin/env python3
"""Error map of one conv shape against the fp32 CPU reference: which tiles / rows / column blocks are wrong. usage: conv_diag.py B H C N ups"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import rdm_amd
from rdm_amd import _lib
B, H, C, N, ups = [int(v) for v in sys.argv[1:6]]
ctx = _lib.Context(0); d = ctx.device
g = torch.Generator().manual_seed(1)
x = torch.randn(B, H, H, C, generator=g).bfloat16()
w = (torch.randn(N, 3, 3, C, generator=g) * (9 * C) ** -0.5).bfloat16()
b = torch.zeros(N)
xx = x.float().permute(0, 3, 1, 2)
if ups: xx = F.interpolate(xx, scale_factor=2, mode="nearest")
ref = F.conv2d(xx, w.float().permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
out = ctx.op_conv3x3(x.to(d), w.to(d), b.to(d), ups=ups).float().cpu()
Ho = ref.shape[1]
e = (out - ref).abs().reshape(-1, N)                       # [M, N]
M = e.shape[0]
print("M", M, "tiles", M // 256, "rel l2", float((out - ref).norm() / ref.norm()))
et = e.reshape(M // 256, 256, N)
bad_tiles = (et.amax(dim=(1, 2)) > 0.1).nonzero().flatten().tolist()
print("bad tiles:", len(bad_tiles), "of", M // 256, bad_tiles[:40])
if bad_tiles:
    t = bad_tiles[0]
    rows = (et[t].amax(dim=1) > 0.1).nonzero().flatten().tolist()
    cols = (et[t].amax(dim=0) > 0.1).nonzero().flatten().tolist()
    print("tile", t, "bad rows", len(rows), rows[:64])
    print("tile", t, "bad cols", len(cols), cols[:64])
    # per 32x32 fragment
    fr = et[t].reshape(8, 32, N // 32, 32).amax(dim=(1, 3))
    print("fragment max err (8 row-frags x N/32 col-frags):\n", (fr > 0.1).int())
