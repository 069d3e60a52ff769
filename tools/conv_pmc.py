#!/usr/bin/env python3
"""Run one 3x3 conv shape a few times (for rocprofv3 --pmc passes).  usage: conv_pmc.py B H C N [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib
B, H, C, N = [int(v) for v in sys.argv[1:5]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
ctx = _lib.Context(0)
d = ctx.device
x = torch.randn(B, H, H, C, device=d).bfloat16(); w = torch.randn(N, 3, 3, C, device=d).bfloat16(); b = torch.zeros(N, device=d)
for _ in range(reps):
    ctx.op_conv3x3(x, w, b)
torch.cuda.synchronize()
