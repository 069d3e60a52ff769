#!/usr/bin/env python3
"""Self-attention (d_head = 32) kernel timing at the UNet's three token counts, fused q | k | v operand (token-major V) as the sampler
runs it.  usage: attn_bench.py   env: RDM_FLASH_VAR (inner-loop variant bits, attention.hip), RDM_FLASH_OLD"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib
ctx = _lib.Context(0); d = ctx.device
for (B, n, heads) in ((128, 1024, 12), (128, 256, 18), (128, 64, 30)):
    C = heads * 32
    qkv = torch.randn(B, n, 3 * C, device=d).bfloat16()
    for _ in range(3): out = ctx.op_self_attention_qkv(qkv, heads)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = ctx.op_self_attention_qkv(qkv, heads)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    fl = 4.0 * n * n * 32 * heads * B
    print(f"attn B={B} n={n} heads={heads}: {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TF  checksum {float(out.float().abs().mean()):.6f}", flush=True)
