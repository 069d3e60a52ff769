#!/usr/bin/env python3
"""Whole-UNet optimisation step at the SHIPPED topology on the native ops (GPU box only): loss, backward to all 400.9 M
parameters, AdamW.  Synthetic weights / batch.  usage: train_step_bench.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib, training_unet as TU
from rdm_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _lib.Context(0); d = ctx.device
cfg = _lib.make_unet_cfg()                  # the shipped imagenet topology
spec = TU.TrainSpec(cfg)
sd = synthetic.unet_state_dict(cfg, seed=3)
P = TU.params_from_state_dict(sd, d)
nparam = sum(v.numel() for v in P.values())
state = TU.TrainState(P)                    # fp32 masters + AdamW moments + bf16 working copies (no per-forward casts)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 64, 64, 3, generator=g).to(d, torch.bfloat16); noise = torch.randn(B, 64, 64, 3, generator=g).to(d, torch.bfloat16)
cx = (torch.randn(B, 4, 512, generator=g) * 0.5).to(d, torch.bfloat16); t = torch.randint(0, 1000, (B,), generator=g).to(d)
for s in range(1, steps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = TU.unet_training_step(ctx, P, state, spec, x, t, cx, noise, lr=1e-4)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"step {s}: loss {loss:.5f}  {dt * 1e3:.0f} ms  ({nparam / 1e6:.1f} M parameters, batch {B}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB of torch allocations)", flush=True)
