#!/usr/bin/env python3
"""UNet-forward level timing (shipped config, B'=128 = CFG-doubled batch 64), split by kernel class via the library's
HIP-event hooks.  For A/B runs of two builds on the SAME box:  RDM_HIP_LIB=/path/to/variant.so python tools/unet_fwd_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd
from rdm_amd import _lib, packing, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
ffhq = "--ffhq" in sys.argv        # models/rdm/ffhq: model_channels 224, channel_mult 1-2-3-4 (runs zero-padded to 64-channel slices)
ctx = _lib.Context(0); d = ctx.device
cfg = _lib.make_unet_cfg(model_channels=224, channel_mult=(1, 2, 3, 4)) if ffhq else _lib.make_unet_cfg()
ctx.load_unet(cfg, packing.pack("unet", cfg, synthetic.unet_state_dict(cfg)))
g = torch.Generator(device=d).manual_seed(0)
x = torch.randn(128, 3, 64, 64, device=d, generator=g); t = torch.full((128,), 500, device=d, dtype=torch.long)
c = torch.randn(128, 4, 512, device=d, generator=g) * 0.45
for _ in range(2): ctx.unet_forward(x, t, c)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): ctx.unet_forward(x, t, c)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
ctx.prof_reset(); ctx.prof_enable(range(6))
for _ in range(n): ctx.unet_forward(x, t, c)
torch.cuda.synchronize()
ctx.prof_enable(())
parts = []
for name, kind in (("conv", 0), ("linear", 1), ("flash", 3), ("gn", 4), ("ln", 5)):
    k, ms, w = ctx.prof_collect(kind)
    parts.append(f"{name} {ms/n:.2f} ms" + (f" ({w/ms/1e9:.0f} TF)" if kind in (0, 1, 3) and ms > 0 else ""))
print(f"{os.environ.get('RDM_HIP_LIB', 'default')}: forward {dt*1e3:.2f} ms | " + " | ".join(parts), flush=True)
