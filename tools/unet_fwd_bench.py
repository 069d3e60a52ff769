#!/usr/bin/env python3
"""UNet-forward level timing (shipped config, B'=128 = CFG-doubled batch 64), split by kernel class via the library's
HIP-event hooks.  For A/B runs of two builds on the SAME box:  RDM_HIP_LIB=/path/to/variant.so python tools/unet_fwd_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rdm_amd
from rdm_amd import _lib, packing
from oracle import unet as ounet
from _util import spec_to_unet_cfg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = _lib.Context(0); d = ctx.device
spec = ounet.shipped_spec()
sd = ounet.synth_state_dict(ounet.param_shapes(spec), seed=1234)
cfg = spec_to_unet_cfg(spec)
ctx.load_unet(cfg, packing.pack("unet", cfg, sd))
g = torch.Generator(device=d).manual_seed(0)
x = torch.randn(128, 3, 64, 64, device=d, generator=g); t = torch.full((128,), 500, device=d, dtype=torch.long)
c = torch.randn(128, 4, 512, device=d, generator=g) * 0.45
for _ in range(2): ctx.unet_forward(x, t, c)
torch.cuda.synchronize()
ctx.prof_reset(); ctx.prof_enable(True)
t0 = time.perf_counter()
for _ in range(n): ctx.unet_forward(x, t, c)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
ctx.prof_enable(False)
nc, msc, flc = ctx.prof_collect(0); nl, msl, fll = ctx.prof_collect(1)
print(f"{os.environ.get('RDM_HIP_LIB', 'default')}: forward {dt*1e3:.2f} ms | conv {msc/n:.2f} ms ({flc/msc/1e9:.0f} TF) | linear {msl/n:.2f} ms ({fll/msl/1e9:.0f} TF) | other {dt*1e3-(msc+msl)/n:.2f} ms")
