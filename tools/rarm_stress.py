#!/usr/bin/env python3
"""Repeatability of the RARM decode path: N repeated 24-token decodes of a 64-sequence batch at the shipped size, compared bit for bit with the
first one (tests/test_gpu_rarm.py::test_rarm_decode_repeats_bitwise, longer).  RDM_RARM_XSPLIT=1 selects the opt-in four-block
cross-attention.  Optional second argument: megabytes of a scratch tensor rewritten between repeats (perturbs L2 / memory-side cache state)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rdm_amd
from rdm_amd import _lib, packing
from oracle import rarm as orarm, unet as ounet
import test_gpu_rarm as T
torch.set_grad_enabled(False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = _lib.Context(0)
spec = orarm.shipped_rarm_spec()
T._load(ctx, spec, 77)
gen = torch.Generator().manual_seed(5)
tokens = torch.randint(0, spec.vocab_out, (64, 24), generator=gen)
context = torch.randn((64, 8, spec.context_dim), generator=gen) * 0.45
first = ctx.rarm_forward(tokens, context).cpu()
scratch = torch.empty(mb << 18, device=ctx.device) if mb else None
heavy = len(sys.argv) > 3          # third argument: also run a UNet forward at batch 16 between repeats (clock / power / cache state of a loaded GPU)
if heavy:
    from _util import spec_to_unet_cfg
    us = ounet.shipped_spec()
    ucfg = spec_to_unet_cfg(us)
    ctx.load_unet(ucfg, packing.pack("unet", ucfg, ounet.synth_state_dict(ounet.param_shapes(us), seed=1234)))
    ux = torch.randn(16, 3, 64, 64); ut = torch.full((16,), 500); uc = torch.randn(16, 4, 512) * 0.45
bad = []
for rep in range(n):
    if scratch is not None:
        scratch.normal_()
    if heavy:
        ctx.unet_forward(ux, ut, uc)
    again = ctx.rarm_forward(tokens, context).cpu()
    if not torch.equal(again, first):
        d = (again - first).abs()
        rows = sorted(set(int(i) for i in torch.nonzero(d.amax(dim=(1, 2)) > 0).flatten()))
        pos = sorted(set(int(i) for i in torch.nonzero(d.amax(dim=(0, 2)) > 0).flatten()))
        bad.append((rep, float(d.max()), rows[:8], pos[:8]))
print(f"split={'on' if os.environ.get('RDM_RARM_XSPLIT') else 'off'} scratch={mb} MB: {len(bad)} of {n} repeats differ from the first", bad[:6])
