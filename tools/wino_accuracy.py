#!/usr/bin/env python3
"""Verdict round 5, item 1(a): would Winograd F(2x2, 3x3) on bf16 MFMA keep the path inside its parity contract?  ZERO GPU minutes: the
library's CPU restatement (oracle/unet_emul.py) with the ResBlocks' stride-1 3x3 convs replaced by a Winograd conv rounded the way a fused
kernel would round (U = bf16(G g G^T), V = bf16(B^T d B) from fp32 transform math, fp32 products and sums, fp32 output transform, one
bf16 rounding of the result), at the shipped 400.9 M-parameter UNet, against the REFERENCE goldens:

    forward     tests/golden/unet_shipped.npz            eps of one forward                  (gate <= 2.0e-2; contract 2.5e-2; direct 1.2e-2)
    trajectory  tests/golden/full_ddim_k4.npz            50-step DDIM, CFG 2.0, k = 4        (gate <= 8e-3;   contract 1e-2;   direct 3.8e-3)

per level set: none (the direct conv: the emulator as the tests use it) / 64 + 32 / all four levels.  Also the op-level error of one conv.

    python tools/wino_accuracy.py forward | trajectory [levels ...]      e.g.  trajectory none 64,32 64,32,16,8
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import diffusion as odiff, unet as ounet
from oracle.unet_emul import unet_forward_emulated, wino_conv3x3, _R
torch.set_grad_enabled(False)
torch.set_num_threads(min(len(os.sched_getaffinity(0)), 32))
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
golden = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n))
what = sys.argv[1] if len(sys.argv) > 1 else "forward"
sets = [a for a in sys.argv[2:]] or ["none", "64,32", "64,32,16,8"]
parse = lambda s: None if s == "none" else set(int(v) for v in s.replace("+2", "").split(","))
spec = ounet.shipped_spec()

if what == "op":
    g = torch.Generator().manual_seed(3)
    bf = lambda t: t.to(torch.bfloat16).float()
    for (C, N, H) in ((192, 192, 64), (384, 384, 32), (576, 576, 16), (960, 960, 8)):
        x = bf(F.silu(torch.randn(2, C, H, H, generator=g)))                # what a ResBlock conv reads: SiLU of a normalised tensor
        w = bf(torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5)
        ref = F.conv2d(x.double(), w.double(), padding=1).float()
        direct = bf(F.conv2d(x, w, padding=1))
        wino = bf(wino_conv3x3(x, w, _R(True)))
        wino2 = bf(wino_conv3x3(x, w, _R(True), True))
        print(f"op {C}->{N} @ {H}x{H}: rel L2 vs exact:  direct + one rounding {rel(direct, ref):.3e}   Winograd {rel(wino, ref):.3e}   Winograd, two-stage bf16 input transform {rel(wino2, ref):.3e}", flush=True)
    sys.exit(0)

if what == "forward":
    g = golden("unet_shipped.npz")
    sd = ounet.synth_state_dict(ounet.param_shapes(spec), seed=int(g["seed"]))
    x, t, c, eps = (torch.from_numpy(g[k]) for k in ("x", "t", "ctx", "eps"))
    for s in sets:
        for two in ((False, True) if s != "none" else (False,)):
            t0 = time.time()
            e = unet_forward_emulated(sd, spec, x, t, c, wino=parse(s), wino_two_stage=two)
            print(f"forward, Winograd levels {s:12s}{' (two-stage bf16 input transform)' if two else '':36s}: eps rel L2 vs the reference golden {rel(e, eps):.3e}   ({time.time() - t0:.0f} s)", flush=True)
    sys.exit(0)

if what == "trajectory":
    g = golden("full_ddim_k4.npz")
    sd = ounet.synth_state_dict(ounet.param_shapes(spec), seed=1234)
    x_T, cond = torch.from_numpy(g["x_T"]), torch.from_numpy(g["cond"])
    sched = odiff.Schedule()
    for s in sets:
        lv = parse(s)
        t0 = time.time()
        apply = lambda x_, t_, c_: unet_forward_emulated(sd, spec, x_, t_, c_, wino=lv)
        z, inter = odiff.ddim_sample(apply, sched, 50, x_T, cond, scale=float(g["scale"]), uncond=torch.zeros_like(cond), log_every_t=1)
        xs = inter["x_inter"]                       # [x_T, after iteration 0, 1, ...]
        errs = {int(i): rel(xs[int(i) + 1], torch.from_numpy(g[f"x_{int(i)}"])) for i in g["steps"]}
        print(f"trajectory (50-step DDIM, CFG {float(g['scale']):.1f}, k = 4), Winograd levels {s:12s}: state after iteration "
              + ", ".join(f"{i}: {e:.3e}" for i, e in errs.items()) + f"; final {rel(z, torch.from_numpy(g['z'])):.3e}   ({time.time() - t0:.0f} s)", flush=True)
