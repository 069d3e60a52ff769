#!/usr/bin/env python3
"""Is the conv kernel power-bound?  Runs one long-K conv shape in a loop on random and on zero-filled operands and samples
rocm-smi power / sclk meanwhile (GPU box only).  usage: power_probe.py"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RDM_OP_FRAG_CACHE", "1")
import torch
import rdm_amd
from rdm_amd import _lib
ctx = _lib.Context(0); d = ctx.device
B, H, C, N = 128, 32, 768, 384
samples = []
def sampler(stop):
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = [l for l in out.splitlines() if "Power" in l and "W" in l]
            ck = [l for l in out.splitlines() if "sclk" in l]
            samples.append((pw[:1], ck[:1]))
        except Exception as e:
            samples.append((str(e), None))
        time.sleep(0.2)
for fill in ("random", "zeros"):
    x = (torch.randn(B, H, H, C, device=d) if fill == "random" else torch.zeros(B, H, H, C, device=d)).bfloat16()
    w = ((torch.randn(N, 3, 3, C, device=d) * (9 * C) ** -0.5) if fill == "random" else torch.zeros(N, 3, 3, C, device=d)).bfloat16()
    b = torch.zeros(N, device=d)
    for _ in range(5): ctx.op_conv3x3(x, w, b)
    torch.cuda.synchronize()
    samples.clear(); stop = threading.Event(); th = threading.Thread(target=sampler, args=(stop,)); th.start()
    t0 = time.perf_counter(); n = 3000
    for _ in range(n): ctx.op_conv3x3(x, w, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    stop.set(); th.join()
    print(f"{fill}: {dt*1e3:.3f} ms  {2.0*B*H*H*N*9*C/dt/1e12:.0f} TF; samples: {samples[len(samples)//2:len(samples)//2+2]}", flush=True)
