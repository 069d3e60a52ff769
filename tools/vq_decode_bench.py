#!/usr/bin/env python3
"""VQ-f4 decode of 64 latents at the shipped size (the decode leg of the headline step), with and without the wide-image strips of
conv_halo4 (RDM_NO_HALO4_STRIP=1 in a child process for the A/B).  GPU box only."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import rdm_amd
    from rdm_amd import _lib, packing, synthetic
    ctx = _lib.Context(0)
    cfg = _lib.make_vq_cfg()
    ctx.load_vq(cfg, packing.pack("vq", cfg, synthetic.vq_state_dict(cfg)))
    z = torch.randn(64, 3, 64, 64, device=ctx.device) * 0.6
    for _ in range(2): ctx.vq_decode(z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): img = ctx.vq_decode(z)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per 64-image decode (checksum {float(img.double().abs().mean()):.6f})")
else:
    for tag, env in (("generic implicit GEMM at 128 / 256 px", {"RDM_NO_HALO4_STRIP": "1"}), ("conv_halo4 strips", {})):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, **env}, capture_output=True, text=True)
        print(f"{tag}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]}")
