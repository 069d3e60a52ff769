#!/usr/bin/env python3
"""Reference gradients of the SHIPPED-topology UNet (400.9 M parameters) for the training-step parity test (VERDICT r03
item 3c): torch autograd on the CPU through the oracle's fp32 restatement of UNetModel (oracle/unet.py, asserted equal to the reference
class by tools/gen_golden.py) of ldm p_losses' loss_simple = mean((eps_theta(x_t, t, c) - noise)^2) at B = 2, 64 x 64, k = 4.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_grads.py        ->  tests/golden/unet_shipped_grads.npz

Stored per parameter of input_blocks.{1,4,7,10}, middle_block, output_blocks.{0,5,11}, out, time_embed: the gradient's L2 norm and
NSAMP elements at positions drawn from default_rng(crc32(name)) (fp16 relative to the tensor's max |g|: values / scale), plus the loss.
Inputs and weights are re-derived from the seeds by the test (weights with >= 2 dims rounded to bf16 on both sides)."""
import os
import sys
import time
import zlib

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import unet as ounet

NSAMP = 1024
SEED_W, SEED_X = 1234, 77
BLOCKS = ("input_blocks.1.", "input_blocks.4.", "input_blocks.7.", "input_blocks.10.", "middle_block.", "output_blocks.0.",
          "output_blocks.5.", "output_blocks.11.", "out.", "time_embed.")


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def inputs():
    rng = np.random.default_rng(SEED_X)
    x = bf16_round(torch.from_numpy(rng.standard_normal((2, 3, 64, 64)).astype(np.float32)))
    cx = bf16_round(torch.from_numpy((rng.standard_normal((2, 4, 512)) * 0.45).astype(np.float32)))
    noise = bf16_round(torch.from_numpy(rng.standard_normal((2, 3, 64, 64)).astype(np.float32)))
    return x, cx, noise, torch.tensor([481, 37])


def sample_positions(name, numel):
    return np.random.default_rng(zlib.crc32(name.encode())).integers(0, numel, size=min(NSAMP, numel))


if __name__ == "__main__":
    torch.set_num_threads(8)
    spec = ounet.shipped_spec()
    sd = {k: (bf16_round(v) if v.dim() >= 2 else v) for k, v in ounet.synth_state_dict(ounet.param_shapes(spec), seed=SEED_W).items()}
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x, cx, noise, t = inputs()
    t0 = time.time()
    loss = ((ounet.unet_forward(ref, spec, x, t, cx) - noise) ** 2).mean()
    loss.backward()
    print(f"loss {loss.item():.6f}; forward + backward {time.time() - t0:.0f} s")
    out = {"loss": np.float64(loss.item()), "nsamp": np.int64(NSAMP), "seed_w": np.int64(SEED_W), "seed_x": np.int64(SEED_X)}
    names = [k for k in sd if k.startswith(BLOCKS)]
    for k in names:
        g = ref[k].grad.detach().reshape(-1)
        pos = sample_positions(k, g.numel())
        scale = float(g.abs().max()) or 1.0
        out["n:" + k] = np.float32(g.double().norm().item())
        out["s:" + k] = np.float32(scale)
        out["v:" + k] = (g[torch.from_numpy(pos)] / scale).numpy().astype(np.float16)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "unet_shipped_grads.npz"), **out)
    print(f"{len(names)} tensors stored")
