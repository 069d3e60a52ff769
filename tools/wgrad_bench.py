"""Weight-gradient kernels at the shipped UNet's training shapes (batch 64): time per call and MFMA rate of op_conv3x3_wgrad /
op_linear_wgrad.  RDM_NO_WGRAD_TN=1 in the environment times the round-3 path (transposes + K-major GEMM + plane sums) instead.
    python tools/wgrad_bench.py [--batch 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rdm_amd  # noqa: E402,F401
from rdm_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    ctx = _lib.Context(0)
    d = ctx.device
    B = a.batch
    convs = [(64, 192, 192), (64, 384, 192), (32, 192, 384), (32, 384, 384), (32, 768, 384), (16, 576, 576), (16, 1152, 576), (8, 960, 960), (8, 1920, 960)]
    lins = [(32 * 32, 384, 1152), (32 * 32, 384, 3072), (32 * 32, 1536, 384), (16 * 16, 576, 1728), (16 * 16, 576, 4608), (16 * 16, 2304, 576),
            (8 * 8, 960, 2880), (8 * 8, 960, 7680), (8 * 8, 3840, 960)]
    tot = 0.0
    for H, C, N in convs:
        x = torch.randn(B, H, H, C, device=d).to(torch.bfloat16); dy = torch.randn(B, H, H, N, device=d).to(torch.bfloat16)
        for _ in range(2): ctx.op_conv3x3_wgrad(x, dy)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps): ctx.op_conv3x3_wgrad(x, dy)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        fl = 2.0 * B * H * H * N * 9 * C
        tot += ms
        print(f"conv wgrad {H:3d}x{H:<3d} {C:5d}->{N:<5d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
    for T, K, N in lins:
        M = B * T
        x = torch.randn(M, K, device=d).to(torch.bfloat16); dy = torch.randn(M, N, device=d).to(torch.bfloat16)
        for _ in range(2): ctx.op_linear_wgrad(dy, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps): ctx.op_linear_wgrad(dy, x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        fl = 2.0 * M * N * K
        tot += ms
        print(f"linear wgrad M={M:6d} {K:5d}->{N:<5d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
    print(f"sum of the listed calls: {tot:.2f} ms")


if __name__ == "__main__":
    main()
