#!/usr/bin/env python3
"""Headline benchmark: images/sec at 256x256, 50 DDIM steps, k=4 retrieval (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path over one batch of synthetic inputs that are already resident in
HBM:  exact kNN of B query embeddings over the CLIP-embedding DB -> gather neighbours -> conditioning
[q, nn_0..nn_{k-2}] -> 50-step DDIM with classifier-free guidance (scale 2.0, batch doubling) over the
shipped-config UNet -> VQ-f4 decode to [B,3,256,256] fp32 (+ RCCL all-gather of the images when N > 1).
Workload = BASELINE config #3 (B=64 per GPU, k=4, bf16 compute). Weights are seeded random tensors of the
shipped architecture (no checkpoints are reachable), DB and queries are synthetic.

For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU, RCCL); the batch is
sharded (64 images per GPU, weak scaling), the only collective is the all-gather of finished images.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel = the 3x3 convolution (input-stationary halo kernel + the generic implicit GEMM for
                strided / upsampled / decoder convs; 63 % of all FLOPs), MFMA-bound; achieved = algorithmic FLOPs
                (2*M*N*9*Cin per launch) / HIP-event time of those launches, measured live in the timed region on the
                library's stream; peak = 2.5 PFLOP/s dense bf16.  `traffic` = HBM bytes per launch of the halo kernel from the
                committed PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs; profiles/r01_pmc_v4.json).
  cpu_baseline  the fp32 PyTorch oracle (kind "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=64, help="images per GPU per step (BASELINE config #3: 64)")
    p.add_argument("--ddim-steps", type=int, default=50)
    p.add_argument("--k", type=int, default=4)
    p.add_argument("--scale", type=float, default=2.0)
    p.add_argument("--db-rows", type=int, default=20_927_907, help="OpenImages DB rows (SURVEY §8 a-13)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the untimed extras (CLIP text encode, guidance-scale-1.0 step)")
    return p.parse_args()


def cpu_baseline(sd_unet, spec, sd_vq, vspec, ddim_steps, scale):
    """Oracle (fp32 PyTorch restatement of the reference arithmetic) on the host cores, bounded sample:
    BASELINE config #1 shapes (B=1, CFG => UNet batch 2): 1 warm-up + 2 timed UNet forwards and 1 timed VQ
    decode, extrapolated to ddim_steps forwards + 1 decode per image."""
    import torch
    from oracle import unet as ounet, vqdecoder as ovq
    cores = min(len(os.sched_getaffinity(0)), 32)      # threads actually used (more oversubscribes the small convs)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    nb = 2 if scale > 1.0 else 1
    x = torch.randn(nb, 3, 64, 64, generator=g)
    t = torch.full((nb,), 981, dtype=torch.long)
    c = torch.randn(nb, 4, 512, generator=g) * 0.45
    with torch.no_grad():
        ounet.unet_forward(sd_unet, spec, x, t, c)
        t0 = time.perf_counter()
        for _ in range(2):
            ounet.unet_forward(sd_unet, spec, x, t, c)
        t_unet = (time.perf_counter() - t0) / 2
        t0 = time.perf_counter()
        ovq.vq_decode(sd_vq, vspec, x[:1])
        t_dec = time.perf_counter() - t0
    per_img = ddim_steps * t_unet + t_dec
    return {"value": 1.0 / per_img, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"fp32 oracle, B=1 (UNet batch {nb} with CFG): 2 timed UNet forwards ({t_unet:.3f} s each) + 1 VQ decode "
                      f"({t_dec:.3f} s), extrapolated to {ddim_steps} forwards + 1 decode per image; retrieval excluded"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world == 1:
        # convenience: self-launch one rank per GPU (nothing has touched the GPU yet in this process)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist
    import rdm_amd  # noqa: F401
    from rdm_amd import _lib, packing
    from oracle import diffusion as odiff, unet as ounet, vqdecoder as ovq
    from _util import spec_to_unet_cfg, spec_to_vq_cfg

    torch.set_grad_enabled(False)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    ctx = _lib.Context(local)

    # ---- model: shipped architecture, seeded random weights
    spec, vspec = ounet.shipped_spec(), ovq.shipped_vq_spec()
    sd_unet = ounet.synth_state_dict(ounet.param_shapes(spec), seed=1234)
    sd_vq = ounet.synth_state_dict(ovq.vq_param_shapes(vspec), seed=4321)
    ucfg, vcfg = spec_to_unet_cfg(spec), spec_to_vq_cfg(vspec)
    ctx.load_unet(ucfg, packing.pack("unet", ucfg, sd_unet))
    ctx.load_vq(vcfg, packing.pack("vq", vcfg, sd_vq))
    sched = odiff.Schedule()

    # ---- retrieval DB: synthetic CLIP-like rows, generated on device, replicated per GPU
    N, D, B, k = a.db_rows, 512, a.batch, a.k
    gen = torch.Generator(device=dev).manual_seed(7)
    db = torch.empty((N, D), device=dev, dtype=torch.float16)
    for r0 in range(0, N, 1 << 20):
        r1 = min(N, r0 + (1 << 20))
        db[r0:r1] = (torch.randn((r1 - r0, D), device=dev, generator=gen) * 0.45).half()
    ctx.db_load(db)
    del db
    torch.cuda.empty_cache()

    total_steps = a.warmup + a.steps
    qgen = torch.Generator(device=dev).manual_seed(11 + rank)
    queries = torch.randn((total_steps, B, D), device=dev, generator=qgen) * 0.45
    x_Ts = torch.randn((total_steps, B, 3, 64, 64), device=dev, generator=qgen)
    uncond = torch.zeros((B, k, D), device=dev)
    gathered = torch.empty((world * B, 3, 256, 256), device=dev) if world > 1 else None

    ev_ret = []                                                         # (start, end) events around the retrieval of each timed step

    def step(i, scale=None, record=False):
        scale = a.scale if scale is None else scale
        q = queries[i]
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        idx, _ = ctx.knn(q, k)
        nbrs = ctx.db_gather(idx, D)                                   # [B,k,512] raw neighbour embeddings
        if record:
            e1.record(torch.cuda.current_stream()); ev_ret.append((e0, e1))
        cond = torch.cat([q[:, None], nbrs[:, :k - 1]], dim=1).contiguous()   # ddpm.py:775 (query first)
        z, _, _ = ctx.ddim_sample(a.ddim_steps, x_Ts[i], cond, uncond if scale > 1 else None, sched.alphas_cumprod,
                                  eta=0.0, scale=scale)
        img = ctx.vq_decode(z)
        if world > 1:
            dist.all_gather_into_tensor(gathered, img)
        return img

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(a.warmup, total_steps):
        img = step(i, record=True)
    fence()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert bool(torch.isfinite(img).all()), "non-finite images"

    # HBM traffic of the dominant kernel: not measurable from inside this process -- taken from the committed PMC passes
    # (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this script, gfx950 corrections applied; profiles/README.md)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_v4.json")) as f:
            traffic = json.load(f)["hbm_bytes_per_launch"]
    except Exception:
        pass
    # ---- untimed extras (SURVEY 8d asks for them next to the headline): retrieval latency per batch, one step without
    # classifier-free guidance (no batch doubling), CLIP text encoding of one batch of captions
    extras = {}
    if rank == 0:
        try:
            extras["retrieval_ms_per_batch"] = sum(e0.elapsed_time(e1) for e0, e1 in ev_ret) / max(len(ev_ret), 1)
        except Exception:      # events of another stream: library runs on the legacy stream by default
            pass
    if world == 1 and not a.no_extras:      # single-process only: step() contains the all-gather collective when world > 1
        torch.cuda.synchronize(); t1 = time.perf_counter()
        step(0, scale=1.0)
        torch.cuda.synchronize()
        extras["images_per_s_guidance_scale_1"] = B / (time.perf_counter() - t1)
        from oracle import clip as oclip
        from _util import spec_to_clip_cfg
        cspec = oclip.vitb32_spec()
        ccfg = spec_to_clip_cfg(cspec)
        ctx.load_clip(ccfg, packing.pack("clip", ccfg, ounet.synth_state_dict(oclip.clip_param_shapes(cspec), seed=99)))
        toks = torch.randint(1, cspec.vocab_size - 2, (B, cspec.context_length), device=dev)
        toks[:, 0] = cspec.vocab_size - 2; toks[:, 20] = cspec.vocab_size - 1          # <start> ... <end> (argmax position)
        ctx.clip_encode_text(toks); torch.cuda.synchronize(); t1 = time.perf_counter()
        ctx.clip_encode_text(toks); torch.cuda.synchronize()
        extras["clip_text_encode_ms_per_batch"] = (time.perf_counter() - t1) * 1e3
    n_conv, ms_conv, fl_conv = ctx.prof_collect(0)
    n_lin, ms_lin, fl_lin = ctx.prof_collect(1)
    if rank == 0:
        images = world * B * a.steps
        achieved = fl_conv / (ms_conv * 1e-3) / 1e12 if ms_conv > 0 else 0.0
        out = {
            "metric": "images/sec at 256x256, 50 DDIM steps, k=4 OpenImages retrieval",
            "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE config #3: RDM-OpenImages sampling, exact kNN (k=%d) over a synthetic %d x 512 fp16 "
                                   "CLIP DB -> %d-step DDIM (eta 0, CFG scale %.1f) over the shipped-config UNet (400.9M params, "
                                   "random weights) -> VQ-f4 decode to 256x256" % (k, N, a.ddim_steps, a.scale),
                       "batch_per_gpu": B, "global_batch": world * B, "ddim_steps": a.ddim_steps, "k": k,
                       "guidance_scale": a.scale, "db_rows": N, "parallelism": f"dp{world} (batch-sharded, DB replicated, "
                                                                                "all-gather of images only)"},
            "roofline": {"kernel": "conv3x3_halo_kernel<192> + igemm_kernel<..,conv> (3x3 conv, bf16 MFMA, fp32 accumulate)", "bound": "mfma",
                         "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0, "traffic": traffic,
                         "traffic_note": "bytes per launch of conv3x3_halo_kernel<192> from the committed rocprofv3 PMC passes (profiles/r01_pmc_v4.json), not collected in this run",
                         "launches": n_conv, "avg_launch_ms": ms_conv / max(n_conv, 1),
                         "algorithmic_tflop_per_launch": fl_conv / max(n_conv, 1) / 1e12,
                         "linear_gemm": {"achieved": (fl_lin / (ms_lin * 1e-3) / 1e12) if ms_lin > 0 else 0.0, "launches": n_lin,
                                         "time_ms": ms_lin},
                         "conv_time_frac_of_step": ms_conv * 1e-3 / dt},
        }
        if extras:
            out["extras"] = extras
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd_unet, spec, sd_vq, vspec, a.ddim_steps, a.scale)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
