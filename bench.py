#!/usr/bin/env python3
"""Headline benchmark: images/sec at 256x256, 50 DDIM steps, k=4 retrieval (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config {2,3,4}]

A "step" is one pass of the whole hot path over one batch of synthetic inputs that are already resident in HBM.
  --config 3 (default, the configuration the metric is quoted on): exact kNN of B=64 query embeddings over the
      20 927 907 x 512 fp16 CLIP-embedding DB -> gather neighbours -> conditioning [q, nn_0..nn_{k-2}] (k=4) -> 50-step DDIM
      (eta 0) with classifier-free guidance 2.0 (UNet batch 2B) over the shipped-config UNet -> VQ-f4 decode to
      [B,3,256,256] fp32.
  --config 2: text-only (no retrieval): CLIP ViT-B/32 text tower on B=64 tokenised captions -> cond [B,1,512] (k=1) -> the same
      50-step DDIM + decode.
  --config 4: k=16 retrieval, 250 ancestral DDPM steps (ldm p_sample_loop(timesteps=250): no CFG on this path, as in the
      reference), B=64 per GPU (512 over 8 GPUs).
  --config 5: RARM (scripts/rarm_sample.py path): k=8 retrieval -> 256 autoregressive tokens (18-layer RetrievalPatchTransformer with a
      K/V cache, top-k 256 multinomial, guidance scale 1.0 = the script's default) -> VQGAN-f16 decode, B=2048 sequences per GPU (BASELINE.json
      does not fix this config's batch; one token step is ~110 dependent launches whatever the batch, so img/s per GPU grows 200 / 300 / 424 /
      585 / 692 / 878 / 994 for 64 / 128 / 256 / 512 / 1024 / 2048 / 4096 sequences (round 5) and flattens from 2048 on (+ 13 % for the next
      doubling, at 4.1 s per step): the K/V-cache attention is then HBM-bound and the first-stage decode, walked in 128-image ranges, costs the
      same per image at any batch).
Weights are seeded random tensors of the shipped architectures (no checkpoints are reachable), DB / queries / captions
are synthetic.

N > 1: the driver launches this file under torch.distributed.run (one rank per GPU, RCCL).  The global batch N*B is sharded
with the product's helpers (rdm_amd.parallel: shard_range / per_sample_noise / all_gather_images): every rank samples its
rows — inputs are a function of the GLOBAL row index — weights and DB are replicated, the only collective is the all-gather of
the finished images (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline      dominant kernel = the 3x3 convolution (63 % of all FLOPs), MFMA-bound: achieved = algorithmic FLOPs
                (2*M*N*9*Cin per launch) / HIP-event time of those launches, recorded live INSIDE the timed region on the
                library's stream; peak 2.5 PFLOP/s dense bf16.  `traffic` = HBM bytes per launch from the committed rocprofv3
                PMC passes of this command (profiles/).  Sub-objects from one extra UNTIMED step with events around the other
                kernel classes: linear GEMMs, flash attention, GroupNorm, LayerNorm, and `knn` (HBM-bound: bytes of the
                database pass / ms, against 8 TB/s).
  cpu_baseline  the fp32 PyTorch oracle (kind "port") on this box's host cores on a bounded sample of config #1.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None, help="timed steps (default 3; 1 for --config 4)")
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", type=int, default=3, choices=(2, 3, 4, 5), help="BASELINE.json config number")
    p.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 64; --config 5: 2048 sequences, where the decode step's img/s per GPU "
                                                            "levels off -- BASELINE.json does not fix config #5's batch)")
    p.add_argument("--ddim-steps", type=int, default=None, help="sampler steps (default 50; 250 for --config 4)")
    p.add_argument("--k", type=int, default=None, help="neighbours (default 4; 1 for --config 2; 16 for --config 4)")
    p.add_argument("--scale", type=float, default=2.0)
    p.add_argument("--db-rows", type=int, default=20_927_907, help="OpenImages DB rows (SURVEY §8 a-13)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--full-cpu-baseline", action="store_true",
                   help="SURVEY 8d's protocol instead of the bounded sample: 1 warm-up + 3 FULL config-#1 runs (50 DDIM steps + decode), median; ~5 min on 32 cores")
    p.add_argument("--no-extras", action="store_true", help="skip the untimed extras (per-class roofline step, guidance-scale-1.0 step)")
    p.add_argument("--gather", choices=("f32", "uint8"), default="f32",
                   help="N > 1: what the one collective moves -- the decoded fp32 images [B,3,256,256] (50 MB per rank at B = 64, default) or their "
                        "uint8 HWC form (rdm_to_uint8, the conversion of scripts/rdm_sample.py: 12.6 MB per rank; SURVEY 8e)")
    p.add_argument("--no-calibration", action="store_true", help="skip the box-calibration probes (~2 s before the warm-up) and the clock / power sampler")
    p.add_argument("--dump-images", default=None, help="rank 0 writes the last timed step's gathered images to this .npy (parity tests of the N > 1 path)")
    a = p.parse_args()
    a.k = a.k if a.k is not None else {2: 1, 3: 4, 4: 16, 5: 8}[a.config]
    a.batch = a.batch if a.batch is not None else (2048 if a.config == 5 else 64)
    a.ddim_steps_given = a.ddim_steps is not None
    a.ddim_steps = a.ddim_steps if a.ddim_steps is not None else (250 if a.config == 4 else 50)
    a.steps = a.steps if a.steps is not None else (1 if a.config == 4 else 3)
    return a


def cpu_baseline_full(scale):
    """SURVEY.md 8d as written: the oracle's whole config-#1 pipeline (B=1, k=4, 50 DDIM steps with CFG, VQ-f4 decode), one warm-up run +
    three timed runs, median wall-clock per image.  Minutes of CPU: behind --full-cpu-baseline, result recorded in BASELINE.md."""
    import numpy as np
    import torch
    from oracle import diffusion as odiff, unet as ounet, vqdecoder as ovq
    cores = min(len(os.sched_getaffinity(0)), 32)
    torch.set_num_threads(cores)
    spec, vspec = ounet.shipped_spec(), ovq.shipped_vq_spec()
    sd = ounet.synth_state_dict(ounet.param_shapes(spec), seed=1234)
    sdv = ounet.synth_state_dict(ovq.vq_param_shapes(vspec), seed=4321)
    sched = odiff.Schedule()
    sch = odiff.ddim_schedule(sched, 50, 0.0)
    runs = []
    with torch.no_grad():
        for r in range(4):
            g = torch.Generator().manual_seed(r)
            x = torch.randn(1, 3, 64, 64, generator=g)
            c = torch.randn(1, 4, 512, generator=g) * 0.45
            t0 = time.perf_counter()
            for index in range(49, -1, -1):
                t = torch.full((1,), int(sch[0][index]), dtype=torch.long)
                x, _ = odiff.p_sample_ddim(lambda x_, t_, c_: ounet.unet_forward(sd, spec, x_, t_, c_), x, c, t, index, sch, scale=scale,
                                           uc=torch.zeros_like(c))
            ovq.vq_decode(sdv, vspec, x)
            runs.append(time.perf_counter() - t0)
    per_img = float(np.median(runs[1:]))
    return {"value": 1.0 / per_img, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"fp32 oracle, BASELINE config #1 in full (B=1, k=4, CFG scale {scale}, 50 DDIM steps + VQ-f4 decode): 1 warm-up run + 3 timed runs "
                      f"({', '.join(f'{t:.1f}' for t in runs[1:])} s), median {per_img:.1f} s per image; retrieval excluded"}


def cpu_baseline(scale):
    """Oracle (fp32 PyTorch restatement of the reference arithmetic; test infrastructure, imported ONLY here) on the host
    cores, bounded sample of BASELINE config #1 (B=1, k=4, CFG => UNet batch 2): a real 4-step DDIM trajectory (1 warm-up
    step + 3 timed, median) + 1 VQ decode, extrapolated to 50 steps + 1 decode per image."""
    import numpy as np
    import torch
    from oracle import diffusion as odiff, unet as ounet, vqdecoder as ovq
    cores = min(len(os.sched_getaffinity(0)), 32)      # threads actually used (more oversubscribes the small convs)
    torch.set_num_threads(cores)
    spec, vspec = ounet.shipped_spec(), ovq.shipped_vq_spec()
    sd = ounet.synth_state_dict(ounet.param_shapes(spec), seed=1234)
    sdv = ounet.synth_state_dict(ovq.vq_param_shapes(vspec), seed=4321)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, 64, 64, generator=g)
    c = torch.randn(1, 4, 512, generator=g) * 0.45
    sched = odiff.Schedule()
    sch = odiff.ddim_schedule(sched, 50, 0.0)
    times = []
    with torch.no_grad():
        for i in range(4):
            index = 49 - i
            t = torch.full((1,), int(sch[0][index]), dtype=torch.long)
            t0 = time.perf_counter()
            x, _ = odiff.p_sample_ddim(lambda x_, t_, c_: ounet.unet_forward(sd, spec, x_, t_, c_), x, c, t, index, sch, scale=scale,
                                       uc=torch.zeros_like(c))
            times.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        ovq.vq_decode(sdv, vspec, x)
        t_dec = time.perf_counter() - t0
    t_step = float(np.median(times[1:]))
    per_img = 50 * t_step + t_dec
    return {"value": 1.0 / per_img, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"fp32 oracle, BASELINE config #1 shapes (B=1, k=4, CFG scale {scale} => UNet batch 2): 4 consecutive DDIM steps of "
                      f"the 50-step schedule (first = warm-up, median of the other 3 = {t_step:.3f} s/step) + 1 VQ-f4 decode ({t_dec:.3f} s), "
                      f"extrapolated to 50 steps + 1 decode per image; retrieval excluded"}


class BoxSampler:
    """Shader clock and package power of this rank's GPU, sampled from a thread during the timed region (calibration object of the
    JSON line).  amdgpu's hwmon files when they are readable (microwatts / Hz: no child process), else `rocm-smi` (one child per sample,
    as tools/power_insitu.py does from outside).  Sampling failures leave the fields null: the benchmark never depends on them."""

    def __init__(self, period=0.2):
        import glob
        import threading
        self.period, self.samples, self._stop, self._th = period, [], threading.Event(), None
        self.hw = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            pw = next((f for f in (os.path.join(d, "power1_average"), os.path.join(d, "power1_input")) if os.access(f, os.R_OK)), None)
            fq = os.path.join(d, "freq1_input")
            if pw and os.access(fq, os.R_OK):
                self.hw.append((pw, fq))
        self.source = "hwmon" if self.hw else "rocm-smi"
        self._threading = threading

    def _read(self):
        import re
        if self.hw:
            best = None
            for pw, fq in self.hw:             # several cards visible: the loaded one is the one this process drives
                try:
                    w = int(open(pw).read()) * 1e-6
                    mhz = int(open(fq).read()) * 1e-6
                except Exception:
                    continue
                if best is None or w > best[0]:
                    best = (w, mhz)
            return best
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.search(r"Power \(W\):\s*([0-9.]+)", out); ck = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
            return (float(pw.group(1)), float(ck.group(1))) if pw and ck else None
        except Exception:
            return None

    def start(self):
        def run():
            while not self._stop.is_set():
                r = self._read()
                if r:
                    self.samples.append(r)
                self._stop.wait(self.period)
        self._th = self._threading.Thread(target=run, daemon=True)
        self._th.start()

    def stop(self):
        self._stop.set()
        if self._th:
            self._th.join(timeout=10)
        if not self.samples:
            return {"sclk_mhz_mean": None, "sclk_mhz_min": None, "power_w_mean": None, "samples": 0, "sampler": self.source}
        n = len(self.samples)
        return {"sclk_mhz_mean": sum(s[1] for s in self.samples) / n, "sclk_mhz_min": min(s[1] for s in self.samples),
                "power_w_mean": sum(s[0] for s in self.samples) / n, "power_w_max": max(s[0] for s in self.samples), "samples": n,
                "sampler": self.source}


def load_calibration_reference():
    """The committed probe readings of the box every headline is normalised to (profiles/calibration_reference.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "calibration_reference.json")) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world == 1:
        # convenience self-launch, one rank per GPU.  Only legal while nothing in this process has touched the GPU: under a
        # profiler (rocprofv3 preloads a tool library that initialises the GPU before main) this hop would be the forbidden
        # exec-after-GPU-init on this pool -- wrap the per-rank python inside torchrun instead (profiles/README.md).
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import rdm_amd  # noqa: F401  (no GPU touched by the import)
        from rdm_amd import parallel
        sys.exit(parallel.safe_self_launch(__file__, a.gpus, sys.argv[1:], os.environ.get("MASTER_PORT", "29533")))

    import numpy as np
    import torch
    import rdm_amd  # noqa: F401
    from rdm_amd import _lib, packing, parallel, synthetic
    from rdm_amd.models.diffusion.ddpm import MinimalRETRODiffusion

    torch.set_grad_enabled(False)
    rank, local = parallel.init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ctx = _lib.Context(local)
    # N > 1 on RCCL: the image all-gather runs through the library's own communicator (C ABI rdm_comm_all_gather); gloo test groups
    # (ranks sharing one GPU) and a failed communicator keep torch.distributed
    lib_comm = parallel.attach_library_comm(ctx) if world > 1 else False
    collective = (None if world == 1 else "rdm_comm_all_gather (RCCL through the C ABI, library stream)" if lib_comm else
                  f"torch.distributed all_gather ({torch.distributed.get_backend()})")

    def gather(img):
        if a.gather == "uint8":
            img = ctx.to_uint8(img)                                                   # [b,256,256,3] uint8: clamp, (x+1)/2*255, truncation
        return parallel.all_gather_images(img, world * B, ctx=ctx if lib_comm else None)      # no-op for one rank

    # ---- model: shipped architecture, seeded random weights; schedule from the product's register_schedule
    model = MinimalRETRODiffusion(unet_config={"params": {}}, first_stage_config={"params": {"ddconfig": {}}}, k_nn=a.k, ctx=ctx)
    if a.config == 5:
        rcfg, vcfg = _lib.make_rarm_cfg(), _lib.make_vqgan_f16_cfg()
        ctx.load_rarm(rcfg, packing.pack("rarm", rcfg, synthetic.rarm_state_dict(rcfg)))
        ctx.load_vq(vcfg, packing.pack("vq", vcfg, synthetic.vq_state_dict(vcfg, synthetic.VQGAN_SEED)))
        if not a.ddim_steps_given:
            a.ddim_steps = 256                         # tokens per image (16 x 16 codes); fewer only for short profiling passes (the image is then incomplete)
    else:
        model.load_unet_state_dict(synthetic.unet_state_dict(model.unet_cfg))
        model.load_first_stage_state_dict(synthetic.vq_state_dict(model.vq_cfg))
    sched_ddpm = {n: getattr(model, n).numpy() for n in ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
                                                         "posterior_mean_coef2", "posterior_log_variance_clipped")}
    N, D, B, k = a.db_rows, 512, a.batch, a.k
    total_steps = a.warmup + a.steps
    lo, hi = parallel.shard_range(world * B, world, rank)            # this rank's rows of the global batch

    if a.config == 2:
        # ---- text-only: ViT-B/32 text tower, synthetic token rows <start> w.. <end> 0..
        ccfg = _lib.make_clip_cfg()
        ctx.load_clip(ccfg, packing.pack("clip", ccfg, synthetic.clip_state_dict(ccfg)))
        g = torch.Generator().manual_seed(13)
        toks = torch.zeros((total_steps, world * B, ccfg.context_length), dtype=torch.int64)
        for s_ in range(total_steps):
            for r in range(world * B):
                L = int(torch.randint(6, 20, (1,), generator=g))
                toks[s_, r, 1:L - 1] = torch.randint(1000, 40000, (L - 2,), generator=g)
                toks[s_, r, 0] = ccfg.vocab_size - 2; toks[s_, r, L - 1] = ccfg.vocab_size - 1
        toks = toks[:, lo:hi].contiguous().to(dev)
    else:
        # ---- retrieval DB: synthetic CLIP-like rows, generated on device, replicated per GPU
        gen = torch.Generator(device=dev).manual_seed(7)
        db = torch.empty((N, D), device=dev, dtype=torch.float16)
        for r0 in range(0, N, 1 << 20):
            r1 = min(N, r0 + (1 << 20))
            db[r0:r1] = (torch.randn((r1 - r0, D), device=dev, generator=gen) * 0.45).half()
        ctx.db_load(db)
        del db
        torch.cuda.empty_cache()
        queries = torch.stack([parallel.per_sample_noise(11 + s_, range(lo, hi), (D,), device=dev) * 0.45 for s_ in range(total_steps)])
    x_Ts = torch.stack([parallel.per_sample_noise(1000 + s_, range(lo, hi), (3, 64, 64), device=dev) for s_ in range(total_steps)])
    uncond = torch.zeros((B, k, D), device=dev)

    def step(i, scale=None):
        scale = a.scale if scale is None else scale
        if a.config == 2:
            cond = ctx.clip_encode_text(toks[i])[:, None].contiguous()                # [B,1,512] (rdm_sample.py:276-277, k_nn = 1)
        else:
            q = queries[i]
            idx, _ = ctx.knn(q, k)
            nbrs = ctx.db_gather(idx, D)                                              # [B,k,512] raw neighbour embeddings
            cond = torch.cat([q[:, None], nbrs[:, :k - 1]], dim=1).contiguous()       # ddpm.py:775 (query first)
        if a.config == 5:
            u = parallel.per_sample_noise(7000 + i, range(lo, hi), (a.ddim_steps,), device=dev)          # N(0,1) -> uniform via the normal CDF
            u = (0.5 * (1.0 + torch.erf(u * 0.7071067811865476))).clamp(0.0, 0.99999994).t().contiguous()
            sos = torch.full((B, 1), 16385, dtype=torch.long, device=dev)
            tok = ctx.rarm_sample(sos, nbrs, a.ddim_steps, u, temperature=1.0, top_k=256, guidance_scale=1.0)   # RARM: query NOT prepended
            if tok.shape[1] < 256:                     # short profiling pass (--ddim-steps < 256): the remaining codes are padding
                tok = torch.nn.functional.pad(tok, (0, 256 - tok.shape[1]))
            img = ctx.vq_decode_indices(tok)
            return gather(img)
        if a.config == 4:
            noise = parallel.per_sample_noise(5000 + i, range(lo, hi), (a.ddim_steps, 3, 64, 64), device=dev).transpose(0, 1).contiguous()
            z = ctx.ddpm_sample(a.ddim_steps, x_Ts[i], cond, noise, sched_ddpm, clip_denoised=True)
        else:
            z, _, _ = ctx.ddim_sample(a.ddim_steps, x_Ts[i], cond, uncond if scale > 1 else None, model.alphas_cumprod, eta=0.0, scale=scale)
        img = ctx.vq_decode(z)
        return gather(img)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- box calibration (untimed, < 3 s): fixed MFMA and HBM-stream probes that do not change with the product kernels -- boxes of the
    # pool differ by +-4..5 % under the power cap and the headline moves with them; `value_at_reference_box` rescales by these readings
    calib = None
    if not a.no_calibration:
        try:
            tf_, gb_ = ctx.calib_probe(mfma_ms=800.0, stream_bytes=1 << 30, stream_reps=6)
            calib = {"mfma_probe_tflops": tf_, "hbm_stream_gbps": gb_}
        except Exception as e_:                     # a failed probe must never cost the benchmark line
            calib = {"mfma_probe_tflops": None, "hbm_stream_gbps": None, "error": str(e_)[:200]}
        if world > 1 and calib.get("mfma_probe_tflops"):
            tt = torch.tensor([calib["mfma_probe_tflops"], calib["hbm_stream_gbps"]], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MIN)          # the slowest rank sets the step
            calib["mfma_probe_tflops_min_over_ranks"], calib["hbm_stream_gbps_min_over_ranks"] = float(tt[0]), float(tt[1])

    for i in range(a.warmup):
        step(i)
    fence()
    ctx.prof_reset()
    box_sampler = BoxSampler() if (calib is not None and rank == 0) else None
    if box_sampler:
        box_sampler.start()
    # dominant kernel: HIP events (on the library's stream) around its launches INSIDE the timed region, on the LAST timed step only --
    # 2 524 event pairs per step on every step cost the headline what a sampled step measures just as well (a host-side flag: no sync)
    t0 = time.perf_counter()
    for i in range(a.warmup, total_steps):
        if i == total_steps - 1:
            ctx.prof_enable((_lib.PROF_CONV3X3,))
        img = step(i)
    fence()
    dt = time.perf_counter() - t0
    if box_sampler:
        calib.update(box_sampler.stop())
    ctx.prof_enable(())
    rank_ms = None
    if world > 1:
        # max over ranks is the job's time (contract); every rank's own time rides along so that a straggler is visible in the line
        allt = torch.zeros(world, device=dev, dtype=torch.float64)
        allt[rank] = dt
        torch.distributed.all_reduce(allt, op=torch.distributed.ReduceOp.SUM)
        rank_ms = [float(v) / a.steps * 1e3 for v in allt.tolist()]
        dt = float(allt.max().item())
    assert img.shape[0] == world * B and bool(torch.isfinite(img.float()).all()), "non-finite images"
    if a.dump_images and rank == 0:
        np.save(a.dump_images, img.cpu().numpy())
    n_conv, ms_conv, fl_conv = ctx.prof_collect(_lib.PROF_CONV3X3)

    # ---- one extra UNTIMED step with events around every other kernel class (their event records would otherwise sit in
    # the timed region: ~25 k launches per step)
    classes = {}
    extras = {}
    if not a.no_extras:
        ctx.prof_reset()
        ctx.prof_enable((_lib.PROF_LINEAR, _lib.PROF_KNN, _lib.PROF_ATTENTION, _lib.PROF_GROUPNORM, _lib.PROF_LAYERNORM, _lib.PROF_UPSCONV))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        step(0)
        fence()
        t_prof_step = time.perf_counter() - t1
        ctx.prof_enable(())
        for name, kind in (("linear_gemm", _lib.PROF_LINEAR), ("knn", _lib.PROF_KNN), ("flash_attention", _lib.PROF_ATTENTION),
                           ("groupnorm", _lib.PROF_GROUPNORM), ("layernorm", _lib.PROF_LAYERNORM), ("upsample_conv_by_phase", _lib.PROF_UPSCONV)):
            n_, ms_, w_ = ctx.prof_collect(kind)
            classes[name] = (n_, ms_, w_)
        extras["untimed_profiled_step_ms"] = t_prof_step * 1e3
        lin_shapes = {}
        if a.config == 5:                          # the decode GEMMs by shape (M, N, K): one launch group per kernel instantiation + shape
            import csv, tempfile
            with tempfile.TemporaryDirectory() as td:
                ctx.prof_dump(os.path.join(td, "recs.csv"))
                with open(os.path.join(td, "recs.csv")) as f:
                    for r_ in csv.DictReader(f):
                        if int(r_["kind"]) == _lib.PROF_LINEAR:
                            g_ = lin_shapes.setdefault((int(r_["d0"]), int(r_["d1"]), int(r_["d2"])), [0, 0.0, 0.0])
                            g_[0] += 1; g_[1] += float(r_["ms"]); g_[2] += float(r_["work"])
        if world == 1 and a.config in (2, 3):      # single-process only (step() holds the collective when world > 1)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            step(0, scale=1.0)
            torch.cuda.synchronize()
            extras["images_per_s_guidance_scale_1"] = B / (time.perf_counter() - t1)

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process -- taken from the committed
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (gfx950 corrections applied; profiles/README.md)
    traffic, traffic_src, pmc = None, None, {}
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc_v4.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pmc = json.load(f)
            traffic = pmc["hbm_bytes_per_launch"]; traffic_src = name
            break
        except Exception:
            pmc = {}

    if rank == 0:
        images = world * B * a.steps
        achieved = fl_conv / (ms_conv * 1e-3) / 1e12 if ms_conv > 0 else 0.0
        sampler = (f"{a.ddim_steps}-step ancestral DDPM (p_sample_loop, no CFG)" if a.config == 4
                   else "256 autoregressive tokens (RARM transformer 18 x 768 with K/V cache, top-k 256, guidance 1.0)" if a.config == 5
                   else f"{a.ddim_steps}-step DDIM (eta 0, CFG scale {a.scale:.1f})")
        front = ("CLIP ViT-B/32 text tower on 64 captions (no retrieval, k=1)" if a.config == 2
                 else f"exact kNN (k={k}) over a synthetic {N} x 512 fp16 CLIP DB")
        roof = {"kernel": "conv3x3_halo4_kernel<3> + igemm_kernel<..,conv> (3x3 conv, bf16 MFMA, fp32 accumulate)", "bound": "mfma",
                "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0, "traffic": traffic,
                "traffic_note": f"bytes per launch of the dominant conv kernel from the committed rocprofv3 PMC passes (profiles/{traffic_src}), not collected in this run",
                "launches": n_conv, "avg_launch_ms": ms_conv / max(n_conv, 1),
                "algorithmic_tflop_per_launch": fl_conv / max(n_conv, 1) / 1e12, "conv_time_frac_of_step": ms_conv * 1e-3 / (dt / a.steps)}      # events cover the LAST timed step only
        for name, (n_, ms_, w_) in classes.items():
            if n_ == 0:
                continue
            if name == "upsample_conv_by_phase":
                # Upsample's nearest-2x + conv3x3 as four 2x2-tap convs at source resolution (igemm_kernel<.., 3>): EXECUTED FLOPs = 4/9 of the
                # nine-tap count the reference's F.interpolate + conv2d performs
                ach = w_ / (ms_ * 1e-3) / 1e12
                roof[name] = {"kernel": "igemm_kernel<BM, BN, W, 3> (pre-summed phase weights)", "bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s",
                              "frac": ach / 2500.0, "launches": n_, "time_ms_per_step": ms_, "nine_tap_equivalent_tflops": ach * 2.25}
                continue
            if name in ("linear_gemm", "flash_attention"):
                roof[name] = {"bound": "mfma", "achieved": w_ / (ms_ * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                              "frac": w_ / (ms_ * 1e-3) / 1e12 / 2500.0, "launches": n_, "time_ms_per_step": ms_}
                if name == "linear_gemm":
                    roof[name]["kernel"] = "lin4_kernel<GEGLU, WM> (big-M projections) + igemm_kernel / sgemm_kernel (the rest)"
                # HBM bytes per launch and MFMA-busy share per kernel of the class, from the same committed PMC passes as `traffic`
                per_kernel = pmc.get("linear" if name == "linear_gemm" else "flash_attention", {})
                if per_kernel and a.config != 5:          # (the passes are of the default config: not the RARM decode's kernels)
                    roof[name]["traffic"] = {k_: {"hbm_bytes_per_launch": v_.get("hbm_bytes_per_launch"), "mfma_busy_frac": v_.get("mfma_busy_frac")}
                                             for k_, v_ in per_kernel.items()}
                    roof[name]["traffic_note"] = f"per kernel, profiles/{traffic_src}"
            else:
                roof[name] = {"bound": "hbm", "achieved": w_ / (ms_ * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": w_ / (ms_ * 1e-3) / 1e9 / 8000.0, "launches": n_, "time_ms_per_step": ms_,
                              "algorithmic_bytes_per_launch": w_ / n_}
        if a.config != 5:
            # end to end: ALGORITHMIC FLOPs per image (g S F_unet(k) + F_dec + F_vq) x images/s against the dense bf16 peak
            g_ = 2 if (a.config != 4 and a.scale > 1) else 1
            flop_img = g_ * a.ddim_steps * (208.56e9 + (k - 4) * 0.034e9) + 670.58e9 + 0.20e9
            roof["end_to_end_tflops"] = flop_img * images / dt / 1e12
            roof["end_to_end_frac"] = roof["end_to_end_tflops"] / 2500.0
            roof["algorithmic_tflop_per_image"] = flop_img / 1e12
            # SpatialTransformer block (attention.py:122-196) as one class: its GEMMs (incl. the fused cross-attention and, not separable
            # by the event kinds, the ResBlocks' 1x1 skip_connection GEMMs), self-attention and LayerNorms; its entry GroupNorm sits in `groupnorm`
            st = [classes.get(n_) for n_ in ("linear_gemm", "flash_attention", "layernorm")]
            if all(c_ is not None and c_[0] > 0 for c_ in st):
                st_ms = sum(c_[1] for c_ in st)
                st_fl = st[0][2] + st[1][2]
                roof["spatial_transformer_block"] = {"bound": "mfma", "achieved": st_fl / (st_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                                                     "frac": st_fl / (st_ms * 1e-3) / 1e12 / 2500.0, "time_ms_per_step": st_ms,
                                                     "classes": ["linear_gemm", "flash_attention", "layernorm"]}
        if a.config == 5 and "linear_gemm" in roof:
            # RARM: a token step is ~110 dependent launches: the decode GEMMs (M = batch rows against 768..6144-row weight matrices: weight streaming;
            # bound HBM: N K 2 bytes per launch) and the K/V-cache attention (bound HBM: the cache rows 0..pos of every sequence).  The top-level object
            # describes whichever takes more of the step (at 64 sequences the GEMMs, latency-bound; at 512 the attention, bandwidth-bound); the
            # other sits beside it.  Measured on the extra UNTIMED step (one event pair per launch for ~28 k launches would otherwise sit in the timed region).
            n_, ms_, w_ = classes["linear_gemm"]
            wbytes = w_ / B                                  # 2 M N K FLOP / M rows = 2 N K bytes of bf16 weights
            step_s = dt / a.steps
            conv_roof = {k_: roof[k_] for k_ in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_ms", "conv_time_frac_of_step")}
            lin = {"kernel": "sgemm_kernel<MA,NB,U> (decode-step linear layers: M = batch rows, weights streamed once per launch)",
                   "bound": "hbm", "achieved": wbytes / (ms_ * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                   "frac": wbytes / (ms_ * 1e-3) / 1e9 / 8000.0, "traffic": None, "launches": n_, "avg_launch_ms": ms_ / max(n_, 1),
                   "algorithmic_bytes_per_launch": wbytes / max(n_, 1), "time_frac_of_step": ms_ * 1e-3 / step_s,
                   "note": "weight bytes only: at big batches these launches are bound by latency and MFMA work per launch, not by the weight stream"}
            att = None
            if classes.get("flash_attention", (0, 0, 0))[0] > 0:
                na_, msa_, wa_ = classes["flash_attention"]
                att = {"kernel": "rarm_decode_attention_kernel (one query row per sequence against its K/V cache rows 0..pos, 18 layers x 256 steps)",
                       "bound": "hbm", "achieved": wa_ / (msa_ * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": wa_ / (msa_ * 1e-3) / 1e9 / 8000.0,
                       "traffic": None, "launches": na_, "avg_launch_ms": msa_ / max(na_, 1), "algorithmic_bytes_per_launch": wa_ / max(na_, 1),
                       "time_frac_of_step": msa_ * 1e-3 / step_s}
            # The dominant KERNEL (contract: "the dominant kernel"): the linear class is several kernels (sgemm tile instantiations, the tiled GEGLU
            # GEMM) over five shapes per layer; its largest launch group (one shape) is compared with the attention kernel.  A GEMM of M rows
            # against an [N, K] weight has ~M FLOP per weight byte: weight-stream (HBM) bound below the ridge (2500 / 8 = 312 rows), MFMA above.
            big = None
            lin_shapes = {k_: v_ for k_, v_ in lin_shapes.items() if k_[0] > 0 and k_[1] > 0 and k_[2] > 0}
            if lin_shapes:
                (m_, n2_, k_), (cnt_, gms_, gfl_) = max(lin_shapes.items(), key=lambda kv: kv[1][1])
                hbm_ = m_ < 312
                ach_ = (cnt_ * 2.0 * n2_ * k_ / (gms_ * 1e-3) / 1e9) if hbm_ else gfl_ / (gms_ * 1e-3) / 1e12
                big = {"kernel": f"sgemm_kernel / lin4_kernel, decode GEMM [{m_} x {k_}] x [{n2_} x {k_}]^T (largest launch group of the linear class)",
                       "bound": "hbm" if hbm_ else "mfma", "achieved": ach_, "peak": 8000.0 if hbm_ else 2500.0, "unit": "GB/s" if hbm_ else "TFLOP/s",
                       "frac": ach_ / (8000.0 if hbm_ else 2500.0), "traffic": None, "launches": cnt_, "avg_launch_ms": gms_ / max(cnt_, 1),
                       "time_frac_of_step": gms_ * 1e-3 / step_s}
                lin["largest_shape"] = big
            for k_ in list(roof):
                roof.pop(k_)
            lin_top = big["time_frac_of_step"] if big else lin["time_frac_of_step"]
            if att and att["time_frac_of_step"] > lin_top:
                roof.update(att); roof["decode_linear"] = lin
            else:
                roof.update(big or lin); roof["decode_linear"] = lin
                if att:
                    roof["kv_cache_attention"] = att
            roof["vqgan_conv"] = conv_roof
        out = {
            "metric": "images/sec at 256x256, 50 DDIM steps, k=4 OpenImages retrieval",
            "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"BASELINE config #5: RARM sampling, {front} -> {sampler} -> VQGAN-f16 decode to 256x256 (random weights)" if a.config == 5 else
                                    f"BASELINE config #{a.config}: RDM sampling, {front} -> {sampler} over the shipped-config UNet "
                                    "(400.9M params, random weights) -> VQ-f4 decode to 256x256"),
                       "baseline_config": a.config, "batch_per_gpu": B, "global_batch": world * B, "sampler_steps": a.ddim_steps, "k": k,
                       "guidance_scale": None if a.config == 4 else (1.0 if a.config == 5 else a.scale), "db_rows": None if a.config == 2 else N,
                       "parallelism": f"dp{world} (batch-sharded, weights + DB replicated, all-gather of images only)",
                       "collective": collective, "gathered": None if world == 1 else f"{a.gather} images, {img.element_size() * img[0].numel() * B} bytes per rank",
                       "deterministic_mode": bool(ctx.deterministic)},
            "roofline": roof,
        }
        if rank_ms is not None:
            out["config"]["rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms}
        if calib is not None:
            ref = load_calibration_reference()
            calib["reference"] = ref
            # What differs between boxes of the pool (measured round 6, profiles/r06_calibration_boxes.md): NOT the peak capabilities -- the pure-MFMA
            # probe (current-limited, ~1.74 GHz on every box) and the HBM stream agree to < 1 % -- but the shader clock a box SUSTAINS under this
            # workload's mixed load (2.07 .. 2.18 GHz through the timed region).  The step time follows that clock for the clock-bound share s of
            # the step (fitted on same-tree runs on different boxes; the rest is HBM / fabric time):
            #     time here = time on the reference box x [ s x (sclk reference / sclk here) + (1 - s) ]
            sclk = calib.get("sclk_mhz_mean")
            if ref and sclk and ref.get("sclk_mhz_mean") and a.config != 5:
                s_ = float(ref.get("clock_bound_share", 0.6))
                rel = s_ * (ref["sclk_mhz_mean"] / sclk) + (1.0 - s_)
                calib["clock_bound_share"] = s_
                calib["value_scale_to_reference_box"] = rel
                out["value_at_reference_box"] = out["value"] * rel
            extras["calibration"] = calib
            out["calibration"] = calib
        if extras:
            out["extras"] = extras
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_full(a.scale) if a.full_cpu_baseline else cpu_baseline(a.scale)
            if not a.full_cpu_baseline:
                # the bounded sample above is an extrapolation; the full SURVEY-8d protocol (--full-cpu-baseline, ~5 min) as last committed
                for name in ("r05_bench_full_cpu_baseline.json", "r04_bench_full_cpu_baseline.json"):
                    try:
                        with open(os.path.join(ROOT, "profiles", name)) as f:
                            full = json.load(f)["cpu_baseline"]
                        out["cpu_baseline"]["full_protocol_committed"] = {"value": full["value"], "unit": full["unit"], "cores": full["cores"],
                                                                          "sample": full["sample"], "source": f"profiles/{name} (another box of the same pool)"}
                        break
                    except Exception:
                        pass
        print(json.dumps(out), flush=True)
    ctx.close()
    parallel.shutdown()


if __name__ == "__main__":
    main()
