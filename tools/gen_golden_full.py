#!/usr/bin/env python3
"""Full-size golden vectors for the BENCHMARKED pipeline (BASELINE configs #2, #3, #4), generated in the build
container from the reference's in-tree classes (`UNetModel`, `CLIP`) driven by the oracle's sampling loops.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_full.py [ddim4 ddim1 ddpm16 vq clip knn]

What is written (tests/golden/full_*.npz; weights are NOT stored, they are re-derived from the seeds):
  full_ddim_k4.npz   50-step DDIM (eta 0, CFG 2.0, zero unconditional context) of the shipped-config UNet, B=1, k=4:
                     x_T, cond, x_prev / pred_x0 / eps after loop iterations CHECK_DDIM, final latent     (config #3)
  full_ddim_k1.npz   the same at k=1 (query only)                                                          (config #2)
  full_ddpm_k16.npz  250-step ancestral p_sample_loop(timesteps=250), k=16, noise from default_rng(NOISE_SEED):
                     x after loop iterations CHECK_DDPM, final latent                                      (config #4)
  full_vq.npz        shipped-spec VQ-f4 decode of a fixed latent: code indices + image (fp16)
  full_clip.npz      ViT-B/32 text embeddings of 3 captions and image embeddings of 2 seeded images
  full_ddim_k4_b.npz / full_ddpm_k16_b.npz / full_vq_b.npz   (round 4; `ddim4b ddpm16b vqb`) a SECOND trajectory / latent from other
                     seeds, stored lean (states after the check iterations + final): the golden row placed at batch index 63 of the
                     batch-64 tests (last tile, wrapped-skip partner) beside row 0

The eps-model of every trajectory is the REFERENCE class (rdm/modules/diffusionmodules/openaimodel.py:36-371) —
the oracle UNet is additionally asserted equal on the first step.  DDIM update / schedule: oracle.diffusion (restates
rdm/models/diffusion/ddim.py:142-268, pinned by SURVEY appendix C); DDPM update and VQ decoder: oracle restatements of
un-vendored ldm/taming code (parity unpinned).
"""
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tools", "ldm_shim"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import clip as oclip
from oracle import diffusion as odiff
from oracle import unet as ounet
from oracle import vqdecoder as ovq

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)

CHECK_DDIM = (0, 10, 25, 49)          # loop iterations i (t = 981 - 20 i) whose state is stored
CHECK_DDPM = (0, 50, 125, 249)        # loop iterations n (t = 249 - n)
NOISE_SEED = 2024
UNET_SEED, VQ_SEED, CLIP_SEED = 1234, 4321, 99
CAPTIONS = ["a happy bear reading a newspaper, oil on canvas", "A photo of a dog.", "an armchair in the shape of an avocado"]


def ref_unet():
    from rdm.modules.diffusionmodules.openaimodel import UNetModel
    spec = ounet.shipped_spec()
    m = UNetModel(image_size=64, in_channels=spec.in_channels, out_channels=spec.out_channels,
                  model_channels=spec.model_channels, attention_resolutions=list(spec.attention_resolutions),
                  num_res_blocks=spec.num_res_blocks, channel_mult=list(spec.channel_mult),
                  num_head_channels=spec.num_head_channels, use_spatial_transformer=True, transformer_depth=1,
                  context_dim=spec.context_dim, use_checkpoint=True).eval()
    sd = ounet.synth_state_dict(ounet.param_shapes(spec), seed=UNET_SEED)
    m.load_state_dict(sd, strict=True)
    return m, sd, spec


def inputs(k, seed):
    rng = np.random.default_rng(seed)
    x_T = torch.from_numpy(rng.standard_normal((1, 3, 64, 64)).astype(np.float32))
    cond = torch.from_numpy((rng.standard_normal((1, k, 512)) * 0.45).astype(np.float32))
    return x_T, cond


def gen_ddim(k, tag, seed, lean=False):
    """lean: store only the states after CHECK_DDIM and the final latent (the second golden row of the batch-64 tests)"""
    m, sd, spec = ref_unet()
    apply_ref = lambda x, t, c: m(x, t, context=[c])
    x_T, cond = inputs(k, seed)
    uncond = torch.zeros_like(cond)
    sched = odiff.Schedule()
    sch = odiff.ddim_schedule(sched, 50, 0.0)
    ts = sch[0]
    total = ts.shape[0]
    img = x_T
    keep = {}
    t0 = time.time()
    for i, step in enumerate(np.flip(ts)):
        index = total - i - 1
        t = torch.full((1,), int(step), dtype=torch.long)
        if i == 0:      # the oracle UNet equals the reference class (pin), checked once per trajectory at full size
            e_ref = apply_ref(torch.cat([img] * 2), torch.cat([t] * 2), torch.cat([cond, uncond]))
            e_or = ounet.unet_forward(sd, spec, torch.cat([img] * 2), torch.cat([t] * 2), torch.cat([cond, uncond]))
            err = (e_ref - e_or).abs().max().item()
            print(f"[{tag}] step 0 max|oracle - reference| = {err:.3e}")
            assert err <= 1e-4
        x_in = img
        img, pred_x0 = odiff.p_sample_ddim(apply_ref, img, cond, t, index, sch, scale=2.0, uc=uncond)
        if i in CHECK_DDIM:
            keep[f"x_{i}"] = img.numpy()
            if not lean:
                keep[f"xin_{i}"] = x_in.numpy(); keep[f"px0_{i}"] = pred_x0.numpy()
        if i % 10 == 0:
            print(f"[{tag}] step {i} t={int(step)} |x|={img.norm():.3f} ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(OUT, f"full_{tag}.npz"), x_T=x_T.numpy(), cond=cond.numpy(), z=img.numpy(),
                        steps=np.asarray(CHECK_DDIM), scale=np.float32(2.0), **keep)


def ddpm_noise(T, shape):
    return torch.from_numpy(np.random.default_rng(NOISE_SEED).standard_normal((T,) + tuple(shape)).astype(np.float32))


def gen_ddpm(k=16, T=250, seed=31, tag=None, noise_seed=NOISE_SEED):
    m, sd, spec = ref_unet()
    apply_ref = lambda x, t, c: m(x, t, context=[c])
    x_T, cond = inputs(k, seed)
    sched = odiff.Schedule()
    noise = torch.from_numpy(np.random.default_rng(noise_seed).standard_normal((T,) + tuple(x_T.shape)).astype(np.float32))
    img = x_T
    keep = {}
    t0 = time.time()
    for n, i in enumerate(reversed(range(T))):
        t = torch.full((1,), i, dtype=torch.long)
        x_in = img
        img = odiff.p_sample_ddpm(apply_ref, sched, img, cond, t, noise[n], True)
        if n in CHECK_DDPM:
            keep[f"x_{n}"] = img.numpy()
            if tag is None: keep[f"xin_{n}"] = x_in.numpy()
        if n % 25 == 0:
            print(f"[ddpm_k{k}] n={n} t={i} |x|={img.norm():.3f} ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(OUT, f"full_ddpm_k{k}{tag or ''}.npz"), x_T=x_T.numpy(), cond=cond.numpy(), z=img.numpy(),
                        steps=np.asarray(CHECK_DDPM), noise_seed=np.int64(noise_seed), timesteps=np.int64(T), **keep)


def gen_vq(seed=55, tag=""):
    vs = ovq.shipped_vq_spec()
    sd = ounet.synth_state_dict(ovq.vq_param_shapes(vs), seed=VQ_SEED)
    rng = np.random.default_rng(seed)
    z = torch.from_numpy((rng.standard_normal((1, 3, 64, 64)) * 0.6).astype(np.float32))
    torch.set_num_threads(8)
    img, idx = ovq.vq_decode(sd, vs, z, return_indices=True)
    e = sd["quantize.embedding.weight"]
    flat = z.permute(0, 2, 3, 1).reshape(-1, 3)
    d = ((flat[:, None, :].double() - e[None].double()) ** 2).sum(-1)
    top2 = d.topk(2, dim=1, largest=False).values
    print(f"[vq] image |.|max {img.abs().max():.3f}; distinct codes {idx.unique().numel()}; "
          f"min margin between best two codes {float((top2[:, 1] - top2[:, 0]).min()):.3e}")
    np.savez_compressed(os.path.join(OUT, f"full_vq{tag}.npz"), z=z.numpy(), indices=idx.numpy().astype(np.int32),
                        image=img.numpy().astype(np.float16), seed=np.int64(VQ_SEED))


def clip_images(spec):
    rng = np.random.default_rng(CLIP_SEED + 7)
    return torch.from_numpy(rng.standard_normal((2, 3, spec.image_resolution, spec.image_resolution)).astype(np.float32))


def gen_clip():
    from rdm.modules.custom_clip.model import CLIP
    from rdm.modules.custom_clip.simple_tokenizer import SimpleTokenizer
    spec = oclip.vitb32_spec()
    m = CLIP(spec.embed_dim, spec.image_resolution, spec.vision_layers, spec.vision_width, spec.vision_patch_size,
             spec.context_length, spec.vocab_size, spec.transformer_width, spec.transformer_heads,
             spec.transformer_layers).eval()
    shapes = oclip.clip_param_shapes(spec)
    sd = ounet.synth_state_dict(shapes, seed=CLIP_SEED)
    m.load_state_dict({**sd, "logit_scale": torch.ones([])})
    tk = SimpleTokenizer()
    sot, eot = tk.encoder["<|startoftext|>"], tk.encoder["<|endoftext|>"]
    tokens = np.zeros((len(CAPTIONS), 77), dtype=np.int64)
    for i, c in enumerate(CAPTIONS):
        ids = [sot] + tk.encode(c) + [eot]
        tokens[i, :len(ids)] = ids
    tokens = torch.from_numpy(tokens)
    img = clip_images(spec)
    t_ref, i_ref = m.encode_text(tokens), m.encode_image(img)
    t_or, i_or = oclip.encode_text(sd, spec, tokens), oclip.encode_image(sd, spec, img)
    print(f"[clip] oracle vs reference: text {(t_or - t_ref).abs().max():.3e}, image {(i_or - i_ref).abs().max():.3e}")
    np.savez_compressed(os.path.join(OUT, "full_clip.npz"), captions=np.array(CAPTIONS), tokens=tokens.numpy(),
                        text_out=t_ref.numpy(), image_out=i_ref.numpy(), seed=np.int64(CLIP_SEED))


if __name__ == "__main__":
    what = sys.argv[1:] or ["clip", "vq", "ddim4", "ddim1", "ddpm16"]
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    for w in what:
        t0 = time.time()
        {"clip": gen_clip, "vq": gen_vq, "ddim4": lambda: gen_ddim(4, "ddim_k4", 21),
         "ddim1": lambda: gen_ddim(1, "ddim_k1", 22), "ddpm16": gen_ddpm,
         # second golden rows (round 4): other inputs, placed at batch index 63 of the batch-64 tests (tests/test_gpu_full.py)
         "ddim4b": lambda: gen_ddim(4, "ddim_k4_b", 23, lean=True), "ddpm16b": lambda: gen_ddpm(seed=33, tag="_b", noise_seed=NOISE_SEED + 1),
         "vqb": lambda: gen_vq(seed=56, tag="_b")}[w]()
        print(f"== {w} done in {time.time() - t0:.0f} s", flush=True)
