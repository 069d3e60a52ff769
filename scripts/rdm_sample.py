#!/usr/bin/env python3
"""Native counterpart of the reference's scripts/rdm_sample.py (same flags and defaults, :22-143): load config +
checkpoint, CLIP-encode the caption, retrieve k_nn neighbours, sample with DDIM + classifier-free guidance, decode,
write PNGs named like the reference ({start}-{key}-run{n}-sample{i}.png, :256, :304).

Differences on purpose: `--seed` works (the reference crashes on opt.r_runs, :141), nothing is forced
to the string "cuda", and the searcher is exact brute force on the GPU instead of ScaNN.
`--synthetic` runs the same flow with seeded random weights / DB so the script can be exercised without the 6.2 GB
checkpoints and the 18 GB database (no network in the build environment).
"""
import argparse
import glob
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rdm_amd  # noqa: E402,F401
from rdm_amd.data.retrieval_dataset.dsetbuilder import DatasetBuilder  # noqa: E402
from rdm_amd.models.diffusion.ddpm import MinimalRETRODiffusion  # noqa: E402
from rdm_amd.modules.retrievers import CLIPTextEmbedder, ClipImageRetriever  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("-r", "--resume", type=str, default="models/rdm/imagenet", help="model dir with config.yaml + model.ckpt")
    p.add_argument("-n", "--n_runs", type=int, default=2)
    p.add_argument("-c", "--caption", type=str, default="")
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("-e", "--eta", type=float, default=1.0)
    p.add_argument("-l", "--logdir", type=str, default="none")
    p.add_argument("-s", "--steps", type=int, default=100)
    p.add_argument("-bs", "--batch_size", type=int, default=4)
    p.add_argument("--k_nn", type=int, default=4)
    p.add_argument("--guidance_scale", type=float, default=2.0)
    p.add_argument("--top_m", type=float, default=0.01)
    p.add_argument("--use_weights", action="store_true")
    p.add_argument("--only_caption", action="store_true")
    p.add_argument("--omit_query", action="store_true")
    p.add_argument("--unconditional", action="store_true")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--clip_ckpt", type=str, default=None, help="CLIP ViT-B/32 state_dict (.pt)")
    p.add_argument("--synthetic", action="store_true")
    opt = p.parse_args()
    if opt.top_m > 1:
        opt.top_m = int(opt.top_m)                # rdm_sample.py:138-140
    return opt


def custom_to_pil(x):
    """rdm_sample.py:203-219: clamp(-1,1) -> (x+1)/2 -> HWC -> *255 -> uint8 (truncation)."""
    from PIL import Image
    x = x.detach().cpu().float().clamp(-1., 1.)
    x = ((x + 1.) / 2.).permute(1, 2, 0).numpy()
    return Image.fromarray((255 * x).astype(np.uint8))


def load_model(opt):
    import yaml
    if opt.synthetic:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        from oracle import unet as ounet, vqdecoder as ovq, clip as oclip          # synthetic weights only
        spec, vspec = ounet.shipped_spec(), ovq.shipped_vq_spec()
        model = MinimalRETRODiffusion(unet_config={"params": {}}, first_stage_config={"params": {"ddconfig": {}}}, k_nn=opt.k_nn, device=opt.gpu)
        model.load_unet_state_dict(ounet.synth_state_dict(ounet.param_shapes(spec), 1234))
        model.load_first_stage_state_dict(ounet.synth_state_dict(ovq.vq_param_shapes(vspec), 4321))
        rng = np.random.default_rng(7)
        pool = {"embedding": (rng.standard_normal((200_000, 512)) * 0.45).astype(np.float16), "img_id": np.arange(200_000),
                "patch_coords": np.zeros((200_000, 4), np.int64)}
        clip_sd = ounet.synth_state_dict(oclip.clip_param_shapes(oclip.vitb32_spec()), 99)
        retr = ClipImageRetriever(state_dict=clip_sd, ctx=model.ctx)
        model.retriever = DatasetBuilder(data_pool=pool, retriever=retr, ctx=model.ctx)
        model.nn_memory = torch.arange(10_000); model.use_memory = True
        return model
    cfg = yaml.safe_load(open(os.path.join(opt.resume, "config.yaml")))["model"]["params"]
    model = MinimalRETRODiffusion(unet_config=cfg["unet_config"], first_stage_config=cfg["first_stage_config"], k_nn=cfg.get("k_nn", 4),
                                  timesteps=cfg.get("timesteps", 1000), linear_start=cfg["linear_start"], linear_end=cfg["linear_end"],
                                  image_size=cfg["image_size"], channels=cfg["channels"], device=opt.gpu)
    sd = torch.load(os.path.join(opt.resume, "model.ckpt"), map_location="cpu")["state_dict"]
    model.load_state_dict(sd, strict=False)
    clip_sd = torch.load(opt.clip_ckpt, map_location="cpu") if opt.clip_ckpt else None
    retr = ClipImageRetriever(state_dict=clip_sd, ctx=model.ctx) if clip_sd is not None else None
    rp = cfg["retrieval_cfg"]["params"]
    model.retriever = DatasetBuilder(saved_embeddings=rp["saved_embeddings"], k=rp.get("k", 20), retriever=retr, ctx=model.ctx)
    return model


def main():
    opt = parse_args()
    if opt.seed is not None:
        torch.manual_seed(opt.seed); np.random.seed(opt.seed)
    model = load_model(opt)
    logdir = opt.logdir if opt.logdir != "none" else os.path.join(opt.resume if not opt.synthetic else ".", "samples", time.strftime("%Y-%m-%d-%H-%M-%S"))
    os.makedirs(logdir, exist_ok=True)
    start = len(glob.glob(os.path.join(logdir, "*.png")))
    for n in range(opt.n_runs):
        if opt.unconditional or not opt.caption:
            out = model.sample_from_rdata(opt.batch_size, k_nn=opt.k_nn, memsize=opt.top_m, use_weights=opt.use_weights, ddim=True,
                                          ddim_steps=opt.steps, eta=opt.eta, unconditional_guidance_scale=opt.guidance_scale,
                                          unconditional_retro_guidance_label=0.)
            key = "samples_with_sampled_nns"
        else:
            emb = CLIPTextEmbedder(clip=model.retriever.retriever.model)([opt.caption] * opt.batch_size)     # :275-277
            out = model.sample_with_query(query=emb.cpu(), query_embedded=True, k_nn=1 if opt.only_caption else opt.k_nn,
                                          ddim=True, ddim_steps=opt.steps, eta=opt.eta, omit_query=opt.omit_query,
                                          unconditional_guidance_scale=opt.guidance_scale, unconditional_retro_guidance_label=0.)
            key = "query_samples"
        for i, x in enumerate(out[key]):
            custom_to_pil(x).save(os.path.join(logdir, f"{start:06}-{key}-run{n}-sample{i}.png"))
        start += len(out[key])
    print("done ->", logdir)


if __name__ == "__main__":
    main()
