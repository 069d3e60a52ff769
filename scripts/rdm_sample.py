#!/usr/bin/env python3
"""Native counterpart of the reference's scripts/rdm_sample.py.

Every flag of the reference parser (scripts/rdm_sample.py:22-143) is kept with the same spelling, type, default and
meaning: -s/--savepath, --gpu, --model_path, --save_nns, -bs/--batch_size, -n/--n_runs, --seed, --increase_guidance,
--keep_qids, --guidance_scale, --top_m, --k_nn, --steps, -c/--caption, --only_caption, --omit_query, --unconditional,
--use_weights.  The run loops follow :225-315: caption == "" -> sample_from_rdata, else CLIP-encode the caption once and
sample_with_query; `seed_everything(seed)` before EVERY run; DDIM with the sampler's default eta (0.0, deterministic);
k_nn = 1 with --only_caption; omit_query masked by only_caption; `--increase_guidance` adds 1.0 after each run; files are
`{sampling_start}-{key}-run{n}-sample{i}.png` with a `%Y-%m-%d-%H-%M-%S` stamp; uint8 conversion truncates (:203-214).

Deliberate differences:
  * `--seed N` works (the reference dereferences a non-existent `opt.r_runs`, :141);
  * `--gpu -1` (the reference's default, "none") is refused: the native library has no CPU path;
  * `--save_nns` raises NotImplementedError (needs the raw OpenImages JPEGs behind `get_nn_patches`);
  * retrieval is the exact brute-force search on the GPU instead of ScaNN's approximate index.
Additions (not in the reference, all optional): `--clip_ckpt` (CLIP ViT-B/32 state_dict; the reference downloads it),
`--synthetic` (seeded random weights and database: the 6.2 GB checkpoint / 18 GB database are unreachable without a
network), `--gpus N` (batch-sharded multi-GPU sampling, one process per GPU via torch.distributed.run, RCCL all-gather of
the finished images; `--gpu` is then ignored and each rank uses its LOCAL_RANK).
"""
import argparse
import datetime
import os
import random
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    parser.add_argument("-s", "--savepath", type=Path, default="out/rdm", help="Path to savedir")
    parser.add_argument("--gpu", type=int, default=-1, help="On which gpu to sample, -1 for none")
    parser.add_argument("--model_path", type=Path, default="models/rdm/imagenet", help="Path to pretrained model")
    parser.add_argument("--save_nns", default=False, action="store_true", help="Save nearest neighbors")
    parser.add_argument("-bs", "--batch_size", type=int, default=4, help="How many images to generate at once")
    parser.add_argument("-n", "--n_runs", type=int, default=2, help="repeat sampling this number of times")
    parser.add_argument("--seed", type=int, default=None, help="Seed each iteration")
    parser.add_argument("--increase_guidance", default=False, action="store_true", help="Increase cfg after each iteration")
    parser.add_argument("--keep_qids", default=False, action="store_true", help="Keep same queries for each run")
    parser.add_argument("--guidance_scale", type=float, default=2., help="classifier free (transformer) guidance")
    parser.add_argument("--top_m", type=float, default=0.01, help="top-m sampling")
    parser.add_argument("--k_nn", type=int, default=4, help="number of neighbors drawn for sampling")
    parser.add_argument("--steps", type=int, default=100, help="number of ddim steps")
    parser.add_argument("-c", "--caption", type=str, default="", help="Caption used for neighbor retrieval")
    parser.add_argument("--only_caption", default=False, action="store_true", help="use the caption only, no neighbors")
    parser.add_argument("--omit_query", default=False, action="store_true", help="Omit the caption query as nearest neighbor itself")
    parser.add_argument("--unconditional", default=False, action="store_true",
                        help="Sample 'unconditonal' as in the unconditional part of cfg")    # parsed, never read (as in the reference)
    parser.add_argument("--use_weights", default=False, action="store_true",
                        help="Use proposal distribution weights (else sample uniform under top_m)")
    # ---- additions of the native counterpart
    parser.add_argument("--clip_ckpt", type=Path, default=None, help="[native] CLIP ViT-B/32 state_dict (.pt)")
    parser.add_argument("--synthetic", default=False, action="store_true",
                        help="[native] seeded random weights + synthetic database instead of checkpoint files")
    parser.add_argument("--synthetic_db_rows", type=int, default=200_000, help="[native] rows of the --synthetic database")
    parser.add_argument("--gpus", type=int, default=1, help="[native] shard each batch over this many GPUs (RCCL)")
    parser.add_argument("--shard_db", action="store_true", help="[native] with --gpus N: shard the database ROWS over the GPUs instead of "
                        "replicating them (databases beyond one GPU's memory); neighbours are merged in one exchange per search")
    return parser


def parse_args(argv=None) -> argparse.Namespace:
    opt = build_parser().parse_args(argv)
    if opt.top_m > 1.0:
        opt.top_m = int(opt.top_m)          # top_m should be int if a fixed number of images is given (:138-140)
    if opt.seed is not None and (not opt.increase_guidance) and opt.n_runs > 1:
        print("Warning: You will get the same images each run")
    return opt


def seed_everything(seed: int):
    """pytorch_lightning.seed_everything (rdm_sample.py:14, 235, 281): python, numpy and torch (all devices) RNGs."""
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def custom_to_pil(x):
    """rdm_sample.py:203-214: clamp(-1,1) -> (x+1)/2 -> HWC -> (255 x).astype(uint8), i.e. truncation."""
    import torch
    from PIL import Image
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    x = x.detach().cpu()
    x = torch.clamp(x, -1., 1.)
    x = (x + 1.) / 2.
    x = x.permute(1, 2, 0).numpy()
    x = (255 * x).astype(np.uint8)
    x = Image.fromarray(x)
    if not x.mode == "RGB":
        x = x.convert("RGB")
    return x


def save_image(x, savename: str):
    custom_to_pil(x).save(savename)


def load_model(opt: argparse.Namespace):
    """rdm_sample.py:146-187: config.yaml + model.ckpt -> MinimalRETRODiffusion on the selected GPU (EMA weights resident in
    HBM), retriever = DatasetBuilder over the saved embeddings with the CLIP towers on the same device."""
    import torch
    import rdm_amd  # noqa: F401
    from rdm_amd import _lib, synthetic
    from rdm_amd.data.retrieval_dataset.dsetbuilder import DatasetBuilder
    from rdm_amd.models.diffusion.ddpm import MinimalRETRODiffusion
    from rdm_amd.modules.retrievers import ClipImageRetriever

    if opt.save_nns:
        raise NotImplementedError("--save_nns needs the raw OpenImages patches behind get_nn_patches (out of scope, SURVEY.md §2 #8)")
    if opt.gpu < 0:
        raise SystemExit("rdm_sample.py (native): --gpu must name a HIP device; the native library has no CPU path")
    if opt.synthetic:
        model = MinimalRETRODiffusion(unet_config={"params": {}}, first_stage_config={"params": {"ddconfig": {}}}, k_nn=4, device=opt.gpu)
        model.load_unet_state_dict(synthetic.unet_state_dict(model.unet_cfg))
        model.load_first_stage_state_dict(synthetic.vq_state_dict(model.vq_cfg))
        n = opt.synthetic_db_rows
        pool = {"embedding": synthetic.clip_like_rows(n), "img_id": np.arange(n), "patch_coords": np.zeros((n, 4), np.int64)}
        retr = ClipImageRetriever(state_dict=synthetic.clip_state_dict(_lib.make_clip_cfg()), ctx=model.ctx)
        model.retriever = DatasetBuilder(data_pool=pool, retriever=retr, ctx=model.ctx)
        model.nn_memory = torch.arange(min(10_000, n)); model.use_memory = True
        return model.eval()
    import yaml
    model_dir = opt.model_path
    config_path, ckpt_path = model_dir / "config.yaml", model_dir / "model.ckpt"
    assert config_path.is_file(), f"Did not found config at {config_path}"
    assert ckpt_path.is_file(), f"Did not found ckpt at {ckpt_path}"
    cfg = yaml.safe_load(open(config_path))["model"]["params"]
    pl_sd = torch.load(ckpt_path, map_location="cpu")
    rp = dict(cfg["retrieval_cfg"]["params"])
    nn_memory = cfg.get("nn_memory")
    if isinstance(nn_memory, str) and os.path.isfile(nn_memory):          # ddpm.py:168-176: pickled {'nn_memory', 'id_count'}
        import pickle
        with open(nn_memory, "rb") as f:
            mem = pickle.load(f)
        nn_memory, id_count = mem["nn_memory"], mem.get("id_count")
    else:
        nn_memory, id_count = None, None
    model = MinimalRETRODiffusion(unet_config=cfg["unet_config"], first_stage_config=cfg["first_stage_config"], k_nn=cfg.get("k_nn", 4),
                                  timesteps=cfg.get("timesteps", 1000), linear_start=cfg["linear_start"], linear_end=cfg["linear_end"],
                                  image_size=cfg["image_size"], channels=cfg["channels"], log_every_t=cfg.get("log_every_t", 200),
                                  scale_factor=cfg.get("scale_factor", 1.0), nn_memory=nn_memory, id_count=id_count, device=opt.gpu)
    m, u = model.load_state_dict(pl_sd["state_dict"], strict=False)
    print("Loaded model.")
    clip_sd = torch.load(opt.clip_ckpt, map_location="cpu") if opt.clip_ckpt else None
    if clip_sd is None:
        raise SystemExit("rdm_sample.py (native): pass --clip_ckpt <ViT-B/32 state_dict>; the reference downloads it (no network here)")
    retr = ClipImageRetriever(model=rp.get("retriever_config", {}).get("params", {}).get("model", "ViT-B/32"), state_dict=clip_sd, ctx=model.ctx)
    model.retriever = DatasetBuilder(saved_embeddings=rp["saved_embeddings"], k=rp.get("k", 20), retriever=retr, ctx=model.ctx,
                                     load_patch_dataset=False)
    return model.eval()


def _save_logs(logs, keys, opt, sampling_start, n):
    for key in logs:
        if keys is not None and key not in keys:
            continue
        for bi, be in enumerate(logs[key]):
            savename = os.path.join(opt.savepath, f'{sampling_start}-{key}-run{n}-sample{bi}.png')
            if be.ndim == 3:
                save_image(be, savename)
            else:
                raise NotImplementedError("image grids (batched_nns) belong to --save_nns")


def sample_unconditional(model, opt: argparse.Namespace, is_writer=True):
    """rdm_sample.py:225-266."""
    qids = model.get_qids(opt.top_m, opt.batch_size, use_weights=opt.use_weights) if opt.keep_qids else None
    sampling_start = datetime.datetime.now().strftime("%Y-%m-%d-%H-%M-%S")
    for n in range(opt.n_runs):
        if opt.seed is not None:
            seed_everything(opt.seed)
        print("Sampling query and neighbors (wait for the sampling to start)")
        logs = model.sample_from_rdata(opt.batch_size, qids=qids, k_nn=opt.k_nn, return_nns=opt.save_nns, use_weights=opt.use_weights,
                                       memsize=opt.top_m, unconditional_guidance_scale=opt.guidance_scale, ddim_steps=opt.steps,
                                       ddim=True, unconditional_retro_guidance_label=0.)
        if is_writer:
            _save_logs(logs, ["samples_with_sampled_nns", "batched_nns"], opt, sampling_start, n)
        if opt.increase_guidance:
            opt.guidance_scale += 1.0
            print(f"New guidance scale: {opt.guidance_scale}")
    print("Done")
    return sampling_start


def sample_conditional(model, opt: argparse.Namespace, is_writer=True):
    """rdm_sample.py:270-315."""
    from rdm_amd.modules.custom_clip.tokenizer import tokenize
    import torch
    sampling_start = datetime.datetime.now().strftime("%Y-%m-%d-%H-%M-%S")
    tokenized = torch.from_numpy(tokenize([opt.caption] * opt.batch_size))
    clip = model.retriever.retriever.model
    query_embeddings = clip.encode_text(tokenized).cpu()
    del tokenized
    for n in range(opt.n_runs):
        if opt.seed is not None:
            seed_everything(opt.seed)
        print("Sampling query and neighbors (wait for the sampling to start)")
        logs = model.sample_with_query(query=query_embeddings, query_embedded=True, k_nn=opt.k_nn if not opt.only_caption else 1,
                                       return_nns=opt.save_nns and not opt.only_caption, visualize_nns=opt.save_nns and not opt.only_caption,
                                       use_weights=opt.use_weights, unconditional_guidance_scale=opt.guidance_scale, ddim_steps=opt.steps,
                                       ddim=True, unconditional_retro_guidance_label=0., omit_query=opt.omit_query and not opt.only_caption)
        print(f"Run {n + 1}/{opt.n_runs}")
        if is_writer:
            _save_logs(logs, None, opt, sampling_start, n)
        if opt.increase_guidance:
            opt.guidance_scale += 1.0
            print(f"New guidance scale: {opt.guidance_scale}")
    print("Done")
    return sampling_start


def main(argv=None):
    opt = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if opt.gpus > 1 and world == 1:
        # one process per GPU; launched BEFORE anything in this process touches the GPU (importing the package / torch does not);
        # refused under a profiler, where the preloaded tool already has (rdm_amd.parallel.safe_self_launch)
        sys.path.insert(0, ROOT)
        import rdm_amd  # noqa: F401
        from rdm_amd import parallel
        sys.exit(parallel.safe_self_launch(__file__, opt.gpus, sys.argv[1:] if argv is None else list(argv)))
    sys.path.insert(0, ROOT)
    opt.savepath.mkdir(parents=True, exist_ok=True)
    is_writer = True
    if world > 1:
        from rdm_amd import parallel
        rank, local = parallel.init_distributed()
        opt.gpu = local
        is_writer = rank == 0
    model = load_model(opt)
    if world > 1:
        model.set_distributed(True, shard_db=opt.shard_db)       # noise streams: f(shared seed drawn after seed_everything, global row)
    if opt.caption == "":
        sample_unconditional(model, opt, is_writer)
    else:
        sample_conditional(model, opt, is_writer)
    if world > 1:
        from rdm_amd import parallel
        parallel.shutdown()


if __name__ == "__main__":
    main()
