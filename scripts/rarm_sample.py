#!/usr/bin/env python3
"""Native counterpart of the reference's scripts/rarm_sample.py (RARM: retrieval-augmented autoregressive sampling).

Same flags, defaults and run loop as scripts/rarm_sample.py:100-293: -s/--savepath (out/rarm), --gpu, --model_path
(models/rarm/imagenet/dogs), --save_nns, -bs, -n, --seed, --increase_guidance, --keep_qids, --guidance_scale (1.0), --top_k (256),
--temperature (1.0), --top_m (0.01), --k_nn (4), -c/--caption, --only_caption, --unconditional, --use_weights; 256 tokens
(f16 first stage), `seed_everything` before every run, files `{start}-{key}-run{n}-sample{i}.png`.
Deliberate differences: as scripts/rdm_sample.py (no CPU path, --save_nns unsupported, --seed works).  Additions:
--clip_ckpt, --synthetic.
"""
import argparse
import datetime
import os
import sys
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, ROOT)
from rdm_sample import custom_to_pil, save_image, seed_everything  # noqa: E402,F401


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    parser.add_argument("-s", "--savepath", type=Path, default="out/rarm", help="Path to savedir")
    parser.add_argument("--gpu", type=int, default=-1, help="On which gpu to sample, -1 for none")
    parser.add_argument("--model_path", type=Path, default="models/rarm/imagenet/dogs", help="Path to pretrained model")
    parser.add_argument("--save_nns", default=False, action="store_true", help="Save nearest neighbors")
    parser.add_argument("-bs", "--batch_size", type=int, default=4, help="How many images to generate at once")
    parser.add_argument("-n", "--n_runs", type=int, default=2, help="repeat sampling this number of times")
    parser.add_argument("--seed", type=int, default=None, help="Seed each iteration")
    parser.add_argument("--increase_guidance", default=False, action="store_true", help="Increase cfg after each iteration")
    parser.add_argument("--keep_qids", default=False, action="store_true", help="Keep same queries for each run")
    parser.add_argument("--guidance_scale", type=float, default=1., help="classifier free (transformer) guidance")
    parser.add_argument("--top_k", type=int, default=256, help="top-k sampling")
    parser.add_argument("--temperature", type=float, default=1., help="temperature sampling")
    parser.add_argument("--top_m", type=float, default=0.01, help="top-m sampling")
    parser.add_argument("--k_nn", type=int, default=4, help="number of neighbors drawn for sampling")
    parser.add_argument("-c", "--caption", type=str, default="", help="Caption used for neighbor retrieval")
    parser.add_argument("--only_caption", default=False, action="store_true", help="use the caption only, no neighbors")
    parser.add_argument("--unconditional", default=False, action="store_true",
                        help="Sample 'unconditonal' as in the unconditional part of cfg")
    parser.add_argument("--use_weights", default=False, action="store_true",
                        help="Use proposal distribution weights (else sample uniform under top_m)")
    parser.add_argument("--clip_ckpt", type=Path, default=None, help="[native] CLIP ViT-B/32 state_dict (.pt)")
    parser.add_argument("--synthetic", default=False, action="store_true", help="[native] seeded random weights + synthetic database")
    parser.add_argument("--synthetic_db_rows", type=int, default=200_000, help="[native] rows of the --synthetic database")
    return parser


def parse_args(argv=None) -> argparse.Namespace:
    opt = build_parser().parse_args(argv)
    if opt.top_m > 1.0:
        opt.top_m = int(opt.top_m)
    if opt.seed is not None and (not opt.increase_guidance) and opt.n_runs > 1:
        print("Warning: You will get the same images each run")
    return opt


def load_model(opt):
    """rarm_sample.py:25-70."""
    import torch
    import rdm_amd  # noqa: F401
    from rdm_amd import _lib, synthetic
    from rdm_amd.data.retrieval_dataset.dsetbuilder import DatasetBuilder
    from rdm_amd.models.autoregression.transformer import LatentImageRETRO
    from rdm_amd.modules.retrievers import ClipImageRetriever
    if opt.save_nns:
        raise NotImplementedError("--save_nns needs the raw OpenImages patches behind get_nn_patches (out of scope, SURVEY.md §2 #8)")
    if opt.gpu < 0:
        raise SystemExit("rarm_sample.py (native): --gpu must name a HIP device; the native library has no CPU path")
    if opt.synthetic:
        model = LatentImageRETRO(transformer_config={"params": {}}, first_stage_config={"params": {"ddconfig": {}}}, k_nn=opt.k_nn, device=opt.gpu,
                                 nn_memory=np.arange(min(10_000, opt.synthetic_db_rows)))
        model.load_transformer_state_dict(synthetic.rarm_state_dict(model.rarm_cfg))
        model.load_first_stage_state_dict(synthetic.vq_state_dict(model.vq_cfg, synthetic.VQGAN_SEED))
        n = opt.synthetic_db_rows
        pool = {"embedding": synthetic.clip_like_rows(n), "img_id": np.arange(n), "patch_coords": np.zeros((n, 4), np.int64)}
        retr = ClipImageRetriever(state_dict=synthetic.clip_state_dict(_lib.make_clip_cfg()), ctx=model.ctx)
        model.retriever = DatasetBuilder(data_pool=pool, retriever=retr, ctx=model.ctx)
        return model.eval()
    import yaml
    model_dir = opt.model_path
    config_path, ckpt_path = model_dir / "config.yaml", model_dir / "model.ckpt"
    assert config_path.is_file(), f"Did not found config at {config_path}"
    assert ckpt_path.is_file(), f"Did not found ckpt at {ckpt_path}"
    cfg = yaml.safe_load(open(config_path))["model"]["params"]
    pl_sd = torch.load(ckpt_path, map_location="cpu")
    nn_memory, id_count = None, None
    if isinstance(cfg.get("nn_memory"), str) and os.path.isfile(cfg["nn_memory"]):
        import pickle
        with open(cfg["nn_memory"], "rb") as f:
            mem = pickle.load(f)
        nn_memory, id_count = mem["nn_memory"], mem.get("id_count")
    model = LatentImageRETRO(transformer_config=cfg["transformer_config"], first_stage_config=cfg["first_stage_config"],
                             mask_token=cfg.get("mask_token", 16384), sos_token=cfg.get("sos_token", 16385), nn_memory=nn_memory,
                             id_count=id_count, device=opt.gpu)
    model.load_state_dict(pl_sd["state_dict"])
    print("Loaded model.")
    if opt.clip_ckpt is None:
        raise SystemExit("rarm_sample.py (native): pass --clip_ckpt <ViT-B/32 state_dict>; the reference downloads it (no network here)")
    rp = dict(cfg["retrieval_cfg"]["params"])
    retr = ClipImageRetriever(state_dict=torch.load(opt.clip_ckpt, map_location="cpu"), ctx=model.ctx)
    model.retriever = DatasetBuilder(saved_embeddings=rp["saved_embeddings"], k=rp.get("k", 20), retriever=retr, ctx=model.ctx)
    return model.eval()


def sample(model, opt):
    """rarm_sample.py:225-293."""
    import torch
    from rdm_amd.modules.custom_clip.tokenizer import tokenize
    qids = None
    sampling_start = datetime.datetime.now().strftime("%Y-%m-%d-%H-%M-%S")
    query_embeddings = None
    nn_embeddings = None
    if opt.caption != "":
        tokenized = torch.from_numpy(tokenize([opt.caption] * opt.batch_size))
        query_embeddings = model.retriever.retriever.model.encode_text(tokenized).cpu()
    if opt.only_caption:
        assert opt.caption != "", "Need a caption"
        nn_embeddings = query_embeddings.unsqueeze(1).to(model.device).float()
    elif opt.unconditional:
        nn_embeddings = torch.zeros((opt.batch_size, 1, 512), dtype=torch.float, device=model.device)
    for n in range(opt.n_runs):
        if opt.seed is not None:
            seed_everything(opt.seed)
        print("Sampling query and neighbors (wait for the sampling to start)")
        logs = model.sample_from_rdata(opt.batch_size, qids=qids, query_embeddings=query_embeddings, nn_embeddings=nn_embeddings,
                                       k_nn=opt.k_nn, return_nns=opt.save_nns, use_weights=opt.use_weights, memsize=opt.top_m,
                                       top_k=opt.top_k, temperature=opt.temperature, guidance_scale=opt.guidance_scale)
        if opt.keep_qids:
            assert "qids" in logs
            qids = logs["qids"]
        print(f"Run {n + 1}/{opt.n_runs}")
        for key in logs:
            if key in ["samples_with_sampled_nns", "batched_nns"]:
                for bi, be in enumerate(logs[key]):
                    savename = os.path.join(opt.savepath, f'{sampling_start}-{key}-run{n}-sample{bi}.png')
                    if be.ndim == 3:
                        save_image(be, savename)
        if opt.increase_guidance:
            opt.guidance_scale += 1.0
            print(f"New guidance scale: {opt.guidance_scale}")
    print("Done")
    return sampling_start


if __name__ == "__main__":
    opt = parse_args()
    opt.savepath.mkdir(parents=True, exist_ok=True)
    model = load_model(opt)
    sample(model, opt)
