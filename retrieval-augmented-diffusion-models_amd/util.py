"""rdm/util.py counterparts used on the sampling path."""
import numpy as np
import torch


def ischannellastimage(x) -> bool:
    """rdm/util.py:17-20."""
    return hasattr(x, "shape") and len(x.shape) == 4 and x.shape[-1] in (1, 3)


def convert_nn_tree(nn_tree):
    """rdm/util.py:23-30: ScaNN ids uint32 -> int32 (recursively over dict / array)."""
    if isinstance(nn_tree, dict):
        return {k: convert_nn_tree(v) for k, v in nn_tree.items()}
    if isinstance(nn_tree, np.ndarray) and nn_tree.dtype == np.uint32:
        return nn_tree.astype(np.int32)
    return nn_tree


def instantiate_from_config(config):
    """ldm.util.instantiate_from_config for `target:` / `params:` trees, with rdm.* / ldm.* targets that have a
    native counterpart redirected to rdm_amd.*"""
    import importlib
    target = config["target"]
    redirect = {
        "rdm.models.diffusion.ddpm.MinimalRETRODiffusion": "rdm_amd.models.diffusion.ddpm.MinimalRETRODiffusion",
        "rdm.data.retrieval_dataset.dsetbuilder.DatasetBuilder": "rdm_amd.data.retrieval_dataset.dsetbuilder.DatasetBuilder",
        "rdm.modules.retrievers.ClipImageRetriever": "rdm_amd.modules.retrievers.ClipImageRetriever",
        "rdm.modules.retrievers.CLIPTextEmbedder": "rdm_amd.modules.retrievers.CLIPTextEmbedder",
        "rdm.models.autoregression.transformer.LatentImageRETRO": "rdm_amd.models.autoregression.transformer.LatentImageRETRO",
    }
    target = redirect.get(target, target)
    mod, cls = target.rsplit(".", 1)
    return getattr(importlib.import_module(mod), cls)(**config.get("params", dict()))
