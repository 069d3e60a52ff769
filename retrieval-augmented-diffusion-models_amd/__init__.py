"""rdm_amd — MI355X-native retrieval-augmented diffusion sampling path.

Host-side mirror of the reference's Python surface for the sampling path (same module layout under
`rdm_amd.` as under `rdm.` in CompVis/retrieval-augmented-diffusion-models):

    rdm_amd.models.diffusion.ddim.DDIMSampler                  <- rdm/models/diffusion/ddim.py
    rdm_amd.models.diffusion.ddpm.MinimalRETRODiffusion        <- rdm/models/diffusion/ddpm.py (sampling methods)
    rdm_amd.modules.retrievers.{ClipImageRetriever,CLIPTextEmbedder}   <- rdm/modules/retrievers.py
    rdm_amd.data.retrieval_dataset.dsetbuilder.DatasetBuilder  <- rdm/data/retrieval_dataset/dsetbuilder.py

All compute goes through the C-ABI library `librdm_hip.so` (include/rdm_hip.h).  There is no CPU
fallback: importing `rdm_amd._lib` without the built library, or creating a context without a HIP
device, raises.
"""
__version__ = "0.1.0"
