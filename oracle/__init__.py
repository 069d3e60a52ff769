"""CPU oracle for the retrieval-augmented diffusion sampling path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product path (``rdm_amd`` -> ``librdm_hip.so``) never routes through here and
fails loudly when the HIP library is missing.

What it is: a plain fp32 PyTorch / numpy restatement of the reference's
sampling arithmetic (UNet, DDIM/DDPM loop, VQ-f4 decode, CLIP towers, exact
top-k retrieval).  Each function cites the reference file:line it follows.

Pinning status (see DESIGN.md "Oracle"):
  * PINNED by golden vectors generated in the build container from the
    reference's own in-tree classes (``tools/gen_golden.py``):
      - UNetModel wiring, SpatialTransformer, BasicTransformerBlock,
        CrossAttention          (rdm/modules/diffusionmodules/openaimodel.py,
                                 rdm/modules/attention.py)
      - CLIP text / image towers (rdm/modules/custom_clip/model.py)
  * PINNED by known-answer constants: diffusion
    schedule, DDIM timesteps / alphas / sigmas, one DDIM step.
  * PARITY UNPINNED: everything whose source lives in the un-vendored
    third-party packages ``ldm`` (ResBlock, Down/Upsample, GroupNorm32,
    timestep_embedding, GEGLU FeedForward, LatentDiffusion.p_sample*, the
    VQ decoder), ``taming`` (VectorQuantizer2) and ``scann`` (approximate
    kNN; our oracle is the exact search ScaNN approximates).  These are
    restated from their published algorithms.
"""
