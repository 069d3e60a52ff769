"""ctypes binding of librdm_hip.so (include/rdm_hip.h) + a thin torch-tensor convenience layer.

There is deliberately NO fallback: if the shared library is missing this module raises at import,
and `Context()` raises when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RDM_HIP_LIB") or os.path.join(_HERE, "librdm_hip.so")   # env override: A/B builds (dev)

RDM_MAX_LEVELS = 8
ACT_NONE, ACT_GEGLU, ACT_QUICKGELU, ACT_SILU = 0, 1, 2, 3
PROF_CONV3X3, PROF_LINEAR, PROF_KNN, PROF_ATTENTION, PROF_GROUPNORM, PROF_LAYERNORM, PROF_UPSCONV = range(7)


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("model_channels", C.c_int),
                ("num_res_blocks", C.c_int), ("n_attention_resolutions", C.c_int),
                ("attention_resolutions", C.c_int * RDM_MAX_LEVELS), ("n_channel_mult", C.c_int),
                ("channel_mult", C.c_int * RDM_MAX_LEVELS), ("num_head_channels", C.c_int), ("context_dim", C.c_int)]


class VqCfg(C.Structure):
    _fields_ = [("embed_dim", C.c_int), ("n_embed", C.c_int), ("z_channels", C.c_int), ("ch", C.c_int),
                ("n_ch_mult", C.c_int), ("ch_mult", C.c_int * RDM_MAX_LEVELS), ("num_res_blocks", C.c_int),
                ("out_ch", C.c_int), ("resolution", C.c_int), ("mid_attn", C.c_int), ("kl", C.c_int),
                ("n_attn_resolutions", C.c_int), ("attn_resolutions", C.c_int * RDM_MAX_LEVELS)]


class RarmCfg(C.Structure):
    _fields_ = [("vocab_in", C.c_int), ("vocab_out", C.c_int), ("n_heads", C.c_int), ("d_head", C.c_int), ("depth", C.c_int),
                ("context_dim", C.c_int), ("sequence_length", C.c_int)]


class RarmSampleArgs(C.Structure):
    _fields_ = [("batch", C.c_int), ("k", C.c_int), ("cond_len", C.c_int), ("steps", C.c_int), ("temperature", C.c_float),
                ("top_k", C.c_int), ("guidance_scale", C.c_float)]


class ClipCfg(C.Structure):
    _fields_ = [("embed_dim", C.c_int), ("image_resolution", C.c_int), ("vision_layers", C.c_int),
                ("vision_width", C.c_int), ("vision_patch_size", C.c_int), ("context_length", C.c_int),
                ("vocab_size", C.c_int), ("transformer_width", C.c_int), ("transformer_heads", C.c_int),
                ("transformer_layers", C.c_int)]


class DdimArgs(C.Structure):
    _fields_ = [("S", C.c_int), ("batch", C.c_int), ("k", C.c_int), ("channels", C.c_int), ("height", C.c_int),
                ("width", C.c_int), ("eta", C.c_float), ("temperature", C.c_float),
                ("unconditional_guidance_scale", C.c_float), ("log_every_t", C.c_int), ("T", C.c_int),
                ("alphas_cumprod", C.POINTER(C.c_float))]


class DdpmArgs(C.Structure):
    _fields_ = [("timesteps", C.c_int), ("batch", C.c_int), ("k", C.c_int), ("channels", C.c_int), ("height", C.c_int),
                ("width", C.c_int), ("clip_denoised", C.c_int), ("temperature", C.c_float), ("T", C.c_int),
                ("sqrt_recip_alphas_cumprod", C.POINTER(C.c_float)), ("sqrt_recipm1_alphas_cumprod", C.POINTER(C.c_float)),
                ("posterior_mean_coef1", C.POINTER(C.c_float)), ("posterior_mean_coef2", C.POINTER(C.c_float)),
                ("posterior_log_variance_clipped", C.POINTER(C.c_float))]


_P = C.c_void_p
# name -> (restype, argtypes); exactly the symbols include/rdm_hip.h declares
SIGNATURES = {
    "rdm_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "rdm_ctx_destroy": (None, [_P]),
    "rdm_last_error": (C.c_char_p, [_P]),
    "rdm_set_stream": (C.c_int, [_P, _P]),
    "rdm_version": (C.c_char_p, []),
    "rdm_unet_manifest": (C.c_longlong, [C.POINTER(UNetCfg), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rdm_vq_manifest": (C.c_longlong, [C.POINTER(VqCfg), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rdm_clip_manifest": (C.c_longlong, [C.POINTER(ClipCfg), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rdm_load_unet": (C.c_int, [_P, C.POINTER(UNetCfg), _P, C.c_size_t]),
    "rdm_load_vq": (C.c_int, [_P, C.POINTER(VqCfg), _P, C.c_size_t]),
    "rdm_load_clip": (C.c_int, [_P, C.POINTER(ClipCfg), _P, C.c_size_t]),
    "rdm_rarm_manifest": (C.c_longlong, [C.POINTER(RarmCfg), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rdm_load_rarm": (C.c_int, [_P, C.POINTER(RarmCfg), _P, C.c_size_t]),
    "rdm_rarm_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, _P]),
    "rdm_rarm_sample": (C.c_int, [_P, C.POINTER(RarmSampleArgs), _P, _P, _P, _P]),
    "rdm_vq_decode_indices": (C.c_int, [_P, _P, C.c_int, _P]),
    "rdm_unet_forward": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "rdm_ddim_num_intermediates": (C.c_int, [C.c_int, C.c_int]),
    "rdm_ddim_sample": (C.c_int, [_P, C.POINTER(DdimArgs), _P, _P, _P, _P, _P, _P, _P]),
    "rdm_ddpm_sample": (C.c_int, [_P, C.POINTER(DdpmArgs), _P, _P, _P, _P]),
    "rdm_vq_decode": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "rdm_release_scratch": (C.c_int, [_P]),
    "rdm_vq_quantize": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "rdm_to_uint8": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "rdm_clip_encode_text": (C.c_int, [_P, _P, C.c_int, _P]),
    "rdm_clip_encode_image": (C.c_int, [_P, _P, C.c_int, _P]),
    "rdm_clip_preprocess": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "rdm_clip_encode_image_raw": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "rdm_db_load": (C.c_int, [_P, _P, C.c_longlong, C.c_int, C.c_int, C.c_int]),
    "rdm_db_size": (C.c_longlong, [_P]),
    "rdm_knn": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "rdm_knn_f64": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "rdm_knn_last_fallback": (C.c_int, [_P]),
    "rdm_db_gather": (C.c_int, [_P, _P, C.c_longlong, _P]),
    "rdm_set_deterministic": (C.c_int, [_P, C.c_int]),
    "rdm_get_deterministic": (C.c_int, [_P]),
    "rdm_prof_enable": (C.c_int, [_P, C.c_int]),
    "rdm_prof_collect": (C.c_int, [_P, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rdm_prof_reset": (C.c_int, [_P]),
    "rdm_prof_dump": (C.c_int, [_P, C.c_char_p]),
    "rdm_debug_tap": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int]),
    "rdm_op_ffn_fused": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int]),
    "rdm_debug_counter": (C.c_int, [_P, C.c_int, C.POINTER(C.c_ulonglong)]),
    "rdm_calib_probe": (C.c_int, [_P, _P, C.c_size_t, C.c_double, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rdm_comm_unique_id": (C.c_int, [_P, _P]),
    "rdm_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "rdm_comm_all_gather": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "rdm_comm_all_reduce_f32": (C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    "rdm_comm_destroy": (C.c_int, [_P]),
    "rdm_vqenc_manifest": (C.c_longlong, [_P, _P, C.c_size_t, _P]),
    "rdm_load_vqenc": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "rdm_vq_encode": (C.c_int, [_P, _P, C.c_int, _P]),
    "rdm_op_q_sample": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rdm_op_mse_loss": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rdm_op_where_rows": (C.c_int, [_P, _P, _P, _P, _P, C.c_longlong, C.c_longlong]),
    "rdm_op_timestep_embedding": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "rdm_op_colsum_samples": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "rdm_op_expand2": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rdm_op_linear": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "rdm_op_conv3x3": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int]),
    "rdm_op_rarm_sampler": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _P, _P]),
    "rdm_op_conv3x3_dgrad": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rdm_op_conv3x3_wgrad": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rdm_op_groupnorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P, _P, _P]),
    "rdm_op_layernorm_bwd": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P, _P, _P]),
    "rdm_op_groupnorm_bwd_add": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P, _P, _P, _P]),
    "rdm_op_layernorm_bwd_add": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P]),
    "rdm_op_colsum": (C.c_int, [_P, _P, _P, C.c_longlong, C.c_int]),
    "rdm_op_transpose": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "rdm_op_add": (C.c_int, [_P, _P, _P, _P, C.c_longlong]),
    "rdm_op_geglu": (C.c_int, [_P, _P, _P, _P, C.c_longlong, C.c_int]),
    "rdm_op_linear_ln": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "rdm_op_linear_wgrad": (C.c_int, [_P, _P, _P, _P, C.c_longlong, C.c_int, C.c_int]),
    "rdm_op_ema": (C.c_int, [_P, _P, _P, C.c_longlong, C.c_float]),
    "rdm_op_silu": (C.c_int, [_P, _P, _P, _P, C.c_longlong]),
    "rdm_op_sumpool2": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rdm_op_adamw": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_longlong, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]),
    "rdm_op_attention_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "rdm_op_bmm": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "rdm_op_heads": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rdm_op_transpose_batched": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "rdm_op_softmax": (C.c_int, [_P, _P, _P, C.c_longlong, C.c_int, C.c_int]),
    "rdm_op_softmax_bwd": (C.c_int, [_P, _P, _P, _P, C.c_longlong, C.c_int]),
    "rdm_op_groupnorm": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_float, C.c_int, _P]),
    "rdm_op_layernorm": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_float, _P]),
    "rdm_op_self_attention": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "rdm_op_self_attention_qkv": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "rdm_op_head_conv": (C.c_int, [_P, _P, _P, _P, C.c_float, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "rdm_op_linear_rowvec": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int]),
    "rdm_op_xattn_fused_ln3": (C.c_int, [_P, _P, _P, _P, C.c_float, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "rdm_op_xattn_fused": (C.c_int, [_P, _P, _P, _P, C.c_float, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "rdm_op_small_attention": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_float, _P, C.c_int]),
    "rdm_op_adamw_multi": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]),
    "rdm_op_ema_multi": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_float]),
    "rdm_op_small_attention_bwd": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P]),
}


def load_library(path: str = LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"librdm_hip.so not found at {path}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C retrieval-augmented-diffusion-models_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib


lib = load_library()


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous(), "tensor must be contiguous"
        return C.c_void_p(t.data_ptr())
    raise TypeError(type(t))


def make_unet_cfg(in_channels=3, out_channels=3, model_channels=192, num_res_blocks=2, attention_resolutions=(8, 4, 2),
                  channel_mult=(1, 2, 3, 5), num_head_channels=32, context_dim=512, **other) -> UNetCfg:
    """UNetModel keyword arguments (openaimodel.py:228-262) -> rdm_unet_cfg.  Arguments that do not change the sampling graph are
    accepted and dropped; ones that select a graph the library does not build raise instead of being silently ignored."""
    inert = {"image_size", "use_checkpoint", "use_fp16", "dropout", "legacy", "use_new_attention_order", "num_heads", "num_heads_upsample"}
    required = {"conv_resample": True, "dims": 2, "num_classes": None, "use_scale_shift_norm": False, "resblock_updown": False,
                "use_spatial_transformer": True, "transformer_depth": 1, "n_embed": None}
    for k_, v_ in other.items():
        if k_ in inert:
            continue
        if k_ not in required:
            raise TypeError(f"make_unet_cfg: unknown UNetModel argument {k_!r}")
        if v_ != required[k_]:
            raise NotImplementedError(f"UNetModel({k_}={v_!r}): the native graph is built for {k_}={required[k_]!r} (every shipped RDM config)")
    c = UNetCfg()
    c.in_channels, c.out_channels, c.model_channels, c.num_res_blocks = in_channels, out_channels, model_channels, num_res_blocks
    c.n_attention_resolutions = len(attention_resolutions)
    for i, v in enumerate(attention_resolutions):
        c.attention_resolutions[i] = int(v)
    c.n_channel_mult = len(channel_mult)
    for i, v in enumerate(channel_mult):
        c.channel_mult[i] = int(v)
    c.num_head_channels, c.context_dim = num_head_channels, context_dim
    return c


def make_vq_cfg(embed_dim=3, n_embed=8192, z_channels=3, ch=128, ch_mult=(1, 2, 4), num_res_blocks=2, out_ch=3,
                resolution=256, mid_attn=True, kl=False, attn_resolutions=(), **_ignored) -> VqCfg:
    c = VqCfg()
    c.embed_dim, c.n_embed, c.z_channels, c.ch = embed_dim, n_embed, z_channels, ch
    c.n_ch_mult = len(ch_mult)
    for i, v in enumerate(ch_mult):
        c.ch_mult[i] = int(v)
    c.num_res_blocks, c.out_ch, c.resolution, c.mid_attn, c.kl = num_res_blocks, out_ch, resolution, int(mid_attn), int(kl)
    c.n_attn_resolutions = len(attn_resolutions)
    for i, v in enumerate(attn_resolutions):
        c.attn_resolutions[i] = int(v)
    return c


def make_vqgan_f16_cfg(**kw) -> VqCfg:
    """taming VQModel of the RARM models (models/rarm/imagenet/dogs/config.yaml:28-51)."""
    d = dict(embed_dim=256, n_embed=16384, z_channels=256, ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, out_ch=3, resolution=256,
             attn_resolutions=(16,))
    d.update(kw)
    return make_vq_cfg(**d)


def make_rarm_cfg(in_channels=16386, out_channels=16384, n_heads=12, d_head=64, depth=18, context_dim=512, sequence_length=256,
                  **other) -> RarmCfg:
    """RetrievalPatchTransformer params (models/rarm/imagenet/dogs/config.yaml:14-27; rdm/modules/attention.py:206-220).  The native
    graph is the shipped one: token input (continuous=False), learned positions, causal self-attention + cross-attention to the
    neighbours, no outer residual; anything else raises instead of being silently ignored."""
    required = {"positional_encodings": True, "cross_attend": True, "causal": True, "continuous": False, "residual": False}
    for k_, v_ in other.items():
        if k_ in ("dropout", "checkpoint"):
            continue
        if k_ not in required:
            raise TypeError(f"make_rarm_cfg: unknown RetrievalPatchTransformer argument {k_!r}")
        if v_ != required[k_]:
            raise NotImplementedError(f"RetrievalPatchTransformer({k_}={v_!r}): the native graph is built for {k_}={required[k_]!r}")
    return RarmCfg(vocab_in=in_channels, vocab_out=out_channels, n_heads=n_heads, d_head=d_head, depth=depth, context_dim=context_dim,
                   sequence_length=sequence_length)


def make_clip_cfg(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
                  context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8,
                  transformer_layers=12, **_ignored) -> ClipCfg:
    c = ClipCfg()
    (c.embed_dim, c.image_resolution, c.vision_layers, c.vision_width, c.vision_patch_size, c.context_length,
     c.vocab_size, c.transformer_width, c.transformer_heads, c.transformer_layers) = (
        embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length, vocab_size,
        transformer_width, transformer_heads, transformer_layers)
    return c


def manifest(kind: str, cfg):
    """-> (list of (offset, nbytes, kind, [src...]), blob_bytes)"""
    fn = {"unet": lib.rdm_unet_manifest, "vq": lib.rdm_vq_manifest, "vqenc": lib.rdm_vqenc_manifest, "clip": lib.rdm_clip_manifest,
          "rarm": lib.rdm_rarm_manifest}[kind]
    blob = C.c_size_t(0)
    n = fn(C.byref(cfg), None, 0, C.byref(blob))
    if n < 0:
        raise ValueError(f"unsupported {kind} config")
    buf = C.create_string_buffer(int(n) + 1)
    fn(C.byref(cfg), buf, int(n) + 1, C.byref(blob))
    out = []
    for line in buf.value.decode().splitlines():
        off, nb, kd, srcs = line.split(" ")
        out.append((int(off), int(nb), kd, srcs.split(",")))
    return out, int(blob.value)


class RdmError(RuntimeError):
    pass


class Context:
    """One library context per HIP device (include/rdm_hip.h: rdm_ctx)."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise RdmError("rdm_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        h = _P()
        rc = lib.rdm_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise RdmError(f"rdm_ctx_create failed ({rc}); no usable HIP device {device}")
        self._h = h
        self.device = torch.device("cuda", device)
        self.unet_cfg = self.vq_cfg = self.vqenc_cfg = self.clip_cfg = self.rarm_cfg = None
        self.comm_world = 0                   # ranks of the context's RCCL communicator (comm_init), 0 = none

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "comm_world", 0):
                try:
                    self.comm_destroy()
                except Exception:
                    pass
            lib.rdm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RdmError(f"librdm_hip error {rc}: {lib.rdm_last_error(self._h).decode()}")

    def release_scratch(self):
        """Hand the grow-only work buffers (training scratch, split-K planes, ...) back to the allocator; re-created on demand."""
        self._check(lib.rdm_release_scratch(self._h))

    def use_current_stream(self):
        self._check(lib.rdm_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    # ---- weights
    def load_unet(self, cfg: UNetCfg, blob: np.ndarray):
        self._check(lib.rdm_load_unet(self._h, C.byref(cfg), blob.ctypes.data_as(_P), blob.nbytes)); self.unet_cfg = cfg

    def load_vq(self, cfg: VqCfg, blob: np.ndarray):
        self._check(lib.rdm_load_vq(self._h, C.byref(cfg), blob.ctypes.data_as(_P), blob.nbytes)); self.vq_cfg = cfg

    def load_vq_encoder(self, cfg: VqCfg, blob: np.ndarray):
        """First-stage ENCODER weights (`encoder.*`, `quant_conv.*` of the first-stage state dict; packing.pack("vqenc", cfg, sd))."""
        self._check(lib.rdm_load_vqenc(self._h, C.byref(cfg), blob.ctypes.data_as(_P), blob.nbytes)); self.vqenc_cfg = cfg

    def vq_encode(self, img):
        """VQModelInterface.encode: image f32 [b,out_ch,R,R] in [-1,1] -> latent f32 [b,embed_dim,R/f,R/f] (not quantised)."""
        img = self._dev(img, torch.float32)
        cfg = self._need("vq_encode", "vqenc")
        if img.ndim != 4 or tuple(img.shape[1:]) != (cfg.out_ch, cfg.resolution, cfg.resolution):
            raise RdmError(f"vq_encode: image must be [b,{cfg.out_ch},{cfg.resolution},{cfg.resolution}], got {tuple(img.shape)}")
        zr = cfg.resolution >> (cfg.n_ch_mult - 1)
        z = torch.empty((img.shape[0], cfg.embed_dim, zr, zr), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_vq_encode(self._h, _ptr(img), img.shape[0], _ptr(z)))
        return z

    def load_clip(self, cfg: ClipCfg, blob: np.ndarray):
        self._check(lib.rdm_load_clip(self._h, C.byref(cfg), blob.ctypes.data_as(_P), blob.nbytes)); self.clip_cfg = cfg

    def load_rarm(self, cfg: RarmCfg, blob: np.ndarray):
        self._check(lib.rdm_load_rarm(self._h, C.byref(cfg), blob.ctypes.data_as(_P), blob.nbytes)); self.rarm_cfg = cfg

    # ---- RARM
    def _check_rarm(self, what, context, b):
        if self.rarm_cfg is None:
            raise RdmError(f"{what}: rarm weights not loaded")
        if context.ndim != 3 or context.shape[0] != b or context.shape[2] != self.rarm_cfg.context_dim:
            raise RdmError(f"{what}: neighbours must be [b={b},k,{self.rarm_cfg.context_dim}], got {tuple(context.shape)}")

    def _check_ids(self, what, ids, n, name):
        """nn.Embedding / get_codebook_entry raise IndexError on an id outside the table; the kernels would silently read row 0.
        One tiny reduction + a host read per call (not per token)."""
        if ids.numel():
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= n:
                raise RdmError(f"{what}: {name} must lie in [0, {n}), got values in [{lo}, {hi}]")

    def rarm_forward(self, tokens, context):
        """RetrievalPatchTransformer.forward: tokens int64 [b,t], context f32 [b,k,ctx] -> logits f32 [b,t,vocab_out]."""
        tokens = self._dev(tokens, torch.int64); context = self._dev(context, torch.float32)
        b, t = tokens.shape
        self._check_rarm("rarm_forward", context, b)
        self._check_ids("rarm_forward", tokens, self.rarm_cfg.vocab_in, "tokens")
        out = torch.empty((b, t, self.rarm_cfg.vocab_out), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_rarm_forward(self._h, _ptr(tokens), b, t, _ptr(context), context.shape[1], _ptr(out)))
        return out

    def rarm_sample(self, cond_tokens, context, steps, uniforms, temperature=1.0, top_k=None, guidance_scale=1.0):
        """LatentImageRETRO.sample with sample=True: -> tokens int64 [b,steps].  uniforms f32 [steps,b] in [0,1)."""
        cond_tokens = self._dev(cond_tokens, torch.int64); context = self._dev(context, torch.float32)
        uniforms = self._dev(uniforms, torch.float32)
        b, tc = cond_tokens.shape
        self._check_rarm("rarm_sample", context, b)
        self._check_ids("rarm_sample", cond_tokens, self.rarm_cfg.vocab_in, "cond_tokens")
        if tuple(uniforms.shape) != (steps, b):
            raise RdmError(f"rarm_sample: uniforms must be [{steps},{b}], got {tuple(uniforms.shape)}")
        a = RarmSampleArgs(batch=b, k=context.shape[1], cond_len=tc, steps=steps, temperature=temperature,
                           top_k=int(top_k) if top_k is not None else 0, guidance_scale=guidance_scale)
        out = torch.empty((b, steps), device=self.device, dtype=torch.int64)
        self._check(lib.rdm_rarm_sample(self._h, C.byref(a), _ptr(cond_tokens), _ptr(context), _ptr(uniforms), _ptr(out)))
        return out

    def vq_decode_indices(self, indices):
        """decode_to_img: code indices int64 [b, h*w] -> image f32 [b,out_ch,R,R] (VQGAN first stage with a wide latent)."""
        indices = self._dev(indices, torch.int64)
        self._need("vq_decode_indices", "vq")
        zr = self.vq_cfg.resolution >> (self.vq_cfg.n_ch_mult - 1)
        if indices.ndim != 2 or indices.shape[1] != zr * zr:
            raise RdmError(f"vq_decode_indices: indices must be [b,{zr * zr}], got {tuple(indices.shape)}")
        self._check_ids("vq_decode_indices", indices, self.vq_cfg.n_embed, "indices")
        img = torch.empty((indices.shape[0], self.vq_cfg.out_ch, self.vq_cfg.resolution, self.vq_cfg.resolution), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_vq_decode_indices(self._h, _ptr(indices), indices.shape[0], _ptr(img)))
        return img

    # ---- model calls (torch CUDA tensors in / out)
    def _dev(self, t, dtype):
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _need(self, what, name):
        """The binding sizes its outputs from the loaded configuration: fail like the C ABI does (a message, not an AttributeError)."""
        cfg = getattr(self, name + "_cfg")
        if cfg is None:
            raise RdmError(f"{what}: {name} weights not loaded (load_{name})")
        return cfg

    def _check_sampler_shapes(self, what, x, cond, uncond=None, noise=None, steps=None):
        """The C ABI takes raw pointers: reject every shape it would silently mis-read (the reference raises a torch
        shape error in the same situations, e.g. `torch.cat([c, uc])` in ddim.py:232 for an unconditional conditioning of
        the wrong rank)."""
        if self.unet_cfg is None:
            raise RdmError(f"{what}: unet weights not loaded")
        cd, cin = self.unet_cfg.context_dim, self.unet_cfg.in_channels
        if x.ndim != 4 or x.shape[1] != cin:
            raise RdmError(f"{what}: latent must be [B,{cin},H,W], got {tuple(x.shape)}")
        if cond.ndim != 3 or cond.shape[0] != x.shape[0] or cond.shape[2] != cd or cond.shape[1] < 1:
            raise RdmError(f"{what}: conditioning must be [B={x.shape[0]},k,{cd}], got {tuple(cond.shape)}")
        if uncond is not None and tuple(uncond.shape) != tuple(cond.shape):
            raise RdmError(f"{what}: unconditional conditioning {tuple(uncond.shape)} must match the conditioning {tuple(cond.shape)}")
        if noise is not None and tuple(noise.shape) != (steps,) + tuple(x.shape):
            raise RdmError(f"{what}: noise stack must be {(steps,) + tuple(x.shape)}, got {tuple(noise.shape)}")

    def unet_forward(self, x, t, context):
        x = self._dev(x, torch.float32); t = self._dev(t, torch.int64); context = self._dev(context, torch.float32)
        self._check_sampler_shapes("unet_forward", x, context)
        if t.ndim != 1 or t.shape[0] != x.shape[0]:
            raise RdmError(f"unet_forward: timesteps must be [B={x.shape[0]}], got {tuple(t.shape)}")
        b, _, H, W = x.shape
        out = torch.empty((b, self.unet_cfg.out_channels, H, W), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_unet_forward(self._h, _ptr(x), _ptr(t), _ptr(context), b, context.shape[1], H, W, _ptr(out)))
        return out

    def ddim_sample(self, S, x_T, cond, uncond, alphas_cumprod, eta=0.0, scale=1.0, noise=None, log_every_t=100,
                    temperature=1.0, want_intermediates=False):
        x_T = self._dev(x_T, torch.float32); cond = self._dev(cond, torch.float32)
        uncond = None if uncond is None else self._dev(uncond, torch.float32)
        noise = None if noise is None else self._dev(noise, torch.float32)
        ac = np.ascontiguousarray(alphas_cumprod.detach().cpu().numpy() if isinstance(alphas_cumprod, torch.Tensor)
                                  else alphas_cumprod, dtype=np.float32)
        total = len(range(0, ac.shape[0], max(ac.shape[0] // max(int(S), 1), 1)))
        self._check_sampler_shapes("ddim_sample", x_T, cond, uncond, noise if eta != 0.0 else None, total)
        if scale > 1.0 and uncond is None:
            raise RdmError("ddim_sample: unconditional_conditioning is required when unconditional_guidance_scale > 1")
        B, Cc, H, W = x_T.shape
        a = DdimArgs(S=S, batch=B, k=cond.shape[1], channels=Cc, height=H, width=W, eta=eta, temperature=temperature,
                     unconditional_guidance_scale=scale, log_every_t=log_every_t, T=ac.shape[0],
                     alphas_cumprod=ac.ctypes.data_as(C.POINTER(C.c_float)))
        z = torch.empty_like(x_T)
        xi = pi = None
        if want_intermediates:
            T_ = ac.shape[0]; c_ = T_ // S
            total = len(range(0, T_, c_))
            n = lib.rdm_ddim_num_intermediates(total, log_every_t)
            xi = torch.empty((n,) + tuple(x_T.shape), device=self.device, dtype=torch.float32)
            pi = torch.empty_like(xi)
        self._check(lib.rdm_ddim_sample(self._h, C.byref(a), _ptr(x_T), _ptr(cond), _ptr(uncond), _ptr(noise), _ptr(z),
                                        _ptr(xi), _ptr(pi)))
        return z, xi, pi

    def ddpm_sample(self, timesteps, x_T, cond, noise, sched, clip_denoised=True, temperature=1.0):
        x_T = self._dev(x_T, torch.float32); cond = self._dev(cond, torch.float32); noise = self._dev(noise, torch.float32)
        arrs = {k: np.ascontiguousarray(np.asarray(v, dtype=np.float32)) for k, v in sched.items()}
        fp = lambda k: arrs[k].ctypes.data_as(C.POINTER(C.c_float))
        self._check_sampler_shapes("ddpm_sample", x_T, cond, None, noise, int(timesteps))
        B, Cc, H, W = x_T.shape
        a = DdpmArgs(timesteps=timesteps, batch=B, k=cond.shape[1], channels=Cc, height=H, width=W,
                     clip_denoised=int(clip_denoised), temperature=temperature, T=arrs["posterior_mean_coef1"].shape[0],
                     sqrt_recip_alphas_cumprod=fp("sqrt_recip_alphas_cumprod"),
                     sqrt_recipm1_alphas_cumprod=fp("sqrt_recipm1_alphas_cumprod"),
                     posterior_mean_coef1=fp("posterior_mean_coef1"), posterior_mean_coef2=fp("posterior_mean_coef2"),
                     posterior_log_variance_clipped=fp("posterior_log_variance_clipped"))
        z = torch.empty_like(x_T)
        self._check(lib.rdm_ddpm_sample(self._h, C.byref(a), _ptr(x_T), _ptr(cond), _ptr(noise), _ptr(z)))
        return z

    def vq_decode(self, z, force_not_quantize=False, return_indices=False):
        z = self._dev(z, torch.float32)
        cfg = self._need("vq_decode", "vq")
        zr = cfg.resolution >> (cfg.n_ch_mult - 1)
        if z.ndim != 4 or tuple(z.shape[1:]) != (cfg.embed_dim, zr, zr):
            raise RdmError(f"vq_decode: latent must be [b,{cfg.embed_dim},{zr},{zr}], got {tuple(z.shape)}")
        b = z.shape[0]; r = self.vq_cfg.resolution
        img = torch.empty((b, self.vq_cfg.out_ch, r, r), device=self.device, dtype=torch.float32)
        idx = torch.empty((b * z.shape[2] * z.shape[3],), device=self.device, dtype=torch.int32) if return_indices else None
        self._check(lib.rdm_vq_decode(self._h, _ptr(z), b, int(force_not_quantize), _ptr(img), _ptr(idx)))
        return (img, idx) if return_indices else img

    def vq_quantize(self, z, return_indices=False):
        """first_stage_model.quantize(z): z f32 [b,3,h,w] -> z_q (straight-through form), optionally the code indices."""
        z = self._dev(z, torch.float32)
        cfg = self._need("vq_quantize", "vq")
        zr = cfg.resolution >> (cfg.n_ch_mult - 1)
        if z.ndim != 4 or tuple(z.shape[1:]) != (cfg.embed_dim, zr, zr):
            raise RdmError(f"vq_quantize: latent must be [b,{cfg.embed_dim},{zr},{zr}], got {tuple(z.shape)}")
        zq = torch.empty_like(z)
        idx = torch.empty((z.shape[0] * zr * zr,), device=self.device, dtype=torch.int32) if return_indices else None
        self._check(lib.rdm_vq_quantize(self._h, _ptr(z), z.shape[0], _ptr(zq), _ptr(idx)))
        return (zq, idx) if return_indices else zq

    def to_uint8(self, img):
        img = self._dev(img, torch.float32)
        b, c, h, w = img.shape
        out = torch.empty((b, h, w, c), device=self.device, dtype=torch.uint8)
        self._check(lib.rdm_to_uint8(self._h, _ptr(img), b, c, h, w, _ptr(out)))
        return out

    def clip_encode_text(self, tokens):
        tokens = self._dev(tokens, torch.int64)
        cfg = self._need("clip_encode_text", "clip")
        if tokens.ndim != 2 or tokens.shape[1] != cfg.context_length:
            raise RdmError(f"clip_encode_text: tokens must be [b,{cfg.context_length}], got {tuple(tokens.shape)}")
        out = torch.empty((tokens.shape[0], self.clip_cfg.embed_dim), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_clip_encode_text(self._h, _ptr(tokens), tokens.shape[0], _ptr(out)))
        return out

    def clip_encode_image(self, image):
        image = self._dev(image, torch.float32)
        cfg = self._need("clip_encode_image", "clip")
        if image.ndim != 4 or tuple(image.shape[1:]) != (3, cfg.image_resolution, cfg.image_resolution):
            raise RdmError(f"clip_encode_image: image must be [b,3,{cfg.image_resolution},{cfg.image_resolution}] (already resized / normalised), got {tuple(image.shape)}")
        out = torch.empty((image.shape[0], self.clip_cfg.embed_dim), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_clip_encode_image(self._h, _ptr(image), image.shape[0], _ptr(out)))
        return out

    def clip_preprocess(self, image):
        """[b,3,h,w] in [-1,1] -> bicubic resize to the tower resolution + CLIP normalisation, f32 [b,3,R,R]."""
        image = self._dev(image, torch.float32)
        if image.ndim != 4 or image.shape[1] != 3:
            raise RdmError(f"clip_preprocess: image must be [b,3,h,w], got {tuple(image.shape)}")
        r = self._need("clip_preprocess", "clip").image_resolution
        out = torch.empty((image.shape[0], 3, r, r), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_clip_preprocess(self._h, _ptr(image), image.shape[0], image.shape[2], image.shape[3], _ptr(out)))
        return out

    def clip_encode_image_raw(self, image):
        """ClipImageRetriever.forward: preprocess fused into the image tower's patch gather."""
        image = self._dev(image, torch.float32)
        if image.ndim != 4 or image.shape[1] != 3:
            raise RdmError(f"clip_encode_image_raw: image must be [b,3,h,w], got {tuple(image.shape)}")
        self._need("clip_encode_image_raw", "clip")
        out = torch.empty((image.shape[0], self.clip_cfg.embed_dim), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_clip_encode_image_raw(self._h, _ptr(image), image.shape[0], image.shape[2], image.shape[3], _ptr(out)))
        return out

    # ---- retrieval
    def db_load(self, emb):
        """emb: numpy fp16/fp32 [n,dim] (host) or torch CUDA tensor (device)."""
        if isinstance(emb, torch.Tensor) and emb.is_cuda:
            emb = emb.contiguous()
            dt = {torch.float16: 0, torch.float32: 1}[emb.dtype]
            self._check(lib.rdm_db_load(self._h, _ptr(emb), emb.shape[0], emb.shape[1], dt, 1))
        else:
            emb = np.ascontiguousarray(emb.numpy() if isinstance(emb, torch.Tensor) else emb)
            dt = {np.dtype(np.float16): 0, np.dtype(np.float32): 1}[emb.dtype]
            self._check(lib.rdm_db_load(self._h, emb.ctypes.data_as(_P), emb.shape[0], emb.shape[1], dt, 0))

    def db_size(self):
        return int(lib.rdm_db_size(self._h))

    def knn(self, q, k, f64=False):
        """-> (idx int32 holding uint32 bits [b,k], score f32 [b,k]); f64=True: the fp64 scores the ranking was made on (shard merges)."""
        q = self._dev(q, torch.float32)
        if q.ndim != 2:
            raise RdmError(f"knn: queries must be [b,dim], got {tuple(q.shape)}")
        idx = torch.empty((q.shape[0], k), device=self.device, dtype=torch.int32)   # uint32 bits
        sc = torch.empty((q.shape[0], k), device=self.device, dtype=torch.float64 if f64 else torch.float32)
        fn = lib.rdm_knn_f64 if f64 else lib.rdm_knn
        self._check(fn(self._h, _ptr(q), q.shape[0], k, _ptr(idx), _ptr(sc)))
        return idx, sc

    def knn_last_fallback(self):
        return int(lib.rdm_knn_last_fallback(self._h))

    def db_gather(self, idx, dim):
        idx = idx.contiguous()
        out = torch.empty((idx.numel(), dim), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_db_gather(self._h, _ptr(idx), idx.numel(), _ptr(out)))
        return out.reshape(tuple(idx.shape) + (dim,))

    # ---- measurement
    def prof_enable(self, kinds=(0, 1)):
        """kinds: iterable of PROF_* kernel classes to bracket with HIP events (False / () = off; True = conv3x3 + linear)."""
        if kinds is True:
            kinds = (PROF_CONV3X3, PROF_LINEAR)
        mask = 0
        for k in (kinds or ()):
            mask |= 1 << int(k)
        self._check(lib.rdm_prof_enable(self._h, mask))

    def set_deterministic(self, on: bool = True):
        """Batch-invariant execution (include/rdm_hip.h rdm_set_deterministic): bitwise the same row whatever batch / rank count."""
        self._check(lib.rdm_set_deterministic(self._h, int(bool(on))))

    @property
    def deterministic(self) -> bool:
        return lib.rdm_get_deterministic(self._h) == 1

    # ---- RCCL through the C ABI (include/rdm_hip.h "multi-GPU"); the package's own multi-GPU path uses torch.distributed
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._check(lib.rdm_comm_unique_id(self._h, buf))
        return buf.raw

    def new_comm_context(self):
        """A sibling context on the same device whose ONLY job is to own the RCCL communicator (parallel.attach_library_comm): its stream is a
        fresh non-blocking side stream, so a rendezvous or probe collective that never completes can block nothing but that stream, and a
        context that failed its hand-shake is simply abandoned (advisor, round 5: a late helper thread must not share the product context,
        which is not thread-safe, nor enqueue on the product's stream)."""
        c = Context(self.device.index)
        c._side_stream = torch.cuda.Stream(self.device)
        c._check(lib.rdm_set_stream(c._h, C.c_void_p(c._side_stream.cuda_stream)))
        return c

    def comm_init(self, uid: bytes, rank: int, world: int):
        if len(uid) != 128:
            raise RdmError(f"comm_init: the unique id is 128 bytes, got {len(uid)}")
        self._check(lib.rdm_comm_init(self._h, C.create_string_buffer(uid, 128), int(rank), int(world)))
        self.comm_world = int(world)

    def comm_all_gather(self, local: torch.Tensor, world: int) -> torch.Tensor:
        local = local.contiguous()
        out = torch.empty((world,) + tuple(local.shape), device=self.device, dtype=local.dtype)
        self._check(lib.rdm_comm_all_gather(self._h, _ptr(local), _ptr(out), local.numel() * local.element_size()))
        return out

    def comm_all_reduce(self, buf: torch.Tensor, average=True):
        """In-place sum (mean) over the ranks of an fp32 device tensor through the context's RCCL communicator."""
        if buf.dtype != torch.float32 or not buf.is_contiguous():
            raise RdmError("comm_all_reduce: contiguous float32 tensor required")
        self._check(lib.rdm_comm_all_reduce_f32(self._h, _ptr(buf), buf.numel(), int(bool(average))))
        return buf

    def comm_destroy(self):
        self._check(lib.rdm_comm_destroy(self._h))
        self.comm_world = 0

    def prof_reset(self):
        self._check(lib.rdm_prof_reset(self._h))

    def debug_tap(self, buf, block, sub=0):
        """The next UNet forwards copy one intermediate activation (block `block`, stage `sub`: include/rdm_hip.h) into `buf` (a bf16
        device tensor); buf None: off."""
        self._tap_keepalive = buf
        self._check(lib.rdm_debug_tap(self._h, _ptr(buf) if buf is not None else None, 0 if buf is None else buf.numel() * buf.element_size(), int(block), int(sub)))

    def op_ffn_fused(self, l3, t2, xin, w1, b1, wf, bf):
        """[x gelu(g) | t2] wf^T + bf + xin with [x | g] = l3 w1^T + b1 in one kernel (rdm_op_ffn_fused; C = 384, M % 128 == 0)."""
        M, Cc = l3.shape
        out = torch.empty((M, Cc), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_ffn_fused(self._h, _ptr(l3), _ptr(t2), _ptr(xin), _ptr(w1), _ptr(b1), _ptr(wf), _ptr(bf), _ptr(out), M, Cc))
        return out

    def debug_counter(self, which=0):
        v = C.c_ulonglong(0)
        self._check(lib.rdm_debug_counter(self._h, int(which), C.byref(v)))
        return int(v.value)

    def calib_probe(self, mfma_ms=800.0, stream_bytes=1 << 30, stream_reps=6):
        """Box calibration (rdm_calib_probe): (sustained dense-bf16 MFMA TFLOP/s on random operands, HBM copy GB/s read + write)."""
        buf = torch.empty(max(1 << 20, 2 * stream_bytes), device=self.device, dtype=torch.uint8)
        tf, gb = C.c_double(0.0), C.c_double(0.0)
        self._check(lib.rdm_calib_probe(self._h, _ptr(buf), buf.numel(), float(mfma_ms), int(stream_bytes), int(stream_reps), C.byref(tf), C.byref(gb)))
        del buf
        return float(tf.value), float(gb.value)

    def prof_dump(self, path):
        """One CSV row per recorded launch (kind, role tag, shape, ms, work): tools/op_trace.py."""
        self._check(lib.rdm_prof_dump(self._h, os.fsencode(path)))

    def prof_collect(self, kind):
        n, ms, fl = C.c_longlong(0), C.c_double(0), C.c_double(0)
        self._check(lib.rdm_prof_collect(self._h, kind, C.byref(n), C.byref(ms), C.byref(fl)))
        return int(n.value), float(ms.value), float(fl.value)

    # ---- operator-level (parity tests)
    def op_linear(self, a, w, bias=None, residual=None, act=ACT_NONE, alpha=1.0, out_f32=False):
        M, K = a.shape; N = w.shape[0]
        No = N // 2 if act == ACT_GEGLU else N
        ob = None if out_f32 else torch.empty((M, No), device=self.device, dtype=torch.bfloat16)
        of = torch.empty((M, No), device=self.device, dtype=torch.float32) if out_f32 else None
        self._check(lib.rdm_op_linear(self._h, _ptr(a), _ptr(w), _ptr(bias), _ptr(residual), _ptr(ob), _ptr(of), M, N, K,
                                      act, float(alpha)))
        return of if out_f32 else ob

    def op_linear_rowvec(self, a, w, bias, rowvec, rows_per_group, residual=None):
        """a w^T + bias + rowvec[row // rows_per_group] (+ residual): rowvec f32 [groups, N]."""
        M, K = a.shape; N = w.shape[0]
        assert rowvec.dtype == torch.float32 and rowvec.shape == ((M + rows_per_group - 1) // rows_per_group, N) and rowvec.is_contiguous()
        out = torch.empty((M, N), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_linear_rowvec(self._h, _ptr(a), _ptr(w), _ptr(bias), _ptr(rowvec), int(rows_per_group), _ptr(residual), _ptr(out), M, N, K))
        return out

    def op_linear_ln(self, x, w, bias, gamma, beta, act=ACT_NONE, eps=1e-5):
        """act(LayerNorm(x) w^T + bias) with the LayerNorm folded into the GEMM (raises RdmError for shapes the folded kernel does not take)."""
        M, K = x.shape; N = w.shape[0]
        out = torch.empty((M, N // 2 if act == ACT_GEGLU else N), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_linear_ln(self._h, _ptr(x), _ptr(w), _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(out), M, N, K, act, float(eps)))
        return out

    def op_conv3x3(self, x0, w, bias, x1=None, rowvec=None, residual=None, stride=1, ups=0):
        B, Hin, Win, C0 = x0.shape
        C1 = 0 if x1 is None else x1.shape[3]
        N = w.shape[0]
        Ho = Hin * 2 if ups else (Hin // 2 if stride == 2 else Hin)
        Wo = Win * 2 if ups else (Win // 2 if stride == 2 else Win)
        out = torch.empty((B, Ho, Wo, N), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_conv3x3(self._h, _ptr(x0), _ptr(x1), C0, C1, _ptr(w), _ptr(bias), _ptr(rowvec),
                                       0 if rowvec is None else rowvec.shape[1], _ptr(residual), _ptr(out), B, Hin, Win, N,
                                       stride, ups))
        return out

    def op_rarm_sampler(self, logits, uniforms, guidance_scale=1.0, temperature=1.0, top_k=None):
        """The sampler kernel alone: logits f32 [(2 if guided else 1) * b, vocab] (conditional rows first), uniforms f32 [b] -> int64 [b]."""
        logits = self._dev(logits, torch.float32); uniforms = self._dev(uniforms, torch.float32)
        b = uniforms.shape[0]
        cfg = guidance_scale > 1.0
        if logits.ndim != 2 or logits.shape[0] != (2 * b if cfg else b):
            raise RdmError(f"op_rarm_sampler: logits must be [{2 * b if cfg else b}, vocab], got {tuple(logits.shape)}")
        out = torch.empty((b,), device=self.device, dtype=torch.int64)
        self._check(lib.rdm_op_rarm_sampler(self._h, _ptr(logits), b, logits.shape[1], int(cfg), float(guidance_scale), float(temperature),
                                            int(top_k) if top_k is not None else 0, _ptr(uniforms), _ptr(out)))
        return out

    # ---- backward building blocks (include/rdm_hip.h "backward"; composed in rdm_amd/training.py)
    def op_conv3x3_dgrad(self, dy, w):
        B, H, W, N = dy.shape
        Cc = w.shape[3]
        dx = torch.empty((B, H, W, Cc), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_conv3x3_dgrad(self._h, _ptr(dy), _ptr(w), _ptr(dx), B, H, W, Cc, N))
        return dx

    def op_conv3x3_wgrad(self, x, dy):
        B, H, W, Cc = x.shape
        N = dy.shape[3]
        dw = torch.empty((N, 3, 3, Cc), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_op_conv3x3_wgrad(self._h, _ptr(x), _ptr(dy), _ptr(dw), B, H, W, Cc, N))
        return dw

    def op_groupnorm_bwd(self, x, dy, gamma, beta, eps, silu, residual=None):
        """-> dx (+ residual when given: the skip path's gradient joins inside the kernel), dgamma, dbeta."""
        B, HW, Cc = x.shape
        dx = torch.empty_like(x); dg = torch.empty((Cc,), device=self.device, dtype=torch.float32); db = torch.empty_like(dg)
        assert residual is None or (residual.is_contiguous() and residual.numel() == x.numel())
        self._check(lib.rdm_op_groupnorm_bwd_add(self._h, _ptr(x), _ptr(dy), _ptr(gamma), _ptr(beta), B, HW, Cc, float(eps), int(silu), _ptr(residual),
                                                 _ptr(dx), _ptr(dg), _ptr(db)))
        return dx, dg, db

    def op_layernorm_bwd(self, x, dy, gamma, eps=1e-5, residual=None):
        M, Cc = x.shape
        dx = torch.empty_like(x); dg = torch.empty((Cc,), device=self.device, dtype=torch.float32); db = torch.empty_like(dg)
        assert residual is None or (residual.is_contiguous() and residual.numel() == x.numel())
        self._check(lib.rdm_op_layernorm_bwd_add(self._h, _ptr(x), _ptr(dy), _ptr(gamma), M, Cc, float(eps), _ptr(residual), _ptr(dx), _ptr(dg), _ptr(db)))
        return dx, dg, db

    def op_colsum(self, x):
        M, N = x.shape
        out = torch.empty((N,), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_op_colsum(self._h, _ptr(x), _ptr(out), M, N))
        return out

    def op_transpose(self, x):
        r, c_ = x.shape
        y = torch.empty((c_, r), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_transpose(self._h, _ptr(x), _ptr(y), r, c_))
        return y

    def op_add(self, a, b):
        out = torch.empty_like(a)
        self._check(lib.rdm_op_add(self._h, _ptr(a), _ptr(b), _ptr(out), a.numel()))
        return out

    def op_linear_wgrad(self, dy, a):
        """dy bf16 [M, N], a bf16 [M, K] -> dy^T a fp32 [N, K] (the weight gradient of y = a w^T)."""
        M, N = dy.shape; K = a.shape[1]
        dw = torch.empty((N, K), device=self.device, dtype=torch.float32)
        self._check(lib.rdm_op_linear_wgrad(self._h, _ptr(dy), _ptr(a), _ptr(dw), M, N, K))
        return dw

    @staticmethod
    def _ptr_array(ts, optional=False):
        arr = (C.c_void_p * len(ts))()
        for i, t in enumerate(ts):
            if t is None:
                assert optional
                arr[i] = None
            else:
                assert t.is_contiguous(), "tensor must be contiguous"
                arr[i] = t.data_ptr()
        return arr

    def op_adamw_multi(self, ps, gs, ms, vs, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, p_bf16s=None):
        """op_adamw over lists of fp32 tensors (48 tensors per launch); p_bf16s: list with None where a tensor has no bf16 working copy."""
        n = len(ps)
        assert n == len(gs) == len(ms) == len(vs) and all(g.numel() == p.numel() and g.dtype == torch.float32 for p, g in zip(ps, gs))
        numel = (C.c_longlong * n)(*[p.numel() for p in ps])
        self._check(lib.rdm_op_adamw_multi(self._h, n, self._ptr_array(ps), self._ptr_array(gs), self._ptr_array(ms), self._ptr_array(vs),
                                           self._ptr_array(p_bf16s, True) if p_bf16s is not None else None, numel, float(lr), float(betas[0]),
                                           float(betas[1]), float(eps), float(weight_decay), int(step)))

    def op_ema_multi(self, shadows, ps, one_minus_decay):
        n = len(ps)
        numel = (C.c_longlong * n)(*[p.numel() for p in ps])
        self._check(lib.rdm_op_ema_multi(self._h, n, self._ptr_array(shadows), self._ptr_array(ps), numel, float(one_minus_decay)))

    def op_ema(self, shadow, p, one_minus_decay):
        self._check(lib.rdm_op_ema(self._h, _ptr(shadow), _ptr(p), p.numel(), float(one_minus_decay)))

    def op_silu(self, x, dy=None):
        """x fp32: -> silu(x) bf16, or with dy (fp32) the gradient dy * silu'(x) fp32."""
        out = torch.empty(x.shape, device=self.device, dtype=torch.bfloat16 if dy is None else torch.float32)
        self._check(lib.rdm_op_silu(self._h, _ptr(x), _ptr(dy) if dy is not None else None, _ptr(out), x.numel()))
        return out

    # ---- training-step glue (include/rdm_hip.h: rdm_op_q_sample ...)
    def op_q_sample(self, x0, noise, sqrt_ac, sqrt_1mac, want_nchw=True, cpad=0):
        """x_t = sqrt_ac[b] x0 + sqrt_1mac[b] noise: -> (f32 NCHW or None, bf16 NHWC [B,H,W,cpad] or None)."""
        B, Cc, H, W = x0.shape
        out = torch.empty_like(x0) if want_nchw else None
        onh = torch.empty((B, H, W, cpad), device=self.device, dtype=torch.bfloat16) if cpad else None
        self._check(lib.rdm_op_q_sample(self._h, _ptr(x0), _ptr(noise), _ptr(sqrt_ac), _ptr(sqrt_1mac), _ptr(out), _ptr(onh), B, Cc, H, W, cpad))
        return out, onh

    def op_mse_loss(self, eps_nhwc, target, coef=None, C_=None):
        """eps bf16 [B,H,W,ldc], target f32 [B,C,H,W] -> (se f32 [B], deps bf16 [B,H,W,ldc] = coef[b] (eps - target), or None without coef)."""
        B, H, W, ldc = eps_nhwc.shape
        Cc = target.shape[1] if C_ is None else C_
        se = torch.empty((B,), device=self.device, dtype=torch.float32)
        deps = torch.empty_like(eps_nhwc) if coef is not None else None
        self._check(lib.rdm_op_mse_loss(self._h, _ptr(eps_nhwc), _ptr(target), _ptr(coef), _ptr(se), _ptr(deps), B, Cc, H, W, ldc))
        return se, deps

    def op_where_rows(self, mask, a, x):
        """out[b] = a[b] if mask[b] else x[b]; mask uint8 / bool [B], a / x f32 [B, ...]."""
        out = torch.empty_like(x)
        m8 = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        self._check(lib.rdm_op_where_rows(self._h, _ptr(m8), _ptr(a), _ptr(x), _ptr(out), x.shape[0], x[0].numel()))
        return out

    def op_timestep_embedding(self, t, dim, ld=None):
        ld = dim if ld is None else ld
        out = torch.empty((t.shape[0], ld), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_timestep_embedding(self._h, _ptr(t), _ptr(out), t.shape[0], dim, ld))
        return out

    def op_colsum_samples(self, x):
        """x bf16 [B, HW, N] -> bf16 [B, N]."""
        B, HW, N = x.shape
        out = torch.empty((B, N), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_colsum_samples(self._h, _ptr(x), _ptr(out), B, HW, N))
        return out

    def op_expand2(self, x, mode):
        """x bf16 [B,H,W,C] -> [B,2H,2W,C]: mode 0 zero insertion, mode 1 nearest-neighbour copy."""
        B, H, W, Cc = x.shape
        out = torch.empty((B, 2 * H, 2 * W, Cc), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_expand2(self._h, _ptr(x), _ptr(out), B, H, W, Cc, mode))
        return out

    def op_sumpool2(self, x):
        B, H2, W2, Cc = x.shape
        out = torch.empty((B, H2 // 2, W2 // 2, Cc), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_sumpool2(self._h, _ptr(x), _ptr(out), B, H2 // 2, W2 // 2, Cc))
        return out

    def op_adamw(self, p, g, m, v, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, p_bf16=None):
        """In-place AdamW step on fp32 tensors p / m / v with gradient g; p_bf16 (same shape, bf16) receives the rounded new parameters."""
        self._check(lib.rdm_op_adamw(self._h, _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(p_bf16) if p_bf16 is not None else None, p.numel(),
                                     float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step)))

    def op_attention_bwd(self, q, k, v, o, dout, heads):
        """fused attention backward (d_head 32): q / o / dout bf16 [B, n, C], k / v [B, m, C] -> dq, dk, dv."""
        B, n, _ = q.shape; m = k.shape[1]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        self._check(lib.rdm_op_attention_bwd(self._h, _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(dout), B, n, m, heads, _ptr(dq), _ptr(dk), _ptr(dv)))
        return dq, dk, dv

    def op_bmm(self, a, w, alpha=1.0, out_f32=False):
        """a bf16 [Z, M, K], w bf16 [Z, N, K] -> alpha * a w^T [Z, M, N] (bf16, or fp32 with out_f32)."""
        Z, M, K = a.shape; N = w.shape[1]
        out = torch.empty((Z, M, N), device=self.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
        self._check(lib.rdm_op_bmm(self._h, _ptr(a), _ptr(w), None if out_f32 else _ptr(out), _ptr(out) if out_f32 else None, Z, M, N, K, float(alpha)))
        return out

    def op_heads(self, x, H, D, mode, n=None):
        """mode 0 / 1: x bf16 [B, n, >= H D] -> per-head [B H, n, 64] / transposed [B H, 64, n]; mode 2: [B H, n, 64] -> [B, n, H D]."""
        if mode == 2:
            BH, n_, _ = x.shape; B = BH // H
            out = torch.empty((B, n_, H * D), device=self.device, dtype=torch.bfloat16)
            self._check(lib.rdm_op_heads(self._h, _ptr(x), _ptr(out), B, n_, H, D, H * D, 2))
            return out
        B, n_, ldx = x.shape
        out = torch.empty((B * H, n_, 64) if mode == 0 else (B * H, 64, n_), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_heads(self._h, _ptr(x), _ptr(out), B, n_, H, D, ldx, mode))
        return out

    def op_transpose_batched(self, x):
        Z, r, c_ = x.shape
        y = torch.empty((Z, c_, r), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_transpose_batched(self._h, _ptr(x), _ptr(y), Z, r, c_))
        return y

    def op_softmax(self, s, n_valid=0):
        p = torch.empty(s.shape, device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_softmax(self._h, _ptr(s), _ptr(p), s.numel() // s.shape[-1], s.shape[-1], int(n_valid)))
        return p

    def op_softmax_bwd(self, p, dp):
        ds = torch.empty_like(p)
        self._check(lib.rdm_op_softmax_bwd(self._h, _ptr(p), _ptr(dp), _ptr(ds), p.numel() // p.shape[-1], p.shape[-1]))
        return ds

    def op_geglu(self, pre, dh=None):
        """pre bf16 [M, 2F] = [x | gate] (unpermuted).  dh None -> x * gelu(gate) [M, F]; else the gradient [dx | dgate] [M, 2F]."""
        M, F2 = pre.shape
        out = torch.empty((M, F2 // 2) if dh is None else (M, F2), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_geglu(self._h, _ptr(pre), _ptr(dh) if dh is not None else None, _ptr(out), M, F2 // 2))
        return out

    def op_groupnorm(self, x0, gamma, beta, eps, silu, x1=None):
        B, HW, C0 = x0.shape
        C1 = 0 if x1 is None else x1.shape[2]
        out = torch.empty((B, HW, C0 + C1), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_groupnorm(self._h, _ptr(x0), _ptr(x1), C0, C1, B, HW, _ptr(gamma), _ptr(beta), float(eps),
                                         int(silu), _ptr(out)))
        return out

    def op_layernorm(self, x, gamma, beta, eps=1e-5):
        M, Cc = x.shape
        out = torch.empty((M, Cc), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_layernorm(self._h, _ptr(x), int(x.dtype == torch.float32), _ptr(gamma), _ptr(beta), M, Cc,
                                         float(eps), _ptr(out)))
        return out

    def op_self_attention(self, qk, vt, heads):
        B, n, C2 = qk.shape
        out = torch.empty((B, n, C2 // 2), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_self_attention(self._h, _ptr(qk), _ptr(vt), B, n, heads, _ptr(out)))
        return out

    def op_self_attention_qkv(self, qkv, heads):
        """qkv bf16 [B, n, 3C] = [q | k | v] (one fused projection), n % 64 == 0 -> [B, n, C]."""
        B, n, C3 = qkv.shape
        out = torch.empty((B, n, C3 // 3), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_self_attention_qkv(self._h, _ptr(qkv), B, n, heads, _ptr(out)))
        return out

    def op_head_conv(self, x, w, bias, gn=None):
        """GroupNorm(32) + SiLU (gn = (gamma, beta, eps); None: no norm) + 3x3 conv to few channels: x bf16 [B, H, W, C], w fp32 [Cout, C, 3, 3]
        -> fp32 [B, Cout, H, W]."""
        B, H, W, Cc = x.shape
        Cout = w.shape[0]
        out = torch.empty((B, Cout, H, W), device=self.device, dtype=torch.float32)
        g, b_, eps = gn if gn is not None else (None, None, 0.0)
        opt = lambda t: _ptr(t) if t is not None else None
        self._check(lib.rdm_op_head_conv(self._h, _ptr(x), opt(g), opt(b_), float(eps), _ptr(w), opt(bias), B, H, W, Cc, Cout, _ptr(out)))
        return out

    def op_xattn_fused(self, x, G, U, bias, res, ncols, group, ln=None):
        """out = softmax_groups(x G^T) U^T + bias + res per sample: x / res bf16 [B, n, C], G bf16 [B, NP, C], U bf16 [B, C, NP].
        ln = (gamma, beta, eps): scores on LayerNorm(x), residual = x (res must be None)."""
        B, n, Cc = x.shape
        NP = G.shape[1]
        out = torch.empty((B, n, Cc), device=self.device, dtype=torch.bfloat16)
        g, b_, eps = ln if ln is not None else (None, None, 0.0)
        opt = lambda t: _ptr(t) if t is not None else None
        self._check(lib.rdm_op_xattn_fused(self._h, _ptr(x), opt(g), opt(b_), float(eps), _ptr(G), _ptr(U), opt(bias), opt(res),
                                           B, n, Cc, NP, ncols, group, _ptr(out)))
        return out

    def op_xattn_fused_ln3(self, x, G, U, bias, ncols, group, ln, ln3):
        """IN PLACE x <- softmax_groups(LayerNorm(x; ln) G^T) U^T + bias + x, and -> LayerNorm(new x; ln3) (bf16).  ln / ln3 = (gamma, beta[, eps])."""
        B, n, Cc = x.shape
        assert x.is_contiguous() and x.dtype == torch.bfloat16
        out3 = torch.empty_like(x)
        opt = lambda t: _ptr(t) if t is not None else None
        self._check(lib.rdm_op_xattn_fused_ln3(self._h, _ptr(x), _ptr(ln[0]), _ptr(ln[1]), float(ln[2] if len(ln) > 2 else 1e-5), _ptr(G), _ptr(U), opt(bias),
                                               B, n, Cc, G.shape[1], ncols, group, _ptr(ln3[0]), _ptr(ln3[1]), _ptr(out3)))
        return out3

    def op_small_attention_bwd(self, q, k, v, dout, heads, scale):
        """Gradient of op_small_attention at d_head 32 with 1..32 keys (the UNet's cross-attention): -> dq, dk, dv (bf16)."""
        B, nq, Cc = q.shape
        m = k.shape[1]
        assert k.stride(-2) == v.stride(-2) and q.is_contiguous() and dout.is_contiguous()
        dq = torch.empty((B, nq, heads * 32), device=self.device, dtype=torch.bfloat16)
        dk = torch.empty((B, m, heads * 32), device=self.device, dtype=torch.bfloat16); dv = torch.empty_like(dk)
        self._check(lib.rdm_op_small_attention_bwd(self._h, _ptr(q), Cc, _ptr(k), _ptr(v), k.shape[2], _ptr(dout), dout.shape[2], B, nq, m, heads, float(scale),
                                                   _ptr(dq), _ptr(dk), _ptr(dv)))
        return dq, dk, dv

    def op_small_attention(self, q, k, v, heads, D, causal, scale):
        B, nq, Cc = q.shape
        assert k.stride(-2) == v.stride(-2)
        out = torch.empty((B, nq, heads * D), device=self.device, dtype=torch.bfloat16)
        self._check(lib.rdm_op_small_attention(self._h, _ptr(q), Cc, _ptr(k), _ptr(v), k.shape[2], B, nq, k.shape[1], heads,
                                               D, int(causal), float(scale), _ptr(out), heads * D))
        return out
