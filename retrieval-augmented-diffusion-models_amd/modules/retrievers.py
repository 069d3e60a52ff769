"""Native counterparts of rdm/modules/retrievers.py:67-117 (ClipImageRetriever, CLIPTextEmbedder).

The CLIP towers run inside librdm_hip (rdm_clip_encode_text / rdm_clip_encode_image); the wrappers keep the
reference's call surface: `retriever(x)` on images in [-1, 1], `.model.encode_text(tokens)`, `.to(device)`, and they
return UN-normalised embeddings (callers do `.cpu().numpy()`, dsetbuilder.py:473).
"""
import torch

from .. import _lib, packing
from .custom_clip.tokenizer import tokenize


class _NativeClip:
    """`.model` of the retrievers: the two encode entry points of rdm/modules/custom_clip/model.py:304-320."""

    def __init__(self, ctx, cfg):
        self.ctx, self.cfg = ctx, cfg

    def encode_text(self, tokens):
        return self.ctx.clip_encode_text(torch.as_tensor(tokens))

    def encode_image(self, image):
        return self.ctx.clip_encode_image(image)

    def encode_image_raw(self, image):
        return self.ctx.clip_encode_image_raw(image)


def load_clip(name="ViT-B/32", device=0, jit=False, state_dict=None, ctx=None, clip_cfg=None):
    """Counterpart of `clip.load` (retrievers.py:76): builds the ViT-B/32 graph in the library and uploads
    `state_dict` (keys as in rdm/modules/custom_clip/model.py:363-399).  No network: weights must be supplied."""
    if name != "ViT-B/32" and clip_cfg is None:
        raise NotImplementedError(f"CLIP variant {name}: only ViT-B/32 (models/rdm/*/config.yaml:100) or an explicit clip_cfg")
    if state_dict is None:
        raise ValueError("load_clip needs a CLIP state_dict (no download path in this environment)")
    dev_index = device if isinstance(device, int) else (torch.device(device).index or 0)
    ctx = ctx if ctx is not None else _lib.Context(dev_index)
    cfg = clip_cfg if clip_cfg is not None else _lib.make_clip_cfg()
    ctx.load_clip(cfg, packing.pack("clip", cfg, state_dict))
    return _NativeClip(ctx, cfg), None


class ClipImageRetriever(object):
    def __init__(self, model="ViT-B/32", jit=False, device=0, antialias=False, state_dict=None, ctx=None, clip_cfg=None):
        self.model, _ = load_clip(name=model, device=device, jit=jit, state_dict=state_dict, ctx=ctx, clip_cfg=clip_cfg)
        if antialias:
            raise NotImplementedError("antialias=True (kornia's gaussian pre-blur): no shipped config sets it (retrievers.py:73)")
        self.antialias = antialias
        self.device = self.model.ctx.device
        self.mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device=self.device)
        self.std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device=self.device)

    def to(self, device): return self
    def eval(self): return self

    def preprocess(self, x):
        """retrievers.py:83-91: bicubic resize to the tower resolution (align_corners=True), [-1,1] -> [0,1], CLIP
        mean/std — one HIP kernel (rdm_clip_preprocess).  kornia 0.6.2's resize is `F.interpolate(mode='bicubic',
        align_corners=True)`, i.e. cubic convolution with A = -0.75 and clamped borders; that is what the kernel computes."""
        return self.model.ctx.clip_preprocess(x)

    def forward(self, x):
        """retrievers.py:93-95.  The resized image is not materialised: the resize feeds the patch-embedding GEMM directly."""
        return self.model.encode_image_raw(x)

    __call__ = forward


class CLIPTextEmbedder(object):
    """retrievers.py:98-117."""

    def __init__(self, model="ViT-B/32", device=0, add_k_shape=False, state_dict=None, ctx=None, clip_cfg=None, clip=None):
        self.model = clip if clip is not None else load_clip(model, device=device, state_dict=state_dict, ctx=ctx, clip_cfg=clip_cfg)[0]
        self.device = self.model.ctx.device
        self.add_k_shape = add_k_shape

    def preprocess(self, text):
        return torch.from_numpy(tokenize(text, self.model.cfg.context_length))

    def forward(self, txt):
        emb = self.model.encode_text(self.preprocess(txt))
        return emb[:, None] if self.add_k_shape else emb

    __call__ = forward
