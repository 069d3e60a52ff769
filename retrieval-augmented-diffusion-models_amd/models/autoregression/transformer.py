"""Native counterpart of the SAMPLING methods of rdm/models/autoregression/transformer.py::LatentImageRETRO (RARM):
`sample` (:224-294), `sampling_util` (:296-312), `sample_from_rdata` (:314-404), `get_qids` (:407-430), and of the taming
Net2NetTransformer pieces it inherits for sampling (`encode_to_c` with the SOSProvider, `decode_to_img`, `top_k_logits`).
Training, logging and image-patch neighbour encoders are out of scope (the shipped configs use
IdentityEncoder on CLIP embeddings, models/rarm/imagenet/dogs/config.yaml:10-13).

The transformer (rdm.modules.attention.RetrievalPatchTransformer, 18 x 768, causal self-attention + cross-attention to the k
retrieved neighbours) and the VQGAN-f16 decoder run inside librdm_hip; the 256-step loop is ONE library call
(rdm_rarm_sample) that decodes against a K/V cache — the reference re-runs the whole prefix for every token (:241-248).
The multinomial draw uses uniforms taken from torch's global generator on the model's device (so `seed_everything`
makes a run repeatable) and the inverse-CDF rule documented in include/rdm_hip.h.
"""
import numpy as np
import torch

from ... import _lib, packing


class LatentImageRETRO(object):
    def __init__(self, transformer_config, first_stage_config=None, mask_token=16384, sos_token=16385, nn_key="nn_embeddings",
                 nn_memory=None, id_count=None, retriever=None, k_nn=4, device=0, ctx=None, p_mask_max=0., **ignored):
        self._dev_index = device if isinstance(device, int) else (torch.device(device).index or 0)
        self._ctx = ctx
        self.device = getattr(ctx, "device", None) or torch.device("cuda", self._dev_index)
        tparams = transformer_config.get("params", transformer_config)
        self.rarm_cfg = _lib.make_rarm_cfg(**tparams)
        for flag, want in (("continuous", False), ("causal", True), ("cross_attend", True), ("positional_encodings", True)):
            if flag in tparams and bool(tparams[flag]) != want:
                raise NotImplementedError(f"RetrievalPatchTransformer with {flag}={tparams[flag]} (the shipped RARM configs use {want})")
        self.vq_cfg = None
        if first_stage_config is not None:
            fparams = first_stage_config.get("params", first_stage_config)
            dd = dict(fparams.get("ddconfig", {}))
            self.vq_cfg = _lib.make_vqgan_f16_cfg(embed_dim=fparams.get("embed_dim", 256), n_embed=fparams.get("n_embed", 16384),
                                                  z_channels=dd.get("z_channels", 256), ch=dd.get("ch", 128),
                                                  ch_mult=tuple(dd.get("ch_mult", (1, 1, 2, 2, 4))), num_res_blocks=dd.get("num_res_blocks", 2),
                                                  out_ch=dd.get("out_ch", 3), resolution=dd.get("resolution", 256),
                                                  attn_resolutions=tuple(dd.get("attn_resolutions", (16,))))
        self.sos_token, self.mask_token = int(sos_token), int(mask_token)
        self.nn_key, self.k_nn, self.p_mask_max = nn_key, k_nn, p_mask_max
        self.retriever = retriever
        self.nn_encoder = None                         # IdentityEncoder
        self.use_memory = nn_memory is not None
        if self.use_memory:
            self.nn_memory = torch.as_tensor(np.asarray(nn_memory))
        self.id_count = id_count

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.Context(self._dev_index)
        return self._ctx

    def eval(self): return self
    def to(self, device): return self

    # ---- weights: checkpoint keys `transformer.*` (RetrievalPatchTransformer) and `first_stage_model.*` (taming VQModel)
    def load_state_dict(self, sd, strict=True):
        tsd = packing.strip_prefix(sd, "transformer.") or sd
        self.load_transformer_state_dict(tsd)
        fsd = packing.strip_prefix(sd, "first_stage_model.")
        if fsd and self.vq_cfg is not None:
            self.load_first_stage_state_dict(fsd)
        return [], []

    def load_transformer_state_dict(self, tsd):
        self.ctx.load_rarm(self.rarm_cfg, packing.pack("rarm", self.rarm_cfg, tsd))

    def load_first_stage_state_dict(self, fsd):
        self.ctx.load_vq(self.vq_cfg, packing.pack("vq", self.vq_cfg, fsd))

    # ---- taming pieces
    def encode_to_c(self, c):
        """cond_stage_config '__is_unconditional__' -> SOSProvider: one sos token per sample."""
        n = c.shape[0]
        idx = torch.full((n, 1), self.sos_token, dtype=torch.long)
        return idx, idx

    @torch.no_grad()
    def decode_to_img(self, index, zshape):
        """Net2NetTransformer.decode_to_img: indices [b, h*w] -> image [b,3,256,256]."""
        return self.ctx.vq_decode_indices(index.reshape(index.shape[0], -1))

    def train_searcher(self):
        self.retriever.train_searcher()

    # ---- transformer.py:224-294
    @torch.no_grad()
    def sample(self, x, r, c, steps, temperature=1.0, sample=False, top_k=None, guidance_scale=1.0, callback=lambda k: None,
               uniforms=None, **kwargs):
        x = torch.cat((c.to(self.device), x.to(self.device)), 1)           # conditioning tokens, then any given prefix
        for k_ in range(steps):
            callback(k_)                                                    # the loop itself runs inside the library
        if uniforms is None:
            uniforms = torch.rand((steps, x.shape[0]), device=self.device)
        if not sample:
            top_k = 1                                                       # torch.topk(probs, 1): the arg-max token (:266-267)
        return self.ctx.rarm_sample(x, r, steps, uniforms, temperature=temperature, top_k=top_k, guidance_scale=guidance_scale)

    # ---- transformer.py:296-312
    @torch.no_grad()
    def sampling_util(self, steps, z_start, r, c, temperature, top_k, zshape, callback=None, top_p=1., **kwargs):
        assert top_p == 1., 'not yet implemented'
        index_sample = self.sample(z_start, r, c, steps=steps, temperature=temperature if temperature is not None else 1.0,
                                   sample=True, top_k=top_k if top_k is not None else 100,
                                   callback=callback if callback is not None else lambda k: None, **kwargs)
        return self.decode_to_img(index_sample, zshape)

    # ---- transformer.py:314-404 (IdentityEncoder branch; return_nns needs the raw patches: out of scope)
    @torch.no_grad()
    def sample_from_rdata(self, N, cond=None, return_nns=False, use_weights=False, qids=None, k_nn=None, memsize=100, verbose=False,
                          top_k=256, temperature=1.0, code_side_len=16, z_dimensionality=256, pre_loaded_patches=None,
                          nn_embeddings=None, query_embeddings=None, **kwargs):
        if return_nns or pre_loaded_patches is not None:
            raise NotImplementedError("return_nns / pre_loaded_patches need the raw OpenImages patches (out of scope, SURVEY.md §2 #8)")
        if cond is not None:
            raise NotImplementedError()
        if self.retriever is not None and self.retriever.searcher is None:
            self.train_searcher()
        if k_nn is None:
            k_nn = self.k_nn
        out = {}
        if nn_embeddings is None:
            if query_embeddings is None:
                qids = self.get_qids(memsize, N, qids=qids, use_weights=use_weights, verbose=verbose)
                out["qids"] = qids
                query_embeddings = self.retriever.data_pool['embedding'][qids]
            qe = query_embeddings.cpu().numpy() if isinstance(query_embeddings, torch.Tensor) else np.asarray(query_embeddings)
            qe = qe.astype(np.float32)
            nns, _ = self.retriever.searcher.search_batched(qe / np.linalg.norm(qe, axis=1)[:, np.newaxis], final_num_neighbors=k_nn)
            retro_cond = torch.from_numpy(np.asarray(self.retriever.data_pool['embedding'][nns])).to(self.device).to(torch.float)
        else:
            retro_cond = nn_embeddings
        _, cond = self.encode_to_c(torch.zeros((N, 0)))
        z_shape = (N, z_dimensionality, code_side_len, code_side_len)
        steps = code_side_len ** 2
        z_start = torch.zeros((N, 0), dtype=torch.long)
        out["samples_with_sampled_nns"] = self.sampling_util(steps, z_start, retro_cond, cond, temperature, top_k, z_shape, **kwargs)
        return out

    # ---- transformer.py:407-430
    def get_qids(self, memsize, N, qids=None, use_weights=False, verbose=False):
        if isinstance(memsize, float):
            assert memsize > 0 and memsize <= 1., 'Require memsize in (0,1]'
            memsize = int(memsize * self.nn_memory.shape[0])
        if qids is None:
            if self.use_memory:
                memsize = min(memsize, self.nn_memory.shape[0])
                nn_mem = self.nn_memory.detach().cpu().numpy()[:memsize]
                ps = None
                if use_weights:
                    freqs = np.asarray([self.id_count[int(id_)] for id_ in nn_mem])
                    ps = freqs / freqs.sum(keepdims=True)
                qids = np.random.choice(nn_mem, size=N, p=ps)
            else:
                qids = np.random.choice(len(self.retriever.data_pool['embedding']), size=N)
        else:
            assert qids.shape[0] == N
        return qids
