"""Native counterpart of the sampling-path methods of rdm/data/retrieval_dataset/dsetbuilder.py::DatasetBuilder:
load_embeddings / load_single_file / load_multi_files (:181-236), train_searcher (:534-619), search_k_nearest
(:478-518), embed (:461-473), and the `.searcher.search_batched / .search` surface of the ScaNN object
(:490, rdm/data/base.py:81).

The searcher is exact brute force on the GPU (the reference's ScaNN tree-AH is approximate): the
database is normalised and held in HBM as fp16 (dsetbuilder.py:574), a batch of queries streams it once.
numpy in / numpy out like the reference.
"""
import glob
import os
import time

import numpy as np
import torch

from ... import _lib


class HipSearcher(object):
    """Drop-in for the scann searcher object."""

    def __init__(self, ctx, dim):
        self.ctx, self.dim = ctx, dim

    def search_batched(self, queries, final_num_neighbors=None, **kw):
        q = torch.as_tensor(np.ascontiguousarray(queries, dtype=np.float32))
        idx, dist = self.ctx.knn(q, int(final_num_neighbors))
        return idx.cpu().numpy().view(np.uint32), dist.cpu().numpy()

    def search_batched_device(self, queries: torch.Tensor, k: int):
        """Device-resident variant (no host round trip): -> (idx int32-bits-of-uint32 [B,k], score f32 [B,k])."""
        return self.ctx.knn(queries, k)

    def search(self, query, final_num_neighbors=None, **kw):
        i, d = self.search_batched(np.asarray(query)[None], final_num_neighbors)
        return i[0], d[0]


class DatasetBuilder(object):
    def __init__(self, saved_embeddings=None, k=20, retriever=None, retriever_config=None, ctx=None, device=0,
                 data_pool=None, load_patch_dataset=False, batch_size=100, max_pool_size=None, **ignored):
        self.k = k
        self.batch_size = batch_size
        self.saved_embeddings = saved_embeddings
        self.load_patch_dataset = load_patch_dataset
        self.visualize = False
        self.retriever = retriever
        self.searcher = None
        self._dev_index = device if isinstance(device, int) else (torch.device(device).index or 0)
        self._ctx = ctx if ctx is not None else (retriever.model.ctx if retriever is not None else None)
        self.data_pool = {'embedding': [], 'img_id': [], 'patch_coords': []}
        if data_pool is not None:
            self.data_pool = {k_: np.asarray(v) for k_, v in data_pool.items()}
        elif saved_embeddings is not None:
            self.load_embeddings(saved_embeddings)

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.Context(self._dev_index)       # lazily: loading the npz shards needs no GPU
        return self._ctx

    # ---- dsetbuilder.py:181-236
    def load_single_file(self, saved_embeddings):
        compressed = np.load(saved_embeddings)
        self.data_pool = {key: compressed[key] for key in compressed.files}
        print('Finished loading of retrieval database of length', self.data_pool['embedding'].shape[0])

    def load_multi_files(self, data_archive):
        out = {key: [] for key in self.data_pool}
        for d in data_archive:
            for key in d.files:
                if key in out:
                    out[key].append(d[key])
        return out

    def load_embeddings(self, saved_embeddings):
        if os.path.isfile(saved_embeddings):
            return self.load_single_file(saved_embeddings)
        # the reference globs in filesystem order (dsetbuilder.py:222); here: deterministic, by the part index of
        # '<rows>x512-part_<i>.npz' (dsetbuilder.py:240-254) when present, else by name
        import re
        def part_key(f):
            m = re.search(r'part_(\d+)', os.path.basename(f))
            return (0, int(m.group(1)), f) if m else (1, 0, f)
        files = sorted(glob.glob(os.path.join(saved_embeddings, '*.npz')), key=part_key)
        assert len(files) > 0, f'No embedding shards (*.npz) under {saved_embeddings}'
        t0 = time.time()
        parts = self.load_multi_files([np.load(f) for f in files])
        self.data_pool = {key: np.concatenate(parts[key], axis=0) for key in parts if len(parts[key]) > 0}
        print(f'Finished loading of patch embeddings ({self.data_pool["embedding"].shape[0]} rows) in {time.time() - t0:.1f} s')

    # ---- dsetbuilder.py:534-619: "training" = upload + normalise on device
    def train_searcher(self, k=None, metric=None, **ignored):
        emb = np.ascontiguousarray(self.data_pool['embedding'])
        if emb.dtype not in (np.float16, np.float32):
            emb = emb.astype(np.float32)
        self.ctx.db_load(emb)
        self.searcher = HipSearcher(self.ctx, emb.shape[1])
        return self.searcher

    # ---- dsetbuilder.py:461-473
    @torch.no_grad()
    def embed(self, batch, is_caption=False):
        if is_caption:
            from ...modules.custom_clip.tokenizer import tokenize
            tokens = torch.from_numpy(tokenize(list(batch), self.retriever.model.cfg.context_length))
            out = self.retriever.model.encode_text(tokens)
            bs = len(batch)
        else:
            batch = torch.as_tensor(batch)
            if batch.ndim == 4 and batch.shape[-1] in (1, 3):
                batch = batch.permute(0, 3, 1, 2)                    # b h w c -> b c h w (dsetbuilder.py:465)
            out = self.retriever(batch)
            bs = batch.shape[0]
        return out.cpu().numpy().reshape(bs, -1)

    # ---- dsetbuilder.py:478-518
    def search_k_nearest(self, queries, k=None, is_caption=False, visualize=None, query_embedded=False):
        assert self.searcher is not None, 'Cannot search with uninitialized searcher'
        if k is None:
            k = self.k
        if not query_embedded:
            query_embeddings_ = self.embed(queries, is_caption=is_caption)
        else:
            query_embeddings_ = queries.cpu().numpy() if isinstance(queries, torch.Tensor) else np.asarray(queries)
        start = time.time()
        nns, distances = self.searcher.search_batched(query_embeddings_, final_num_neighbors=k)   # normalises on device (:487)
        end = time.time()
        out = {'embeddings': self.data_pool['embedding'][nns],
               'img_ids': self.data_pool['img_id'][nns] if len(self.data_pool.get('img_id', [])) else None,
               'patch_coords': self.data_pool['patch_coords'][nns] if len(self.data_pool.get('patch_coords', [])) else None,
               'queries': queries, 'exec_time': end - start, 'nns': nns, 'distances': distances,
               'q_embeddings': query_embeddings_}
        if visualize if visualize is not None else self.visualize:
            raise NotImplementedError("nn_patches visualisation needs the raw OpenImages JPEGs (out of scope, SURVEY.md §2 #8)")
        return out
