"""Native counterpart of the sampling-path methods of rdm/data/retrieval_dataset/dsetbuilder.py::DatasetBuilder:
load_embeddings / load_single_file / load_multi_files (:181-236), train_searcher (:534-619), search_k_nearest
(:478-518), embed (:461-473), and the `.searcher.search_batched / .search` surface of the ScaNN object
(:490, rdm/data/base.py:81); plus the database-construction side: build_data_pool (:317-437),
save_datapool (:238-259), reset_data_pool (:261-262) with the CLIP image tower running on the GPU.

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
    """Drop-in for the scann searcher object.  With `row0` / `group` the context holds only rows [row0, row0 + n_local) of the
    database (one shard per rank): every search runs on the local rows, the per-rank lists are merged in one exchange
    (parallel.merge_sharded_topk) and the answer -- global row ids -- is the same on every rank and the same as an unsharded search."""

    def __init__(self, ctx, dim, row0=0, group=None, sharded=False, n_local=None):
        self.ctx, self.dim, self.row0, self.group, self.sharded, self.n_local = ctx, dim, int(row0), group, bool(sharded), n_local

    def _search(self, q: torch.Tensor, k: int):
        if not self.sharded:
            return self.ctx.knn(q, k)
        from ... import parallel
        kl = min(k, self.n_local)
        idx, sc = self.ctx.knn(q, kl, f64=True)
        gid = (idx.to(torch.int64) & 0xffffffff) + self.row0
        if kl < k:                                                  # a shard shorter than k: pad with "no row"
            pad = k - kl
            gid = torch.cat([gid, torch.full((gid.shape[0], pad), 2 ** 62, dtype=torch.int64, device=gid.device)], dim=1)
            sc = torch.cat([sc, torch.full((sc.shape[0], pad), float("-inf"), dtype=torch.float64, device=sc.device)], dim=1)
        gi, gs = parallel.merge_sharded_topk(gid, sc, k, self.group)
        return gi.to(torch.int32), gs.to(torch.float32)             # uint32 bits, like the unsharded call

    def search_batched(self, queries, final_num_neighbors=None, **kw):
        q = torch.as_tensor(np.ascontiguousarray(queries, dtype=np.float32))
        idx, dist = self._search(q, int(final_num_neighbors))
        return idx.cpu().numpy().view(np.uint32), dist.cpu().numpy()

    def search_batched_device(self, queries: torch.Tensor, k: int):
        """Device-resident variant (no host round trip): -> (idx int32-bits-of-uint32 [B,k], score f32 [B,k])."""
        return self._search(queries, k)

    def search(self, query, final_num_neighbors=None, **kw):
        i, d = self.search_batched(np.asarray(query)[None], final_num_neighbors)
        return i[0], d[0]


class DatasetBuilder(object):
    def __init__(self, saved_embeddings=None, k=20, retriever=None, retriever_config=None, ctx=None, device=0,
                 data_pool=None, load_patch_dataset=False, batch_size=100, max_pool_size=None, **ignored):
        self.k = k
        self.out_dir = ignored.get('out_dir')
        self.max_pool_size = max_pool_size
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
    def shard_rows(self, enabled=True, group=None):
        """[native] Hold only this rank's contiguous share of the database rows in HBM (for databases beyond one GPU's memory,
        SURVEY.md 8e); the host-side data_pool stays whole, like the reference's.  Takes effect at the next train_searcher()."""
        self._shard_rows, self._shard_group = bool(enabled), group
        self.searcher = None
        return self

    def train_searcher(self, k=None, metric=None, **ignored):
        emb = self.data_pool['embedding']
        row0, sharded, group = 0, bool(getattr(self, "_shard_rows", False)), getattr(self, "_shard_group", None)
        if sharded:
            from ... import parallel
            world, rank = parallel.world_rank(group)
            row0, row1 = parallel.shard_range(len(emb), world, rank)
            emb = emb[row0:row1]
        emb = np.ascontiguousarray(emb)
        if emb.dtype not in (np.float16, np.float32):
            emb = emb.astype(np.float32)
        self.ctx.db_load(emb)
        self.searcher = HipSearcher(self.ctx, emb.shape[1], row0=row0, group=group, sharded=sharded, n_local=emb.shape[0])
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

    # ---- dsetbuilder.py:261-262
    def reset_data_pool(self):
        self.data_pool = {key: [] for key in self.data_pool}

    # ---- dsetbuilder.py:238-259: '<rows>x<dim>[-<postfix>].npz', compressed, keys embedding / img_id / patch_coords (/ class_id)
    def save_datapool(self, postfix=None, out_dir=None):
        out_dir = out_dir or self.out_dir
        assert out_dir is not None, 'save_datapool needs an output directory (the reference hard-codes its NFS path)'
        pool = {key: np.concatenate([np.asarray(v) for v in self.data_pool[key]]) for key in self.data_pool if len(self.data_pool[key]) > 0}
        identifier = 'x'.join(str(s_) for s_ in pool['embedding'].shape)
        if postfix:
            identifier = identifier + '-' + postfix
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, identifier + '.npz')
        np.savez_compressed(path, **pool)
        self.saved_embeddings = out_dir
        return path

    # ---- dsetbuilder.py:317-437.  `loader` yields the reference's collated batches: {'patch': [b,(n,)h,w,c] in [-1,1],
    # 'img_id': [b(,n)], 'patch_coords': [b(,n),4]} (+ optional 'class_id'); the patches are embedded by the retriever
    # (CLIP image tower on the GPU incl. the bicubic 224x224 preprocessing, retrievers.py:83-95) and written in shards of
    # `chunk_size` rows named 'part_<i>' exactly like the shipped databases (scripts/download_databases.sh:6-15).
    def build_data_pool(self, loader, max_pool_size=None, chunk_size=None, save_embeddings=True, out_dir=None):
        self.out_dir = out_dir or self.out_dir
        max_pool_size = max_pool_size if max_pool_size is not None else (self.max_pool_size or float('inf'))
        self.data_pool = {'embedding': [], 'img_id': [], 'patch_coords': []}
        n_examples, part, files, kept = 0, 1, [], {'embedding': [], 'img_id': [], 'patch_coords': []}

        def flush():
            nonlocal part
            if len(self.data_pool['embedding']) == 0:
                return
            if save_embeddings:
                files.append(self.save_datapool(postfix=f'part_{part}' if chunk_size is not None else None))
            for key in self.data_pool:
                kept.setdefault(key, []).extend(self.data_pool[key])
            self.reset_data_pool()
            part += 1

        for batch in loader:
            if 'patch' not in batch:
                break
            patches = torch.as_tensor(batch['patch'])
            if patches.ndim == 5:                                     # b n h w c -> (b n) h w c (dsetbuilder.py:463-464)
                patches = patches.reshape((-1,) + tuple(patches.shape[2:]))
            emb = self.embed(patches)
            self.data_pool['embedding'].append(emb)
            self.data_pool['img_id'].append(np.asarray(batch['img_id']).reshape(emb.shape[0]))
            self.data_pool['patch_coords'].append(np.asarray(batch['patch_coords']).reshape(emb.shape[0], -1))
            if 'class_id' in batch:
                self.data_pool.setdefault('class_id', []).append(np.asarray(batch['class_id']).reshape(emb.shape[0]))
            n_examples += emb.shape[0]
            if chunk_size is not None and n_examples / chunk_size >= part:
                flush()
            if n_examples >= max_pool_size:
                break
        flush()
        self.data_pool = {key: np.concatenate([np.asarray(v) for v in kept[key]]) for key in kept if len(kept[key]) > 0}
        print(f'Finish extraction of {n_examples} feature embeddings')
        return files

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
