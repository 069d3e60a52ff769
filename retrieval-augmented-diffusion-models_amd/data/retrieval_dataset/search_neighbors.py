"""Bulk neighbour pre-computation: native counterpart of `search_nns` in scripts/search_neighbors.py:380-450.

For every query item (an image's patches or a caption) the k nearest database rows are computed and either
  * saved as one pickle per image, `embeddings/{k}_nns-img{id:09d}.p` = {npatches_perside: {'embeddings', 'img_ids', 'patch_coords',
    'nn_ids'}}, with the `nn_paths` index {id: filename} (:409-431, the files rdm.data.base.QueryDataset joins at training time), or
  * counted per database row (:432-438) — the frequency table behind `nn_memory` (consumed by rdm/models/diffusion/ddpm.py:168-176,
    847-875: ids sorted by how often they were retrieved, and `id_count`).
The search itself is `DatasetBuilder.search_k_nearest` -> librdm_hip `rdm_knn`: batches of >= 128 queries take the query-tiled
bulk scan (128 queries per walker, the database streamed once per 256 queries, MFMA-bound; csrc/knn.hip), exact like the online
search.  Query batches may be raw patches / captions (embedded by the CLIP towers on the GPU) or pre-computed embeddings
(`mode='embedded'`, batch key 'embeddings').
"""
import os
import pickle

import numpy as np
import torch


def save_pkl(filepath, save_it, npatches_perside):
    """scripts/search_neighbors.py:355-378: merge into an existing per-image file, overwrite a corrupt one."""
    if os.path.isfile(filepath):
        try:
            with open(filepath, 'rb') as f:
                old_one = pickle.load(f)
            old_one.update({npatches_perside: save_it[npatches_perside]})
            save_it = old_one
        except Exception as e:                                   # corrupt file: rewrite
            print(f'ERROR: {e.__class__.__name__} : ', e)
    with open(filepath, 'wb') as f:
        pickle.dump(save_it, f, protocol=pickle.HIGHEST_PROTOCOL)


def search_nns(dataset_builder, qloader, device=None, mode='img', save=False, npatches_perside=None, base_savedir=None, nn_paths=None,
               corrupts=None, start_id=0, max_its=None, batch_size=None):
    """Same arguments and return values as the reference function.  `qloader` is any iterable of collated batches:
    {'patches': [b,n,h,w,c] in [-1,1]} (mode 'img'), {'caption': list[str]} (mode 'text') or {'embeddings': [b,n,d] / [b,d]}
    (mode 'embedded')."""
    assert dataset_builder.searcher is not None
    dset_batch_size = batch_size if batch_size is not None else getattr(qloader, 'batch_size', None)
    if save:
        assert base_savedir is not None and npatches_perside is not None
        os.makedirs(os.path.join(base_savedir, 'embeddings'), exist_ok=True)
        if nn_paths is None:
            nn_paths = {}
    return_ids = {}
    for i, batch in enumerate(qloader):
        if max_its is not None and i >= max_its:
            break
        if mode == 'img':
            query = torch.as_tensor(batch['patches'])
            b, n = query.shape[:2]
            query = query.reshape((b * n,) + tuple(query.shape[2:]))
            results = dataset_builder.search_k_nearest(query, visualize=False, is_caption=False)
        elif mode == 'text':
            query = list(batch['caption'])
            b, n = len(query), 1
            results = dataset_builder.search_k_nearest(query, visualize=False, is_caption=True)
        else:
            e = np.asarray(batch['embeddings'], dtype=np.float32)
            b, n = (e.shape[0], e.shape[1]) if e.ndim == 3 else (e.shape[0], 1)
            results = dataset_builder.search_k_nearest(e.reshape(b * n, -1), visualize=False, query_embedded=True)
        if dset_batch_size is None:
            dset_batch_size = b
        if save:
            results = {key: results[key].reshape((b, n) + tuple(results[key].shape[1:])) if isinstance(results[key], np.ndarray) else results[key]
                       for key in results}
            for j in range(len(results['embeddings'])):
                idx = start_id + i * dset_batch_size + j
                filename = f'embeddings/{dataset_builder.k}_nns-img{idx:09d}.p'
                save_it = {npatches_perside: {'embeddings': results['embeddings'][j], 'img_ids': results['img_ids'][j],
                                              'patch_coords': results['patch_coords'][j], 'nn_ids': results['nns'][j]}}
                save_pkl(os.path.join(base_savedir, filename), save_it, npatches_perside)
                nn_paths.update({idx: filename})
        else:
            ids, counts = np.unique(results['nns'], return_counts=True)
            for id_, c in zip(ids, counts):
                return_ids[int(id_)] = return_ids.get(int(id_), 0) + int(c)
    return nn_paths if save else return_ids


def build_nn_memory(return_ids, path=None):
    """The pickle behind `nn_memory:` in the model configs (models/rdm/imagenet/config.yaml:22; loaded at
    rdm/models/diffusion/ddpm.py:168-176): database ids ordered by retrieval frequency (most frequent first, ties by id) and the
    `id_count` table; `get_qids` draws pseudo-queries from the top-m of it."""
    ids = np.asarray(sorted(return_ids, key=lambda i: (-return_ids[i], i)), dtype=np.int64)
    data = {'nn_memory': ids, 'id_count': {int(i): int(return_ids[i]) for i in ids}}
    if path is not None:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, 'wb') as f:
            pickle.dump(data, f, protocol=pickle.HIGHEST_PROTOCOL)
    return data
