"""CPU restatement of the retrieval step and conditioning glue
(oracle — test infrastructure only).

Follows:
  rdm/data/retrieval_dataset/dsetbuilder.py:478-518  search_k_nearest (query normalisation,
        search_batched -> (uint32 idx [B,k] by descending score, f32 dist), gathers)
  rdm/data/retrieval_dataset/dsetbuilder.py:574       searcher built on embedding/||embedding||
  rdm/models/diffusion/ddpm.py:760-777                retro_cond assembly (query first, k-1 nbrs)
  rdm/models/diffusion/ddpm.py:662-686                unconditional conditioning
  rdm/models/diffusion/ddpm.py:847-875                get_qids
  scripts/rdm_sample.py:203-214                       custom_to_np uint8 conversion (truncation)

PARITY UNPINNED w.r.t. the deployed reference: the reference's searcher is ScaNN 1.2.4
(un-vendored, approximate tree+AH for N >= 2e4).  This oracle is the EXACT maximum-inner-
product search that ScaNN approximates (and equals ScaNN's own brute-force mode for N < 2e4,
dsetbuilder.py:590-592).  Definition used by oracle and HIP path alike:
  * database rows  d_i = fp16( x_i / n_i ),  n_i = fp32( sqrt( sum_j x_ij^2 ) ) with the sum in fp64
                   (division in fp32, one RNE rounding to fp16)
  * query          q^  = q / fp32( sqrt( sum_j q_j^2 in fp64 ) )   (division in fp32)
  * score          s_i = sum_j q^_j * d_ij       (exact products, fp64 accumulation)
  * result         top-k by (score descending, index ascending)
"""
import numpy as np


def normalize_db(emb: np.ndarray) -> np.ndarray:
    """Rows -> unit norm in fp32, stored fp16 (what the searcher is built on)."""
    x = emb.astype(np.float32)
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)
    return (x / n).astype(np.float16)


def normalize_queries(q: np.ndarray) -> np.ndarray:
    q = q.astype(np.float32)
    n = np.sqrt((q.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)
    return q / n


def exact_topk(dbn: np.ndarray, qn: np.ndarray, k: int, chunk: int = 262144, f64: bool = False):
    """Exact MIPS top-k. dbn fp16 [N,D] (normalised), qn f32 [B,D] (normalised).
    Returns (idx uint32 [B,k], score f32 [B,k] -- or the fp64 scores themselves with f64=True); ties -> lower index."""
    B = qn.shape[0]
    q64 = qn.astype(np.float64)
    best_s = np.full((B, 0), 0.0)
    best_i = np.zeros((B, 0), dtype=np.int64)
    for s0 in range(0, dbn.shape[0], chunk):
        blk = dbn[s0:s0 + chunk].astype(np.float64)
        sc = q64 @ blk.T                                        # [B, n]
        ids = np.arange(s0, s0 + blk.shape[0], dtype=np.int64)[None].repeat(B, 0)
        cs = np.concatenate([best_s, sc], axis=1)
        ci = np.concatenate([best_i, ids], axis=1)
        order = np.lexsort((ci, -cs), axis=1)[:, :k]            # score desc, then index asc
        best_s = np.take_along_axis(cs, order, 1)
        best_i = np.take_along_axis(ci, order, 1)
    return best_i.astype(np.uint32), (best_s if f64 else best_s.astype(np.float32))


def search_k_nearest(data_pool: dict, dbn: np.ndarray, queries: np.ndarray, k: int):
    """dsetbuilder.py:478-518 with query_embedded=True."""
    qn = normalize_queries(queries)
    nns, dist = exact_topk(dbn, qn, k)
    return {
        "embeddings": data_pool["embedding"][nns],
        "img_ids": data_pool["img_id"][nns],
        "patch_coords": data_pool["patch_coords"][nns],
        "queries": queries,
        "nns": nns,
        "distances": dist,
        "q_embeddings": queries,
    }


def assemble_retro_cond(q_emb: np.ndarray, r_emb: np.ndarray, k_nn: int, omit_query=False, normalize=False,
                        n_reps=None) -> np.ndarray:
    """ddpm.py:760-777 (nn_encoder is None branch, example_maps None). f32 [B,k,512]."""
    q = q_emb.astype(np.float32)
    r = r_emb.astype(np.float32)
    if normalize:
        q = q / np.linalg.norm(q, axis=-1, keepdims=True)
        r = r / np.linalg.norm(r, axis=-1, keepdims=True)
    if omit_query:
        rc = r
    else:
        rc = np.concatenate([q[:, None], r[:, :k_nn - 1]], axis=1)
    if n_reps is not None:
        rc = np.concatenate([rc] * n_reps, axis=1)
    return rc


def unconditional_conditioning(vex: np.ndarray, shape, label, k_nn: int) -> np.ndarray:
    """ddpm.py:662-686 with a label: stack(stack(vex/||vex|| * label, k), bs)."""
    bs = shape[0]
    sig = vex / np.linalg.norm(vex.reshape(-1)) * label
    sig = np.stack([sig] * k_nn, axis=0)
    return np.stack([sig] * bs, axis=0).astype(np.float32)


def get_qids(nn_memory: np.ndarray, memsize, N: int, id_count: dict = None, use_weights=False, rng=np.random):
    """ddpm.py:847-875 (use_memory=True branch)."""
    if isinstance(memsize, float):
        assert 0 < memsize <= 1.0
        memsize = int(memsize * nn_memory.shape[0])
    memsize = min(memsize, nn_memory.shape[0])
    mem = nn_memory[:memsize]
    ps = None
    if use_weights:
        freqs = np.asarray([id_count[int(i)] for i in mem])
        ps = freqs / freqs.sum(keepdims=True)
    return rng.choice(mem, size=N, p=ps)


def custom_to_np_uint8(x: np.ndarray) -> np.ndarray:
    """scripts/rdm_sample.py:203-214: clamp(-1,1) -> (x+1)/2 -> CHW->HWC -> *255 -> astype(uint8)."""
    x = np.clip(x.astype(np.float32), -1.0, 1.0)
    x = (x + np.float32(1.0)) / np.float32(2.0)
    x = np.transpose(x, (0, 2, 3, 1))
    return (np.float32(255.0) * x).astype(np.uint8)


class StreamingTopK:
    """Exact top-k (score descending, index ascending) over a database presented in row chunks — same definition as
    `exact_topk` (which it is checked against in tests/test_oracle_cpu.py), organised for databases that do not fit in
    host memory twice: per chunk only the rows scoring >= the chunk's k-th best are merged (ties included, so the
    tie-break stays exact).  torch CPU ops (multi-threaded) instead of numpy for the fp64 matmul / partition."""

    def __init__(self, qn: np.ndarray, k: int):
        import torch
        self.t = torch
        self.q64 = torch.from_numpy(np.ascontiguousarray(qn)).double()
        self.k = k
        B = qn.shape[0]
        self.best_s = [np.zeros(0) for _ in range(B)]
        self.best_i = [np.zeros(0, dtype=np.int64) for _ in range(B)]

    @staticmethod
    def normalize_chunk(raw):
        """normalize_db for one chunk (torch tensor fp16/fp32 [n,D] on the host) -> fp16 tensor."""
        x = raw.float()
        n = x.double().pow(2).sum(dim=1, keepdim=True).sqrt().float()
        return (x / n).half()

    def push(self, dbn_chunk, row0: int):
        """dbn_chunk: torch fp16 [n,D] normalised rows, global index of its first row = row0."""
        t = self.t
        sc = self.q64 @ dbn_chunk.double().t()                      # [B, n] fp64
        n = sc.shape[1]
        kk = min(self.k, n)
        kth = t.topk(sc, kk, dim=1).values[:, -1:]                   # k-th best of the chunk per query
        mask = sc >= kth
        for b in range(sc.shape[0]):
            cols = t.nonzero(mask[b]).reshape(-1)
            cs = np.concatenate([self.best_s[b], sc[b, cols].numpy()])
            ci = np.concatenate([self.best_i[b], cols.numpy().astype(np.int64) + row0])
            order = np.lexsort((ci, -cs))[:self.k]
            self.best_s[b], self.best_i[b] = cs[order], ci[order]

    def result(self):
        return (np.stack(self.best_i).astype(np.uint32), np.stack(self.best_s).astype(np.float32))


def exact_topk_bulk(dbn: np.ndarray, qn: np.ndarray, k: int, qblock: int = 1024, chunk: int = 131072, slack: int = 8):
    """exact_topk for dataset-scale query batches (same definition; vectorised: fp64 GEMM blocks, per-block top-(k+slack), one
    final (score desc, index asc) sort of the merged candidates).  Exact as long as fewer than `slack` rows tie with a query's
    k-th score inside one chunk (test databases plant at most a few duplicates)."""
    import torch
    B = qn.shape[0]
    out_i = np.zeros((B, k), dtype=np.uint32); out_s = np.zeros((B, k), dtype=np.float32)
    d = torch.from_numpy(np.ascontiguousarray(dbn))
    for q0 in range(0, B, qblock):
        q64 = torch.from_numpy(np.ascontiguousarray(qn[q0:q0 + qblock])).double()
        cs, ci = [], []
        for r0 in range(0, d.shape[0], chunk):
            blk = d[r0:r0 + chunk].double()
            sc = q64 @ blk.t()
            kk = min(k + slack, sc.shape[1])
            v, i = torch.topk(sc, kk, dim=1)
            cs.append(v); ci.append(i + r0)
        cs = torch.cat(cs, dim=1).numpy(); ci = torch.cat(ci, dim=1).numpy()
        order = np.lexsort((ci, -cs), axis=1)[:, :k]
        out_s[q0:q0 + qblock] = np.take_along_axis(cs, order, 1).astype(np.float32)
        out_i[q0:q0 + qblock] = np.take_along_axis(ci, order, 1).astype(np.uint32)
    return out_i, out_s
