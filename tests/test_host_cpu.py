"""CPU suite for the host-side mirror of the reference surface (no GPU needed): schedules, tokenizer, DB shard
loading, pseudo-query sampling, conditioning glue, world_size-2 sharding over gloo."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rdm_amd  # noqa: F401
from oracle import diffusion as odiff
from oracle import retrieval as oret

from _util import golden

torch.set_grad_enabled(False)


class _DummyModel:
    num_timesteps = 1000
    def __init__(self):
        self.alphas_cumprod = odiff.Schedule().alphas_cumprod
        self.device = torch.device("cpu")


@pytest.mark.parametrize("S,eta", [(50, 0.0), (50, 1.0), (100, 0.3), (250, 1.0), (7, 0.0)])
def test_ddim_sampler_schedule_matches_oracle(S, eta):
    from rdm_amd.models.diffusion.ddim import DDIMSampler
    s = DDIMSampler(_DummyModel())
    s.make_schedule(S, ddim_eta=eta, verbose=False)
    ts, a_t, a_prev, sigma, s1m = odiff.ddim_schedule(odiff.Schedule(), S, eta)
    assert np.array_equal(s.ddim_timesteps, ts)
    assert np.array_equal(np.asarray(s.ddim_alphas, np.float32), a_t.numpy())
    assert np.array_equal(np.asarray(s.ddim_alphas_prev, np.float32), a_prev.numpy())
    assert np.array_equal(np.asarray(s.ddim_sigmas, np.float32), sigma.numpy())
    assert np.array_equal(np.asarray(s.ddim_sqrt_one_minus_alphas, np.float32), s1m.numpy())


def test_ddim_timestep_quirks():
    from rdm_amd.models.diffusion.ddim import make_ddim_timesteps
    assert len(make_ddim_timesteps("uniform", 6, 1000)) == 7           # 1000 // 6 = 166 -> 7 steps, like ldm
    assert np.array_equal(make_ddim_timesteps("uniform", 6, 1000), odiff.make_ddim_timesteps(6))
    with pytest.raises(ValueError):
        make_ddim_timesteps("uniform", 3, 1000)                        # 0,333,666,999 (+1) -> 1000 is out of range


def test_ddim_sampler_rejects_unsupported_options():
    from rdm_amd.models.diffusion.ddim import DDIMSampler
    s = DDIMSampler(_DummyModel())
    s.make_schedule(5, verbose=False)
    with pytest.raises(NotImplementedError):      # dead in the reference too (ddim.py:249 reads a sampler buffer off the model)
        s.ddim_sampling(torch.zeros(1, 4, 512), (1, 3, 8, 8), ddim_use_original_steps=True)
    with pytest.raises(NotImplementedError):
        s.p_sample_ddim(torch.zeros(1, 3, 8, 8), torch.zeros(1, 4, 512), torch.zeros(1, dtype=torch.long), 0, use_original_steps=True)
    with pytest.raises(AssertionError):
        s.sample(5, 1, (3, 8, 8), conditioning=torch.zeros(1, 4, 512), unconditional_guidance_scale=0.5, verbose=False)


class _ToyModel(_DummyModel):
    """apply_model stand-in (a fixed nonlinear map of x, t and the context) for the host-side loop logic."""
    parameterization = "eps"
    def __init__(self):
        super().__init__()
        self.sqrt_ac = self.alphas_cumprod.sqrt(); self.sqrt_1mac = (1.0 - self.alphas_cumprod).sqrt()
    def apply_model(self, x, t, c):
        return torch.tanh(x * 0.7 + c.mean(dim=(1, 2)).reshape(-1, 1, 1, 1)) * (1.0 + t.reshape(-1, 1, 1, 1).float() / 1000.0)
    def q_sample(self, x0, t, noise=None):
        noise = torch.zeros_like(x0) if noise is None else noise
        return self.sqrt_ac[t].reshape(-1, 1, 1, 1) * x0 + self.sqrt_1mac[t].reshape(-1, 1, 1, 1) * noise


@pytest.mark.parametrize("case", ["callback", "mask", "style_content", "subset", "corrector", "all"])
def test_ddim_per_step_options_match_oracle(case):
    """ddim.py:143-209 loop-body options (inpainting mask, style / content conditioning by SNR band, timestep subset, score
    corrector, callbacks) through the mirror's per-step path == the oracle's restatement, bit for bit on CPU."""
    from rdm_amd.models.diffusion.ddim import DDIMSampler
    g = torch.Generator().manual_seed(5)
    B, S, eta, scale = 3, 20, 0.4, 2.0
    m = _ToyModel()
    x_T = torch.randn(B, 3, 8, 8, generator=g)
    c = torch.randn(B, 4, 16, generator=g); uc = torch.zeros(B, 4, 16)
    cs = torch.randn(B, 4, 16, generator=g); cc = torch.randn(B, 4, 16, generator=g)
    noise = torch.randn(S, B, 3, 8, 8, generator=g); qn = torch.randn(S, B, 3, 8, 8, generator=g)
    x0 = torch.randn(B, 3, 8, 8, generator=g); mask = (torch.rand(B, 1, 8, 8, generator=g) > 0.5).float()
    kw, okw = {}, {}
    seen = []
    if case in ("callback", "all"):
        kw.update(img_callback=lambda px0, i: seen.append(i))
    if case in ("mask", "all"):
        kw.update(mask=mask, x0=x0, q_noise=qn); okw.update(mask=mask, x0=x0, q_noise=qn)
    if case in ("style_content", "all"):
        kw.update(style_cond=[cs], content_cond={"c_crossattn": cc}); okw.update(style_cond=cs, content_cond=cc)
    if case in ("subset", "all"):
        kw.update(timesteps=12); okw.update(timesteps=12)
    if case in ("corrector", "all"):
        class Corr:
            def modify_score(self, model, e_t, x, t, c, gain=1.0):
                return e_t * gain + 0.1 * x
        kw.update(score_corrector=Corr(), corrector_kwargs={"gain": 0.9})
        okw.update(score_corrector=lambda e, x, t, c_: e * 0.9 + 0.1 * x)
    smp = DDIMSampler(m)
    if "timesteps" in kw:        # like the reference, `sample` does not forward a timestep subset: ddim_sampling is the entry for it
        smp.make_schedule(S, ddim_eta=eta, verbose=False)
        z, inter = smp.ddim_sampling([c], (B, 3, 8, 8), x_T=x_T, unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                     log_every_t=5, noise=noise, S=S, eta=eta, **kw)
    else:
        z, inter = smp.sample(S, B, (3, 8, 8), conditioning=[c], eta=eta, x_T=x_T, unconditional_guidance_scale=scale,
                              unconditional_conditioning=uc, log_every_t=5, verbose=False, noise=noise, **kw)
    rz, rinter = odiff.ddim_sample(m.apply_model, odiff.Schedule(), S, x_T, c, eta=eta, scale=scale, uncond=uc, noise=noise,
                                   log_every_t=5, **okw)
    assert torch.equal(z, rz)
    assert len(inter["x_inter"]) == len(rinter["x_inter"])
    for a, b_ in zip(inter["pred_x0"], rinter["pred_x0"]):
        assert torch.equal(a, b_)
    if case in ("callback", "all"):
        assert seen == list(range(11 if case == "all" else 20))


def test_ddim_quantize_x0_matches_oracle():
    """quantize_x0=True (ddim.py:260-261: pred_x0 snapped to the first stage's codebook every step) through the mirror's per-step
    path, with a stand-in quantiser on the toy model, == the oracle's restatement bit for bit."""
    from rdm_amd.models.diffusion.ddim import DDIMSampler
    g = torch.Generator().manual_seed(9)
    B, S = 2, 10
    m = _ToyModel()
    book = torch.randn(32, 3, generator=g)

    def quant(z):                                     # nearest codebook entry, straight-through form (taming VectorQuantizer2)
        zf = z.permute(0, 2, 3, 1)
        idx = ((zf[..., None, :] - book) ** 2).sum(-1).argmin(-1)
        return (zf + (book[idx] - zf)).permute(0, 3, 1, 2).contiguous()
    m.quantize_first_stage = quant
    x_T = torch.randn(B, 3, 8, 8, generator=g); c = torch.randn(B, 4, 16, generator=g); uc = torch.zeros(B, 4, 16)
    z, inter = DDIMSampler(m).sample(S, B, (3, 8, 8), conditioning=c, eta=0.0, x_T=x_T, unconditional_guidance_scale=1.5,
                                     unconditional_conditioning=uc, quantize_x0=True, log_every_t=2, verbose=False)
    rz, rinter = odiff.ddim_sample(m.apply_model, odiff.Schedule(), S, x_T, c, scale=1.5, uncond=uc, log_every_t=2, quantize=quant)
    assert torch.equal(z, rz)
    for a, b_ in zip(inter["pred_x0"][1:], rinter["pred_x0"][1:]):
        assert torch.equal(a, b_) and (a - quant(a)).abs().max().item() <= 1e-5        # every stored pred_x0 sits ON the codebook (z + (e - z): a few ulps off e)


def test_tokenizer_matches_reference_golden():
    from rdm_amd.modules.custom_clip.tokenizer import tokenize
    g = golden("tokenizer.npz")
    t = tokenize([str(c) for c in g["captions"]])
    assert t.dtype == np.int64 and t.shape == g["tokens"].shape
    assert np.array_equal(t, g["tokens"])
    long = tokenize(["word " * 200])
    assert long.shape == (1, 77) and long[0, 0] == 49406 and (long[0] != 0).all()      # truncated like the reference


def test_schedule_buffers_and_get_qids():
    from rdm_amd.models.diffusion.ddpm import MinimalRETRODiffusion
    mem = np.arange(1000, 2000)
    counts = {int(i): (int(i) % 7) + 1 for i in mem}
    m = MinimalRETRODiffusion(unet_config={"params": {}}, nn_memory=mem, id_count=counts)      # no GPU touched
    o = odiff.Schedule()
    for n in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "posterior_mean_coef1", "posterior_mean_coef2",
              "posterior_log_variance_clipped", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        assert torch.equal(getattr(m, n), getattr(o, n)), n
    for memsize, weights in ((0.01, False), (0.5, True), (100, False)):
        np.random.seed(123); got = m.get_qids(memsize, 16, use_weights=weights)
        np.random.seed(123); ref = oret.get_qids(mem, memsize, 16, id_count=counts, use_weights=weights)
        assert np.array_equal(got, ref)
    # unconditional conditioning: label 0.0 -> exact zeros [B,k,512] (ddpm.py:673-680)
    m.unconditional_guidance_vex = torch.randn(512)
    uc = m.get_unconditional_conditioning((3, 4, 512), unconditional_guidance_label=0., k_nn=4)
    assert uc.shape == (3, 4, 512) and not uc.any()


def test_dataset_builder_loads_npz_shards(tmp_path):
    from rdm_amd.data.retrieval_dataset.dsetbuilder import DatasetBuilder
    rng = np.random.default_rng(0)
    rows = [5, 7, 3]
    for i, r in enumerate(rows):       # <rows>x512-part_<i>.npz (dsetbuilder.py:240-254)
        np.savez_compressed(tmp_path / f"{r}x512-part_{i}.npz", embedding=rng.standard_normal((r, 512)).astype(np.float16),
                            img_id=np.arange(r) + 100 * i, patch_coords=rng.integers(0, 256, (r, 4)))
    db = DatasetBuilder(saved_embeddings=str(tmp_path))
    assert db.data_pool["embedding"].shape == (15, 512) and db.data_pool["embedding"].dtype == np.float16
    assert db.data_pool["img_id"].tolist()[:6] == [0, 1, 2, 3, 4, 100]
    assert db.searcher is None
    with pytest.raises(AssertionError):
        db.search_k_nearest(np.zeros((1, 512), np.float32), k=2, query_embedded=True)


def test_dataset_builder_builds_and_reloads_shards(tmp_path):
    """build_data_pool / save_datapool (dsetbuilder.py:317-437, 238-259): shard naming '<rows>x<dim>-part_<i>.npz', keys,
    chunking by rows, and the round trip through load_embeddings. The retriever is a stand-in (host logic only)."""
    from rdm_amd.data.retrieval_dataset.dsetbuilder import DatasetBuilder

    class FakeRetriever:
        class model:
            ctx = None
        def __call__(self, x):                       # [b,3,h,w] -> [b,16]: deterministic, depends on the pixels
            return x.float().reshape(x.shape[0], -1)[:, :16] * 2.0

    rng = np.random.default_rng(1)
    batches = [{"patch": rng.uniform(-1, 1, (4, 2, 8, 8, 3)).astype(np.float32), "img_id": np.arange(8).reshape(4, 2) + 10 * i,
                "patch_coords": rng.integers(0, 64, (4, 2, 4))} for i in range(3)]
    db = DatasetBuilder(retriever=FakeRetriever(), out_dir=str(tmp_path))
    files = db.build_data_pool(iter(batches), chunk_size=10)
    assert [os.path.basename(f) for f in files] == ["16x16-part_1.npz", "8x16-part_2.npz"]
    assert db.data_pool["embedding"].shape == (24, 16) and db.data_pool["patch_coords"].shape == (24, 4)
    want = np.concatenate([b["patch"].reshape(8, 8, 8, 3).transpose(0, 3, 1, 2).reshape(8, -1)[:, :16] * 2.0 for b in batches])
    assert np.allclose(db.data_pool["embedding"], want)
    again = DatasetBuilder(saved_embeddings=str(tmp_path))
    assert np.array_equal(again.data_pool["embedding"], db.data_pool["embedding"])
    assert again.data_pool["img_id"].tolist() == db.data_pool["img_id"].tolist() == sum([(np.arange(8) + 10 * i).tolist() for i in range(3)], [])
    # a single un-chunked file, capped by max_pool_size
    db2 = DatasetBuilder(retriever=FakeRetriever(), out_dir=str(tmp_path / "one"))
    files2 = db2.build_data_pool(iter(batches), max_pool_size=16)
    assert [os.path.basename(f) for f in files2] == ["16x16.npz"]


def test_util_helpers():
    from rdm_amd.util import convert_nn_tree, ischannellastimage
    assert ischannellastimage(np.zeros((2, 8, 8, 3))) and not ischannellastimage(np.zeros((2, 3, 8, 8)))
    t = convert_nn_tree({"a": np.array([1, 2], dtype=np.uint32), "b": {"c": np.array([3], dtype=np.uint32)}})
    assert t["a"].dtype == np.int32 and t["b"]["c"].dtype == np.int32


# ---- N > 1 path on CPU: world_size 2 over gloo, through the REAL host glue (MinimalRETRODiffusion.set_distributed ->
# sample_with_query / sample_from_rdata -> DatasetBuilder.search_k_nearest -> DDIMSampler / p_sample_loop -> decode ->
# all_gather_images) around a stand-in for the library context (row-wise deterministic arithmetic on the CPU)
class FakeCtx:
    device = torch.device("cpu")

    def db_load(self, emb):
        self.db = torch.as_tensor(np.asarray(emb, dtype=np.float32))

    def knn(self, q, k):
        sc = torch.as_tensor(q).float() @ self.db.t()
        v, i = torch.sort(sc, dim=1, descending=True, stable=True)
        return i[:, :k].to(torch.int32), v[:, :k]

    def ddim_sample(self, S, x_T, cond, uncond, alphas_cumprod, eta=0.0, scale=1.0, noise=None, log_every_t=100, temperature=1.0,
                    want_intermediates=False):
        z = x_T * 0.5 + cond.sum(dim=(1, 2))[:, None, None, None] * 0.01 * scale
        if noise is not None:
            z = z + eta * noise.sum(dim=0) * 0.1
        return z, z[None], z[None]

    def ddpm_sample(self, timesteps, x_T, cond, noise, sched, clip_denoised=True, temperature=1.0):
        return x_T * 0.25 + cond.mean(dim=(1, 2))[:, None, None, None] + noise.sum(dim=0) * 0.01

    def vq_decode(self, z, force_not_quantize=False, return_indices=False):
        return z.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) + 1.0


def _dist_model():
    from rdm_amd.data.retrieval_dataset.dsetbuilder import DatasetBuilder
    from rdm_amd.models.diffusion.ddpm import MinimalRETRODiffusion
    ctx = FakeCtx()
    rng = np.random.default_rng(3)
    n = 300
    pool = {"embedding": rng.standard_normal((n, 512)).astype(np.float16), "img_id": np.arange(n), "patch_coords": np.zeros((n, 4), np.int64)}
    m = MinimalRETRODiffusion(unet_config={"params": {}}, ctx=ctx, image_size=8, nn_memory=np.arange(100), timesteps=1000)
    m.retriever = DatasetBuilder(data_pool=pool, ctx=ctx)
    m.set_distributed(True)
    return m


def _dist_run(n_total):
    m = _dist_model()
    out = []
    q = (np.random.default_rng(9).standard_normal((n_total, 512)) * 0.45).astype(np.float32)
    for kw in (dict(ddim=True, ddim_steps=10), dict(ddim=True, ddim_steps=10, eta=1.0), dict(ddim=False, ddim_steps=None, timesteps=5)):
        torch.manual_seed(5); np.random.seed(5)
        out.append(m.sample_with_query(query=torch.from_numpy(q), query_embedded=True, k_nn=4, unconditional_guidance_scale=2.0,
                                       unconditional_retro_guidance_label=0., **kw)["query_samples"].numpy())
        out.append(m.sample_from_rdata(n_total, k_nn=4, memsize=50, unconditional_guidance_scale=2.0,
                                       unconditional_retro_guidance_label=0., **kw)["samples_with_sampled_nns"].numpy())
    return out


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from rdm_amd import parallel
    assert parallel.init_distributed("gloo") == (rank, rank)
    assert parallel.attach_library_comm(FakeCtx()) is False           # the library's RCCL communicator is for RCCL groups only
    out = _dist_run(n_total)
    if rank == 0:
        q.put(out)
    parallel.shutdown()


@pytest.mark.parametrize("n_total", [8, 7])
def test_batch_sharding_world2_gloo(n_total):
    from rdm_amd.parallel import shard_range
    assert shard_range(7, 2, 0) == (0, 4) and shard_range(7, 2, 1) == (4, 7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + n_total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs: p.start()
    got = q.get(timeout=180)
    for p in procs: p.join(timeout=60)
    ref = _dist_run(n_total)                                      # the 1-rank result (no process group: world 1)
    assert len(got) == len(ref) == 6
    for a, b in zip(got, ref):
        assert a.shape == (n_total, 3, 16, 16)
        assert np.array_equal(a, b)                               # sharding is bit-invariant


class _MockCommCtx:
    """Stands in for a library context in the hand-shake of parallel.attach_library_comm: the 'communicator' is the gloo group itself."""
    device = "cpu"

    def __init__(self, mode, rank, grp):
        self.mode, self.rank, self.comm_world, self.destroyed, self.grp = mode, rank, 0, False, grp     # grp: the mock communicator's OWN group

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, uid, rank, world):
        assert uid == bytes(range(128)) and rank == self.rank
        if self.mode == "hang" and rank == 1:
            import time
            time.sleep(8)                                    # a rendezvous that does not come back within the deadline (3 s here)
        if self.mode == "forever" and rank == 1:
            import threading
            threading.Event().wait()                         # a rendezvous that NEVER comes back
        if self.mode == "raise" and rank == 0:
            raise RuntimeError("no communicator")
        self.comm_world = world

    def comm_all_gather(self, local, world):
        import torch.distributed as dist
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=self.grp)
        out = torch.stack(parts)
        if self.mode == "wrong" and self.rank == 1:
            out = out + 1.0                                  # a gather that answers, wrongly, on one rank only
        return out

    def comm_destroy(self):
        self.destroyed = True; self.comm_world = 0


class _MockProductCtx:
    """The PRODUCT context of the hand-shake: hands out a dedicated sibling for the communicator (as _lib.Context.new_comm_context does) and
    must never be asked for communicator work itself (advisor, round 5: the helper thread must not share the product context)."""
    device = "cpu"

    def __init__(self, mode, rank, grp):
        self.args, self.sibling = (mode, rank, grp), None

    def new_comm_context(self):
        self.sibling = _MockCommCtx(*self.args)
        return self.sibling

    def comm_unique_id(self): raise AssertionError("the product context was asked for communicator work")
    comm_init = comm_all_gather = comm_destroy = comm_unique_id


def _worker_lib_comm_never_returns(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      RDM_LIB_COMM_TIMEOUT="1")
    import time
    from rdm_amd import parallel
    import torch.distributed as dist
    parallel.init_distributed("gloo")
    ctx = _MockProductCtx("forever", rank, dist.new_group(backend="gloo"))
    real = dist.get_backend
    dist.get_backend = lambda g=None: "nccl"
    t0 = time.time()
    try:
        got = parallel.attach_library_comm(ctx)
    finally:
        dist.get_backend = real
    t_attach = time.time() - t0
    # every rank is on the torch.distributed collective: the gather still works, through torch
    out = parallel.all_gather_images(torch.full((2, 3), float(rank)), 4, ctx=ctx)
    t0 = time.time()
    parallel.shutdown()
    q.put((rank, got, getattr(ctx, "lib_comm_agreed", None), getattr(ctx, "lib_comm", "unset"), len(parallel._abandoned_comm_contexts), t_attach,
           time.time() - t0, out[:, 0].tolist()))


def test_library_comm_rendezvous_that_never_returns_world2_gloo():
    """Verdict round 5, item 7: `comm_init` NEVER returns on one rank (RDM_LIB_COMM_TIMEOUT=1).  Every rank must come out on the
    torch.distributed collective within the deadline, the stuck communicator context is abandoned (kept, not closed, not used), the product
    context is never used for communicator work, the gather still works through torch, shutdown() returns and the processes exit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_lib_comm_never_returns, args=(r, 2, 29633, q)) for r in range(2)]
    for p in procs: p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=120), q.get(timeout=120))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        ok, agreed, lib_comm, abandoned, t_attach, t_shutdown, col = got[rank]
        assert ok is False and agreed == 0 and lib_comm is None
        assert abandoned == 1               # rank 1's helper is still inside the rendezvous, rank 0's inside the probe gather that waits for it
        assert t_attach < 20 and t_shutdown < 20
        assert col == [0.0, 0.0, 1.0, 1.0]


def _worker_lib_comm(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      RDM_LIB_COMM_TIMEOUT="3")
    from rdm_amd import parallel
    import torch.distributed as dist
    parallel.init_distributed("gloo")
    real = dist.get_backend
    res = {}
    for mode in ("ok", "raise", "wrong", "hang"):
        # (a real library communicator is independent of the torch group; so is the mock's: one fresh gloo group per case, so that a helper
        #  thread still blocked in its probe cannot interleave with the hand-shake's own collectives)
        ctx = _MockProductCtx(mode, rank, dist.new_group(backend="gloo"))
        dist.get_backend = lambda g=None: "nccl"             # the hand-shake is for RCCL groups; its own traffic here is CPU tensors over gloo
        try:
            got = parallel.attach_library_comm(ctx)
            again = parallel.attach_library_comm(ctx) if got else None
        finally:
            dist.get_backend = real
        res[mode] = (got, again, getattr(ctx, "lib_comm_agreed", None))
        if mode == "ok":                                    # ... and the gather then goes through the context
            assert ctx.lib_comm is ctx.sibling
            res["gather"] = ctx.lib_comm.comm_all_gather(torch.full((2, 3), float(rank)), world).reshape(4, 3)[:, 0].tolist()
    import time
    time.sleep(7)                                            # let the "hang" case's helper threads run out before the group goes away
    q.put((rank, res))
    parallel.shutdown()


def test_library_comm_handshake_agreement_world2_gloo():
    """parallel.attach_library_comm on a mock context over a gloo world of 2 (the real thing needs two GPUs): all ranks must come out with
    the SAME answer -- True when every rank's rendezvous + known-answer gather succeeded; False on every rank when one rank raises, when
    one rank's probe gather answers wrongly, and when one rank's rendezvous does not return within the deadline (RDM_LIB_COMM_TIMEOUT)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_lib_comm, args=(r, 2, 29631, q)) for r in range(2)]
    for p in procs: p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs: p.join(timeout=60)
    for rank in (0, 1):
        r = got[rank]
        assert r["ok"] == (True, True, 2), r
        assert r["gather"] == [0.0, 0.0, 1.0, 1.0]
        for mode in ("raise", "wrong", "hang"):
            assert r[mode][0] is False and r[mode][2] == 0, (rank, mode, r[mode])


def _worker_unseeded(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from rdm_amd import parallel
    parallel.init_distributed("gloo")
    m = _dist_model()
    np.random.seed(100 + rank); torch.manual_seed(7 + rank)          # what an un-seeded run looks like: every process has its own streams
    qids = m.get_qids(50, 6)
    out = m.sample_from_rdata(6, k_nn=4, memsize=50, unconditional_guidance_scale=2.0, unconditional_retro_guidance_label=0.,
                              ddim=True, ddim_steps=10)["samples_with_sampled_nns"].numpy()
    q.put((rank, np.asarray(qids), out))
    parallel.shutdown()


def test_unseeded_ranks_share_rank0_pseudo_queries():
    """Advisor finding (round 2): pseudo-query ids come from numpy's process-local generator; without a seed every rank drew its own,
    and a row-sharded search merged the top-k lists of DIFFERENT queries.  get_qids now broadcasts rank 0's draw: both ranks hold the
    same ids and the sharded result equals the 1-rank run that uses rank 0's streams."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_unseeded, args=(r, 2, 29677, q)) for r in range(2)]
    for p in procs: p.start()
    got = dict((r, (a, b)) for r, a, b in (q.get(timeout=180), q.get(timeout=180)))
    for p in procs: p.join(timeout=60)
    assert np.array_equal(got[0][0], got[1][0])
    assert np.array_equal(got[0][1], got[1][1])
    m = _dist_model()
    np.random.seed(100); torch.manual_seed(7)
    qids = m.get_qids(50, 6)
    ref = m.sample_from_rdata(6, k_nn=4, memsize=50, unconditional_guidance_scale=2.0, unconditional_retro_guidance_label=0.,
                              ddim=True, ddim_steps=10)["samples_with_sampled_nns"].numpy()
    assert np.array_equal(np.asarray(qids), got[0][0]) and np.array_equal(ref, got[0][1])


def test_set_distributed_rejects_tiny_batches():
    m = _dist_model()
    m._shard(1)                                                   # world 1: fine
    with pytest.raises(Exception):
        m.sample_with_query(query=torch.zeros(0, 512), query_embedded=True)


# ---- scripts/rdm_sample.py: the reference's CLI surface
def _script():
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "rdm_sample.py")
    spec = importlib.util.spec_from_file_location("rdm_sample_native", path)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def test_rdm_sample_flags_match_reference():
    """Every flag of the reference parser (scripts/rdm_sample.py:22-143; table extracted by tools/gen_golden.py) exists with the
    same option strings, type, action and default."""
    import json
    from pathlib import Path
    mod = _script()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdm_sample_flags.json")) as f:
        ref = json.load(f)
    acts = {tuple(a.option_strings): a for a in mod.build_parser()._actions}
    for e in ref:
        a = acts.get(tuple(e["options"]))
        assert a is not None, f"missing flag {e['options']}"
        if e["action"] == "store_true":
            assert a.const is True and a.default is False and a.nargs == 0
        else:
            assert a.type is {"int": int, "float": float, "str": str, "Path": Path}[e["type"]], e
            assert a.default == e["default"] or str(a.default) == str(e["default"]), e
    opt = mod.parse_args([])
    assert (opt.batch_size, opt.n_runs, opt.guidance_scale, opt.top_m, opt.k_nn, opt.steps, opt.gpu) == (4, 2, 2.0, 0.01, 4, 100, -1)
    assert mod.parse_args(["--top_m", "500"]).top_m == 500 and isinstance(mod.parse_args(["--top_m", "500"]).top_m, int)
    assert str(mod.parse_args(["-s", "x/y"]).savepath) == "x/y"
    mod.parse_args(["--seed", "3"])                               # the reference crashes here (opt.r_runs, :141)


def test_rdm_sample_documented_deviations():
    """The script's docstring lists its deviations from the reference: --gpu -1 (the reference's CPU default) is refused because
    there is no CPU path, --save_nns needs the raw OpenImages patches."""
    mod = _script()
    with pytest.raises(NotImplementedError):
        mod.load_model(mod.parse_args(["--save_nns", "--gpu", "0", "--synthetic"]))
    with pytest.raises(SystemExit) as e:
        mod.load_model(mod.parse_args(["--synthetic"]))            # --gpu defaults to -1
    assert "gpu" in str(e.value).lower()


def test_rdm_sample_run_loops_follow_reference(tmp_path, monkeypatch):
    """Run-loop semantics of scripts/rdm_sample.py:225-315 on a stand-in model: per-run seeding, k_nn = 1 with --only_caption,
    omit_query masked by only_caption, no eta override, --increase_guidance, --keep_qids, file naming, uint8 truncation."""
    mod = _script()
    calls = []

    class Clip:
        def encode_text(self, tokens):
            return torch.ones(tokens.shape[0], 512)

    class Model:
        device = torch.device("cpu")
        class retriever:
            class retriever:
                model = Clip()
        def get_qids(self, top_m, n, use_weights=False):
            return np.arange(n)
        def sample_with_query(self, **kw):
            calls.append(("q", kw, float(torch.rand(1)), float(np.random.rand())))
            return {"query_samples": torch.full((kw["query"].shape[0], 3, 4, 4), 0.999)}
        def sample_from_rdata(self, n, **kw):
            calls.append(("r", kw, float(torch.rand(1)), float(np.random.rand())))
            return {"samples_with_sampled_nns": torch.full((n, 3, 4, 4), -0.5)}

    opt = mod.parse_args(["-s", str(tmp_path), "-c", "a dog", "-bs", "3", "-n", "2", "--seed", "7", "--only_caption", "--omit_query",
                          "--increase_guidance", "--steps", "9"])
    stamp = mod.sample_conditional(Model(), opt)
    assert [c[0] for c in calls] == ["q", "q"]
    kw0, kw1 = calls[0][1], calls[1][1]
    assert kw0["k_nn"] == 1 and kw0["omit_query"] is False and kw0["query_embedded"] and kw0["ddim"] and kw0["ddim_steps"] == 9
    assert "eta" not in kw0 and kw0["unconditional_retro_guidance_label"] == 0.
    assert kw0["unconditional_guidance_scale"] == 2.0 and kw1["unconditional_guidance_scale"] == 3.0
    assert calls[0][2:] == calls[1][2:]                           # seed_everything before EVERY run
    files = sorted(os.listdir(tmp_path))
    assert files == sorted(f"{stamp}-query_samples-run{n}-sample{i}.png" for n in range(2) for i in range(3))
    from PIL import Image
    px = np.asarray(Image.open(tmp_path / files[0]))
    assert px.shape == (4, 4, 3) and (px == int(255 * ((0.999 + 1.) / 2.))).all()      # truncation, not rounding (:203-214)
    calls.clear()
    opt = mod.parse_args(["-s", str(tmp_path / "u"), "-bs", "2", "-n", "1", "--keep_qids", "--top_m", "50", "--use_weights"])
    (tmp_path / "u").mkdir()
    mod.sample_unconditional(Model(), opt)
    kw = calls[0][1]
    assert calls[0][0] == "r" and np.array_equal(kw["qids"], np.arange(2)) and kw["memsize"] == 50 and kw["use_weights"] and kw["k_nn"] == 4
    assert len(os.listdir(tmp_path / "u")) == 2


def test_synthetic_weights_match_oracle_recipe():
    """rdm_amd.synthetic (product side, used by bench.py / --synthetic) draws exactly the tensors the oracle's recipe draws, so the
    committed golden fixtures apply to it."""
    from oracle import clip as oclip, unet as ounet, vqdecoder as ovq
    from rdm_amd import _lib, synthetic
    assert synthetic.unet_param_shapes(_lib.make_unet_cfg()) == ounet.param_shapes(ounet.shipped_spec())
    assert synthetic.vq_param_shapes(_lib.make_vq_cfg()) == ovq.vq_param_shapes(ovq.shipped_vq_spec())
    assert synthetic.clip_param_shapes(_lib.make_clip_cfg()) == oclip.clip_param_shapes(oclip.vitb32_spec())
    from oracle import rarm as orarm
    assert synthetic.rarm_param_shapes(_lib.make_rarm_cfg()) == orarm.rarm_param_shapes(orarm.shipped_rarm_spec())
    assert synthetic.vq_param_shapes(_lib.make_vqgan_f16_cfg()) == ovq.vq_param_shapes(ovq.vqgan_f16_spec())
    t = ounet.tiny_spec()
    cfg = _lib.make_unet_cfg(model_channels=t.model_channels, num_res_blocks=t.num_res_blocks, attention_resolutions=t.attention_resolutions,
                             channel_mult=t.channel_mult)
    a, b = synthetic.unet_state_dict(cfg, 1234), ounet.synth_state_dict(ounet.param_shapes(t), 1234)
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def test_search_nns_files_and_nn_memory(tmp_path):
    """Bulk neighbour pre-computation (scripts/search_neighbors.py:380-450): per-image pickle format, nn_paths index, frequency
    counting and the nn_memory pickle — host logic around a stand-in searcher (no GPU)."""
    import pickle
    from rdm_amd.data.retrieval_dataset.dsetbuilder import DatasetBuilder
    from rdm_amd.data.retrieval_dataset.search_neighbors import build_nn_memory, search_nns
    rng = np.random.default_rng(0)
    pool = {"embedding": rng.standard_normal((50, 16)).astype(np.float16), "img_id": np.arange(50) * 2, "patch_coords": rng.integers(0, 9, (50, 4))}
    db = DatasetBuilder(data_pool=pool, k=3)

    class Searcher:
        def search_batched(self, q, final_num_neighbors=None):
            sc = np.asarray(q, np.float32) @ pool["embedding"].astype(np.float32).T
            idx = np.argsort(-sc, axis=1, kind="stable")[:, :final_num_neighbors]
            return idx.astype(np.uint32), np.take_along_axis(sc, idx, 1)
    db.searcher = Searcher()
    batches = [{"embeddings": rng.standard_normal((4, 2, 16)).astype(np.float32)} for _ in range(3)]     # 4 images x 2 patches per batch
    paths = search_nns(db, batches, mode="embedded", save=True, npatches_perside=2, base_savedir=str(tmp_path), start_id=100, batch_size=4)
    assert sorted(paths) == list(range(100, 112)) and paths[105] == "embeddings/3_nns-img000000105.p"
    with open(tmp_path / paths[105], "rb") as f:
        d = pickle.load(f)
    assert set(d) == {2} and set(d[2]) == {"embeddings", "img_ids", "patch_coords", "nn_ids"}
    assert d[2]["embeddings"].shape == (2, 3, 16) and d[2]["nn_ids"].shape == (2, 3) and d[2]["patch_coords"].shape == (2, 3, 4)
    want, _ = Searcher().search_batched(batches[1]["embeddings"][1], 3)
    assert np.array_equal(d[2]["nn_ids"], want) and np.array_equal(d[2]["img_ids"], pool["img_id"][want])
    # a second pass with another patch grid merges into the same files
    search_nns(db, [{"embeddings": b["embeddings"][:, :1]} for b in batches], mode="embedded", save=True, npatches_perside=1,
               base_savedir=str(tmp_path), start_id=100, batch_size=4)
    with open(tmp_path / paths[105], "rb") as f:
        assert set(pickle.load(f)) == {1, 2}
    counts = search_nns(db, batches, mode="embedded", save=False)
    assert sum(counts.values()) == 3 * 4 * 2 * 3
    mem = build_nn_memory(counts, str(tmp_path / "nn_memory" / "m.p"))
    assert list(mem["nn_memory"]) == sorted(counts, key=lambda i: (-counts[i], i)) and mem["id_count"] == counts
    with open(tmp_path / "nn_memory" / "m.p", "rb") as f:
        again = pickle.load(f)
    assert np.array_equal(again["nn_memory"], mem["nn_memory"])


def _merge_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from rdm_amd import parallel
    parallel.init_distributed("gloo")
    q.put((rank, _sharded_search(rank, world)))
    parallel.shutdown()


def _sharded_db():
    rng = np.random.default_rng(77)
    db = (rng.standard_normal((1003, 64)) * 0.45).astype(np.float16)
    db[900] = db[17]; db[333] = db[17]                     # one row three times: ties across shards -> global index order
    db[1002] = db[5]
    qs = (rng.standard_normal((9, 64)) * 0.45).astype(np.float32)
    qs[0] = db[17].astype(np.float32); qs[1] = db[5].astype(np.float32)
    return db, qs


def _sharded_search(rank, world, k=7):
    """Every rank: exact fp64 top-k of ITS rows (what rdm_knn_f64 returns), then the product's merge."""
    from rdm_amd import parallel
    db, qs = _sharded_db()
    r0, r1 = parallel.shard_range(len(db), world, rank)
    dbn, qn = oret.normalize_db(db), oret.normalize_queries(qs)
    li, ls = oret.exact_topk(dbn[r0:r1], qn, min(k, r1 - r0), f64=True)
    gi, gs = parallel.merge_sharded_topk(torch.from_numpy(li.astype(np.int64)) + r0, torch.from_numpy(ls.astype(np.float64)), k)
    return gi.numpy(), gs.numpy()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_row_sharded_topk_merge_equals_unsharded(world):
    """SURVEY 8e alternative: database rows sharded over ranks, one all-gather of (index, fp64 score) pairs, merge under the single
    total order (score desc, global index asc) == the exact search over the whole database, ties across shards included."""
    db, qs = _sharded_db()
    ref_i, ref_s = oret.exact_topk(oret.normalize_db(db), oret.normalize_queries(qs), 7, f64=True)
    if world == 1:
        outs = [(0, _sharded_search(0, 1))]
    else:
        ctx = mp.get_context("spawn"); q = ctx.Queue()
        procs = [ctx.Process(target=_merge_worker, args=(r, world, 29570 + world, q)) for r in range(world)]
        for p in procs: p.start()
        outs = [q.get(timeout=300) for _ in range(world)]
        for p in procs: p.join(timeout=60)
    for rank, (gi, gs) in outs:
        assert np.array_equal(gi, ref_i.astype(np.int64)), rank
        assert np.abs(gs - ref_s).max() <= 1e-15, rank            # the oracle's BLAS may block a shard differently from the whole
    assert list(ref_i[0][:3]) == [17, 333, 900]            # the planted triple, in index order


def test_instantiate_from_config_with_the_shipped_config_structure(tmp_path):
    """INTEGRATION.md path A: `instantiate_from_config(config.model)` on a config with every key of models/rdm/imagenet/config.yaml
    (targets redirected to rdm_amd.*; constructing needs no GPU): training-only keys are accepted, the nn_memory pickle is loaded like
    ddpm.py:168-176, and UNet options that would select a graph the library does not build are refused, not ignored."""
    import pickle
    from rdm_amd.util import instantiate_from_config
    from rdm_amd.models.diffusion.ddpm import MinimalRETRODiffusion
    with open(tmp_path / "mem.p", "wb") as f:
        pickle.dump({"nn_memory": np.arange(7, 107), "id_count": {i: 2 for i in range(7, 107)}}, f)
    unet = {"target": "rdm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(
        image_size=64, in_channels=3, out_channels=3, model_channels=192, attention_resolutions=[8, 4, 2], num_res_blocks=2,
        channel_mult=[1, 2, 3, 5], use_scale_shift_norm=False, resblock_updown=False, num_head_channels=32, use_spatial_transformer=True,
        transformer_depth=1, context_dim=512, use_checkpoint=True)}
    cfg = {"base_learning_rate": 1e-4, "target": "rdm.models.diffusion.ddpm.MinimalRETRODiffusion", "params": {
        "k_nn": 4, "query_key": "clip_img_emb", "linear_start": 0.0015, "linear_end": 0.0195, "num_timesteps_cond": 1, "log_every_t": 200,
        "timesteps": 1000, "first_stage_key": "image", "cond_stage_key": "nixda", "image_size": 64, "channels": 3, "cond_stage_trainable": False,
        "nn_key": "nn_embeddings", "nn_memory": str(tmp_path / "mem.p"), "conditioning_key": "retro_only", "monitor": "val/loss_simple_ema",
        "scale_by_std": False, "ignore_keys": ["unconditional_guidance_vex"],
        "scheduler_config": {"target": "ldm.lr_scheduler.LambdaLinearScheduler", "params": {"warm_up_steps": [100]}},
        "unet_config": unet,
        "first_stage_config": {"target": "ldm.models.autoencoder.VQModelInterface", "params": {"embed_dim": 3, "n_embed": 8192, "ddconfig": {
            "double_z": False, "z_channels": 3, "resolution": 256, "in_channels": 3, "out_ch": 3, "ch": 128, "ch_mult": [1, 2, 4],
            "num_res_blocks": 2, "attn_resolutions": [], "dropout": 0.0}, "lossconfig": {"target": "torch.nn.Identity"}}},
        "retrieval_cfg": {"target": "rdm.data.retrieval_dataset.dsetbuilder.DatasetBuilder", "params": {"k": 20, "saved_embeddings": "database/openimages"}},
        "retrieval_encoder_cfg": {"target": "torch.nn.Identity"}, "cond_stage_config": "__is_unconditional__"}}
    m = instantiate_from_config(cfg)
    assert isinstance(m, MinimalRETRODiffusion) and m._ctx is None                      # no GPU touched
    assert (m.k_nn, m.image_size, m.channels, m.num_timesteps) == (4, 64, 3, 1000)
    assert m.unet_cfg.model_channels == 192 and [m.unet_cfg.channel_mult[i] for i in range(4)] == [1, 2, 3, 5]
    assert m.vq_cfg.n_embed == 8192 and m.use_memory and m.nn_memory.shape == (100,) and m.id_count[7] == 2
    cfg["params"]["nn_memory"] = "nn_memory/oi_imagenet.p"                               # file not there: no memory, like the reference
    assert not instantiate_from_config(cfg).use_memory
    for bad in ({"use_scale_shift_norm": True}, {"resblock_updown": True}, {"transformer_depth": 2}, {"use_spatial_transformer": False}, {"num_classes": 10}):
        c2 = {"target": cfg["target"], "params": dict(cfg["params"], unet_config={"params": dict(unet["params"], **bad)})}
        with pytest.raises(NotImplementedError):
            instantiate_from_config(c2)
    with pytest.raises(TypeError):
        instantiate_from_config({"target": cfg["target"], "params": dict(cfg["params"], unet_config={"params": dict(unet["params"], no_such_option=1)})})


def _worker_avg(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from rdm_amd import parallel
    parallel.init_distributed("gloo")
    g = torch.Generator().manual_seed(50 + rank)
    grads = {f"p{i}": torch.randn(sh, generator=g) for i, sh in enumerate([(7,), (33, 5), (1000,), (3, 3, 3, 4), (2049,)])}
    out = parallel.average_gradients({k: v.clone() for k, v in grads.items()}, bucket_bytes=4096)      # several buckets, one tensor > bucket
    q.put((rank, {k: v.numpy() for k, v in grads.items()}, {k: v.numpy() for k, v in out.items()}))
    parallel.shutdown()


def test_gradient_averaging_world2_gloo():
    """The training step's data-parallel exchange: bucketed all-reduce of the gradient dict == the plain per-tensor mean, same on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_avg, args=(r, 2, 29691, q)) for r in range(2)]
    for p in procs: p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    (_, g0, o0), (_, g1, o1) = got
    for k in g0:
        assert o0[k].shape == g0[k].shape
        assert np.array_equal(o0[k], o1[k])
        assert np.allclose(o0[k], (g0[k] + g1[k]) / 2, rtol=0, atol=1e-7)


def test_train_spec_matches_oracle_block_table():
    """training_unet.TrainSpec (built from the C config) lists the same modules, in the same order, as the oracle's restatement of the
    reference constructor (openaimodel.py:144-305) for the shipped and the reduced topology."""
    from oracle import unet as ounet
    from rdm_amd import _lib, synthetic, training_unet as TU
    for kw, spec in (({}, ounet.shipped_spec()),
                     (dict(model_channels=64, num_res_blocks=1, attention_resolutions=(2, 4), channel_mult=(1, 2, 3)), ounet.tiny_spec())):
        cfg = _lib.make_unet_cfg(**kw)
        t = TU.TrainSpec(cfg)
        assert t.blocks == spec.blocks
        assert (t.in_channels, t.out_channels, t.model_channels) == (spec.in_channels, spec.out_channels, spec.model_channels)
        assert synthetic.unet_param_shapes(cfg) == ounet.param_shapes(spec)


def test_configure_optimizers_resumes_litema_from_checkpoint():
    """A checkpoint's LitEma buffers (`model_ema.<name without dots>`, `model_ema.num_updates`, `model_ema.decay`) are restored by the
    reference's load_state_dict; configure_optimizers must continue from them, not restart the warm-up on a clone of the live weights
    (the first update would then pull the average onto the live weights with decay 2/11)."""
    from oracle import unet as ounet
    from rdm_amd.models.diffusion.ddpm import MinimalRETRODiffusion
    from _util import spec_to_unet_cfg

    class Ctx:
        device = torch.device("cpu")
        def load_unet(self, cfg, blob): self.loaded = len(blob)

    spec = ounet.tiny_spec()
    cfg = spec_to_unet_cfg(spec)
    live = ounet.synth_state_dict(ounet.param_shapes(spec), seed=3)
    ema = ounet.synth_state_dict(ounet.param_shapes(spec), seed=4)
    ckpt = {"model.diffusion_model." + k: v for k, v in live.items()}
    ckpt.update({"model_ema." + ("diffusion_model." + k).replace(".", ""): v for k, v in ema.items()})
    ckpt["model_ema.num_updates"] = torch.tensor(12345, dtype=torch.int)
    ckpt["model_ema.decay"] = torch.tensor(0.999, dtype=torch.float32)
    m = MinimalRETRODiffusion(unet_config={"params": {}}, ctx=Ctx())
    m.unet_cfg = cfg
    m.load_state_dict(ckpt)
    st = m.configure_optimizers()
    k0 = "input_blocks.1.0.in_layers.2.weight"
    assert st.ema.num_updates == 12345 and abs(st.ema.decay - 0.999) < 1e-7
    assert torch.equal(st.ema.shadow[k0], ema[k0].permute(0, 2, 3, 1).contiguous())          # native conv layout, EMA values
    assert torch.equal(st.P[k0], live[k0].permute(0, 2, 3, 1).contiguous())                  # masters = the LIVE weights
    # a checkpoint without EMA entries (or explicit weights) starts a fresh average on the live weights
    m.load_state_dict({"model.diffusion_model." + k: v for k, v in live.items()})
    st = m.configure_optimizers()
    assert st.ema.num_updates == 0 and st.ema.decay == 0.9999 and torch.equal(st.ema.shadow[k0], st.P[k0])


def test_bench_box_sampler_without_sensors_and_calibration_reference():
    """bench.py's calibration helpers must never cost the benchmark line: on a host without amdgpu hwmon files and without rocm-smi (this
    container) the sampler comes back with nulls, and the committed reference box has the fields the normalisation reads."""
    import importlib.util
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    s = b.BoxSampler(period=0.05)
    s.start(); time.sleep(0.2)
    out = s.stop()
    assert out["samples"] == 0 or (out["sclk_mhz_mean"] > 0 and out["power_w_mean"] > 0)
    assert set(out) >= {"sclk_mhz_mean", "sclk_mhz_min", "power_w_mean", "samples", "sampler"}
    ref = b.load_calibration_reference()
    assert ref and ref["sclk_mhz_mean"] > 1000 and 0 < ref["clock_bound_share"] <= 1 and ref["mfma_probe_tflops"] > 1000 and ref["hbm_stream_gbps"] > 1000
