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
    with pytest.raises(NotImplementedError):
        s.sample(5, 1, (3, 8, 8), conditioning=torch.zeros(1, 4, 512), mask=torch.ones(1, 3, 8, 8), x0=torch.zeros(1, 3, 8, 8), verbose=False)
    with pytest.raises(AssertionError):
        s.sample(5, 1, (3, 8, 8), conditioning=torch.zeros(1, 4, 512), unconditional_guidance_scale=0.5, verbose=False)


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


# ---- N > 1 path on CPU: world_size 2 over gloo
def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rdm_amd.parallel import all_gather_images, per_sample_noise, shard_range
    a, b = shard_range(n_total, world, rank)
    x = per_sample_noise(7, range(a, b), (3, 4, 4))
    local = x * 2.0 + 1.0                                    # stand-in for "sample + decode" (per-sample independent)
    out = all_gather_images(local, n_total)
    if rank == 0:
        q.put(out.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_batch_sharding_world2_gloo(n_total):
    from rdm_amd.parallel import per_sample_noise, shard_range
    assert shard_range(7, 2, 0) == (0, 4) and shard_range(7, 2, 1) == (4, 7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + n_total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs: p.start()
    got = q.get(timeout=120)
    for p in procs: p.join(timeout=60)
    ref = per_sample_noise(7, range(n_total), (3, 4, 4)).numpy() * 2.0 + 1.0      # the 1-rank result
    assert np.array_equal(got, ref)                                               # sharding is bit-invariant
