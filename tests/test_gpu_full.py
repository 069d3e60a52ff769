"""Full-size parity of the BENCHMARKED pipeline (BASELINE configs #2, #3, #4) through the C ABI, against golden vectors
generated in the build container from the reference's in-tree classes (tools/gen_golden_full.py -> tests/golden/full_*.npz).

Stated tolerances (relative L2 against the fp32 reference; the HIP path stores activations in bf16 and accumulates in fp32):
    one UNet forward / one teacher-forced sampler step at the shipped size   <= 2.5e-2   (as tests/test_gpu_models.py)
    free-running trajectory, state after loop iteration i:
        DDIM  (50 steps, eta 0, CFG 2.0)      rel L2(x_i)  <= DDIM_E0 * (1 + DDIM_G) ** i   with DDIM_G = 0: NO growth allowed
        DDPM  (250 ancestral steps, k = 16)   rel L2(z)    <= DDPM_FINAL
    VQ-f4 decode at the shipped size: code indices agree >= 99.5 %, image rel L2 <= 2.5e-2 when all codes agree
    ViT-B/32 towers <= 2e-2
    retrieval over the full 20 927 907-row database: indices bit-exact, scores <= 1e-6
"""
import time

import numpy as np
import pytest
import torch

from oracle import clip as oclip
from oracle import diffusion as odiff
from oracle import retrieval as oret
from oracle import unet as ounet
from oracle import vqdecoder as ovq

from _util import golden, rel_l2, spec_to_clip_cfg, spec_to_unet_cfg, spec_to_vq_cfg

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

# per-step growth bound of the free-running DDIM trajectory: the state error after loop iteration i (0-based) must stay
# below DDIM_E0 * (1 + DDIM_G)^i.  Measured: 2.1e-3 after the first step, 3.8e-3 from iteration 10 to
# the final latent -- the error of 100 chained bf16 forwards does not accumulate (each step's eps error enters x scaled by
# its DDIM coefficient, and later steps at low t contribute ~1e-5) -- so the bound is flat at ~2.5x the measured value.
DDIM_E0, DDIM_G = 1.0e-2, 0.0
DDPM_FINAL = 1.0e-2             # measured 4.1e-3 after 250 ancestral steps


@pytest.fixture(scope="module")
def shipped(ctx):
    from rdm_amd import packing
    spec = ounet.shipped_spec()
    sd = ounet.synth_state_dict(ounet.param_shapes(spec), seed=1234)
    cfg = spec_to_unet_cfg(spec)
    ctx.load_unet(cfg, packing.pack("unet", cfg, sd))
    return ctx


@pytest.mark.parametrize("tag", ["ddim_k4", "ddim_k1"])
def test_ddim_50_steps_shipped(shipped, tag):
    """Config #3 (k=4) / #2 (k=1): 50-step DDIM, eta 0, CFG 2.0 with zero unconditional context, shipped UNet, B=1
    (rdm/models/diffusion/ddim.py:142-268).  Free-running trajectory against the reference trajectory at the stored
    iterations + one teacher-forced step from each stored reference state.

    What the late iterations test (verdict round 4): the weights are RANDOM (no checkpoint is reachable), so the guided eps prediction is
    not a denoiser and the reference trajectory itself leaves the data distribution -- the stored fp32 reference states have rms 1.0 (x_T),
    1.2, 6.0, 28 and 72 after iterations 0, 10, 25 and 49 (|x| up to 341; the second golden trajectory and the k = 1 one behave alike; the
    numbers are printed below from the fixture).  The late-iteration comparisons are therefore parity on out-of-distribution activations:
    they pin the arithmetic (error growth per step stays under the bound), not image quality; the in-distribution evidence is the stage-level
    comparison of tests/test_gpu_emul.py (unit-scale inputs) and the teacher-forced steps from iterations 0 and 10 here."""
    from rdm_amd.models.diffusion.ddim import DDIMSampler
    ctx = shipped
    g = golden(f"full_{tag}.npz")
    x_T, cond = torch.from_numpy(g["x_T"]), torch.from_numpy(g["cond"])
    uncond = torch.zeros_like(cond)
    sched = odiff.Schedule()
    z, xi, pi = ctx.ddim_sample(50, x_T, cond, uncond, sched.alphas_cumprod, eta=0.0, scale=float(g["scale"]), log_every_t=1,
                                want_intermediates=True)
    torch.cuda.synchronize()
    assert xi.shape[0] == 50
    print(f"[{tag}] rms of the fp32 REFERENCE states (random weights: the trajectory leaves the data distribution):",
          {"x_T": f"{float(x_T.pow(2).mean().sqrt()):.2f}", **{int(i): f"{float(torch.from_numpy(g[f'x_{int(i)}']).double().pow(2).mean().sqrt()):.2f}" for i in g["steps"]}})
    report = []
    for i in g["steps"]:
        i = int(i)
        ex = rel_l2(xi[i], torch.from_numpy(g[f"x_{i}"]))
        ep = rel_l2(pi[i], torch.from_numpy(g[f"px0_{i}"]))
        bound = DDIM_E0 * (1 + DDIM_G) ** i
        report.append((i, ex, ep, bound))
    ez = rel_l2(z, torch.from_numpy(g["z"]))
    print(f"[{tag}] free-running (iteration, rel L2 x, rel L2 pred_x0, bound):", [(i, f"{a:.3e}", f"{b:.3e}", f"{c:.3e}") for i, a, b, c in report],
          f"final {ez:.3e}")
    for i, ex, ep, bound in report:
        assert ex <= bound, f"iteration {i}: {ex} > {bound}"
    assert ez <= DDIM_E0 * (1 + DDIM_G) ** 49

    # teacher-forced: one sampler step (UNet forward with CFG + DDIM update through the Python-visible p_sample_ddim surface)
    class _M:      # the attributes DDIMSampler reads from the model (ddim.py:18, 30-36)
        num_timesteps = sched.num_timesteps; alphas_cumprod = sched.alphas_cumprod; device = ctx.device
        def apply_model(self, x, t, c): return ctx.unet_forward(x, t, c)
    smp = DDIMSampler(_M())
    smp.make_schedule(50, ddim_eta=0.0, verbose=False)
    ts = np.flip(smp.ddim_timesteps)
    for i in g["steps"]:
        i = int(i)
        index = 50 - i - 1
        t = torch.full((1,), int(ts[i]), dtype=torch.long, device=ctx.device)
        x_prev, px0 = smp.p_sample_ddim(torch.from_numpy(g[f"xin_{i}"]).to(ctx.device), cond.to(ctx.device), t, index,
                                        unconditional_guidance_scale=float(g["scale"]), unconditional_conditioning=uncond.to(ctx.device),
                                        noise=torch.zeros_like(x_T).to(ctx.device))
        e1, e2 = rel_l2(x_prev, torch.from_numpy(g[f"x_{i}"])), rel_l2(px0, torch.from_numpy(g[f"px0_{i}"]))
        print(f"[{tag}] teacher-forced iteration {i} (t={int(ts[i])}): x_prev {e1:.3e} pred_x0 {e2:.3e}")
        assert e1 <= 2.5e-2 and e2 <= 2.5e-2


def test_ddim_50_steps_shipped_batch64(shipped):
    """The benchmark's own batch geometry (config #3: B = 64, CFG => UNet batch 128): tall tiles, K-split, wide GEGLU tiles, the shared
    guidance prefix on 64 samples and the zero-context shortcut only run at this size.  Rows 0 and 63 carry the inputs of two different
    golden trajectories, rows 1..62 are random; samples are independent, so both must track their reference within the same flat bound."""
    ctx = shipped
    g, g2 = golden("full_ddim_k4.npz"), golden("full_ddim_k4_b.npz")
    gen = torch.Generator().manual_seed(77)
    # row 0 and row 63 (last tile, the wrapped-skip partner of row 63 - 64 in the doubled batch) carry two DIFFERENT reference trajectories
    x_T = torch.cat([torch.from_numpy(g["x_T"]), torch.randn(62, 3, 64, 64, generator=gen), torch.from_numpy(g2["x_T"])])
    cond = torch.cat([torch.from_numpy(g["cond"]), torch.randn(62, 4, 512, generator=gen) * 0.45, torch.from_numpy(g2["cond"])])
    sched = odiff.Schedule()
    z, xi, pi = ctx.ddim_sample(50, x_T, cond, torch.zeros_like(cond), sched.alphas_cumprod, eta=0.0, scale=float(g["scale"]), log_every_t=1,
                                want_intermediates=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(z).all())
    for row, gg in ((0, g), (63, g2)):
        errs = {int(i): rel_l2(xi[int(i), row:row + 1], torch.from_numpy(gg[f"x_{int(i)}"])) for i in gg["steps"]}
        ez = rel_l2(z[row:row + 1], torch.from_numpy(gg["z"]))
        print(f"[ddim_k4, batch 64] row {row} vs reference trajectory:", {i: f"{e:.3e}" for i, e in errs.items()}, f"final {ez:.3e}")
        assert all(e <= DDIM_E0 for e in errs.values()) and ez <= DDIM_E0, f"row {row}"


def test_ddpm_250_steps_shipped_k16_batch64(shipped):
    """Config #4 per-GPU geometry: B = 64, k = 16, 250 ancestral steps; row 0 = the golden trajectory's inputs and noise."""
    ctx = shipped
    g, g2 = golden("full_ddpm_k16.npz"), golden("full_ddpm_k16_b.npz")
    T = int(g["timesteps"])
    gen = torch.Generator().manual_seed(78)
    x_T = torch.cat([torch.from_numpy(g["x_T"]), torch.randn(62, 3, 64, 64, generator=gen), torch.from_numpy(g2["x_T"])])
    cond = torch.cat([torch.from_numpy(g["cond"]), torch.randn(62, 16, 512, generator=gen) * 0.45, torch.from_numpy(g2["cond"])])
    n0 = torch.from_numpy(np.random.default_rng(int(g["noise_seed"])).standard_normal((T, 1, 3, 64, 64)).astype(np.float32))
    n63 = torch.from_numpy(np.random.default_rng(int(g2["noise_seed"])).standard_normal((T, 1, 3, 64, 64)).astype(np.float32))
    noise = torch.cat([n0, torch.randn(T, 62, 3, 64, 64, generator=gen), n63], dim=1)
    s = odiff.Schedule()
    sched = {n: getattr(s, n).numpy() for n in ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
                                                "posterior_mean_coef2", "posterior_log_variance_clipped")}
    z = ctx.ddpm_sample(T, x_T, cond, noise, sched, clip_denoised=True)
    torch.cuda.synchronize()
    ez, ez63 = rel_l2(z[:1], torch.from_numpy(g["z"])), rel_l2(z[63:], torch.from_numpy(g2["z"]))
    print(f"[ddpm_k16, batch 64] final rel L2: row 0 {ez:.3e}, row 63 (second reference trajectory) {ez63:.3e}")
    assert bool(torch.isfinite(z).all()) and ez <= DDPM_FINAL and ez63 <= DDPM_FINAL


def test_ddpm_250_steps_shipped_k16(shipped):
    """Config #4: ldm p_sample_loop(timesteps=250) (reached from rdm/models/diffusion/ddpm.py:1007-1009), k = 16 neighbours
    (attention-kernel cross-attention path), no CFG, clip_denoised; the per-step noise is default_rng(noise_seed)."""
    ctx = shipped
    g = golden("full_ddpm_k16.npz")
    T = int(g["timesteps"])
    x_T, cond = torch.from_numpy(g["x_T"]), torch.from_numpy(g["cond"])
    noise = torch.from_numpy(np.random.default_rng(int(g["noise_seed"])).standard_normal((T,) + tuple(x_T.shape)).astype(np.float32))
    s = odiff.Schedule()
    sched = {n: getattr(s, n).numpy() for n in ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
                                                "posterior_mean_coef2", "posterior_log_variance_clipped")}
    z = ctx.ddpm_sample(T, x_T, cond, noise, sched, clip_denoised=True)
    torch.cuda.synchronize()
    ez = rel_l2(z, torch.from_numpy(g["z"]))
    print(f"[ddpm_k16] free-running 250 steps: final rel L2 {ez:.3e}")
    assert ez <= DDPM_FINAL
    # teacher-forced single steps from the stored reference states (native UNet forward + the oracle's fp32 update)
    for n in g["steps"]:
        n = int(n); i = T - 1 - n
        t = torch.full((1,), i, dtype=torch.long)
        apply = lambda x, t_, c: ctx.unet_forward(x, t_, c).cpu()
        x_prev = odiff.p_sample_ddpm(apply, s, torch.from_numpy(g[f"xin_{n}"]), cond, t, noise[n], True)
        e = rel_l2(x_prev, torch.from_numpy(g[f"x_{n}"]))
        print(f"[ddpm_k16] teacher-forced n={n} (t={i}): {e:.3e}")
        assert e <= 2.5e-2


def test_vq_decode_shipped(ctx):
    """VQ-f4 decode at the shipped size (ch 128, 8192 codes, 4096-token d=512 mid attention, 128/256-px convs):
    models/rdm/imagenet/config.yaml:60-80; call site rdm/models/diffusion/ddpm.py:840."""
    from rdm_amd import packing
    g = golden("full_vq.npz")
    vs = ovq.shipped_vq_spec()
    sd = ounet.synth_state_dict(ovq.vq_param_shapes(vs), seed=int(g["seed"]))
    cfg = spec_to_vq_cfg(vs)
    ctx.load_vq(cfg, packing.pack("vq", cfg, sd))
    z = torch.from_numpy(g["z"])
    img, idx = ctx.vq_decode(z, return_indices=True)
    torch.cuda.synchronize()
    agree = float((idx.cpu().numpy() == g["indices"]).mean())
    ref = torch.from_numpy(g["image"].astype(np.float32))
    e = rel_l2(img, ref)
    print(f"[vq shipped] code agreement {agree:.5f}, image rel L2 {e:.3e}")
    assert agree >= 0.995
    if agree == 1.0:
        assert e <= 2.5e-2
    else:       # a flipped near-tie code changes its neighbourhood: compare the decoder proper on the reference's codes
        e2 = rel_l2(ctx.vq_decode(z, force_not_quantize=True), ovq.vq_decode(sd, vs, z, force_not_quantize=True))
        print(f"[vq shipped] decoder only (no quantiser) rel L2 {e2:.3e}")
        assert e2 <= 2.5e-2
    # batch of 64 (the benchmark's decode batch): rows are independent; row 63 carries a SECOND reference latent / image
    g2 = golden("full_vq_b.npz")
    zb = torch.cat([z, torch.randn(62, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 0.6, torch.from_numpy(g2["z"])])
    imgb, idxb = ctx.vq_decode(zb, return_indices=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(imgb).all())
    assert rel_l2(imgb[:1], img) <= 2.5e-2
    agree63 = float((idxb.view(64, -1)[63].cpu().numpy() == g2["indices"]).mean())
    e63 = rel_l2(imgb[63:], torch.from_numpy(g2["image"].astype(np.float32)))
    print(f"[vq shipped, batch 64] row 63: code agreement {agree63:.5f}, image rel L2 {e63:.3e}")
    assert agree63 >= 0.995
    if agree63 == 1.0:
        assert e63 <= 2.5e-2


def test_clip_vitb32_full(ctx):
    """ViT-B/32 text tower on real tokenised captions and image tower on seeded images against the reference class
    (rdm/modules/custom_clip/model.py:238-336)."""
    from rdm_amd import packing
    from rdm_amd.modules.custom_clip.tokenizer import tokenize
    g = golden("full_clip.npz")
    spec = oclip.vitb32_spec()
    sd = ounet.synth_state_dict(oclip.clip_param_shapes(spec), seed=int(g["seed"]))
    cfg = spec_to_clip_cfg(spec)
    ctx.load_clip(cfg, packing.pack("clip", cfg, sd))
    tokens = tokenize([str(c) for c in g["captions"]])
    assert np.array_equal(tokens, g["tokens"])                       # own BPE == reference tokenizer on these captions
    t = ctx.clip_encode_text(torch.from_numpy(tokens))
    img = torch.from_numpy(np.random.default_rng(int(g["seed"]) + 7).standard_normal((2, 3, 224, 224)).astype(np.float32))
    i = ctx.clip_encode_image(img)
    torch.cuda.synchronize()
    et, ei = rel_l2(t, torch.from_numpy(g["text_out"])), rel_l2(i, torch.from_numpy(g["image_out"]))
    print(f"[clip ViT-B/32] text rel L2 {et:.3e}, image rel L2 {ei:.3e}")
    assert et <= 2e-2 and ei <= 2e-2


def test_knn_full_database_bit_exact(ctx):
    """The benchmark's retrieval: B = 64 queries, k = 4, N = 20 927 907 rows x 512 fp16, against the exact
    fp64 oracle streamed over the same rows on the host (chunks of 1 Mi rows)."""
    N, B, k, D = 20_927_907, 64, 4, 512
    d = ctx.device
    gen = torch.Generator(device=d).manual_seed(7)
    db = torch.empty((N, D), device=d, dtype=torch.float16)
    for r0 in range(0, N, 1 << 20):
        r1 = min(N, r0 + (1 << 20))
        db[r0:r1] = (torch.randn((r1 - r0, D), device=d, generator=gen) * 0.45).half()
    q = (torch.randn((B, D), device=d, generator=gen) * 0.45)
    # planted rows: near-duplicates of some queries in the first tile, the ragged last tile and across block boundaries
    for j, r in enumerate((0, 255, 256, N - 1, N - 130, 12_345_678, 1 << 20, (1 << 20) - 1)):
        db[r] = q[j].half()
    db[777] = db[12_345_678]                                         # exact duplicate: tie resolves to the lower index
    ctx.db_load(db)
    idx, sc = ctx.knn(q, k)
    torch.cuda.synchronize()
    t0 = time.time()
    st = oret.StreamingTopK(oret.normalize_queries(q.cpu().numpy()), k)
    for r0 in range(0, N, 1 << 20):
        r1 = min(N, r0 + (1 << 20))
        st.push(oret.StreamingTopK.normalize_chunk(db[r0:r1].cpu()), r0)
    ref_i, ref_s = st.result()
    print(f"[knn full] oracle over {N} rows in {time.time() - t0:.0f} s")
    got = idx.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, ref_i), f"top-k indices differ in {(got != ref_i).sum()} places"
    assert np.abs(sc.cpu().numpy() - ref_s).max() <= 1e-6
    assert got[5, 0] == 777                                          # the duplicate pair: lower index first
