"""GPU parity of the training entry on the reference's surface: the first-stage ENCODER, the HIP glue ops
around the UNet in shared_step / p_losses, `MinimalRETRODiffusion.training_step` from images + neighbour embeddings against torch
autograd + torch.optim.AdamW on the oracle, and the gradients of the SHIPPED-topology UNet against a reference-gradient fixture
(tools/gen_golden_grads.py).  Stated tolerances: encoder latent 2.5e-2 relative L2 (as the decoder), glue ops to fp32 / one bf16
rounding, loss curve 3e-2 relative, per-tensor gradients 5e-2 relative L2 on the sampled elements (~60 bf16 layers deep both ways)."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import bf16_round, golden, rel_l2, spec_to_unet_cfg, spec_to_vq_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _autograd_on():
    with torch.enable_grad():
        yield


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("which", ["tiny", "shipped"])
def test_vq_encode(ctx, which):
    """VQModelInterface.encode = quant_conv(Encoder(x)) (ldm, un-vendored: parity unpinned, oracle/vqdecoder.py vq_encode): asymmetric
    (0, 1, 0, 1) padding of the stride-2 Downsample convs, 4096-token mid attention at the shipped size, GroupNorm eps 1e-6."""
    from oracle import unet as ounet, vqdecoder as ovq
    from rdm_amd import packing
    torch.set_grad_enabled(False)
    vs = ovq.tiny_vq_spec() if which == "tiny" else ovq.shipped_vq_spec()
    sd = ounet.synth_state_dict(ovq.vq_encoder_param_shapes(vs), seed=17)
    cfg = spec_to_vq_cfg(vs)
    ctx.load_vq_encoder(cfg, packing.pack("vqenc", cfg, sd))
    B = 3 if which == "tiny" else 1
    img = _rand((B, 3, vs.resolution, vs.resolution), 5, 0.5).clamp(-1, 1)
    z = ctx.vq_encode(img)
    ref = ovq.vq_encode(sd, vs, img)
    e = rel_l2(z, ref)
    print(f"[vq encode {which}] latent {tuple(z.shape)} rel L2 {e:.3e}")
    assert z.shape == ref.shape and e <= 2.5e-2
    if which == "tiny":                       # rows are independent: a batch of 5 reproduces the batch of 3
        z5 = ctx.vq_encode(torch.cat([img, _rand((2, 3, vs.resolution, vs.resolution), 6, 0.5)]))
        assert rel_l2(z5[:3], z) <= 1e-2


def test_training_glue_ops(ctx):
    d = ctx.device
    B, C, H, W = 5, 3, 16, 24
    x0, nz = _rand((B, C, H, W), 1).to(d), _rand((B, C, H, W), 2).to(d)
    a, b = torch.rand(B, generator=torch.Generator().manual_seed(3)).to(d), torch.rand(B, generator=torch.Generator().manual_seed(4)).to(d)
    ref = a.view(-1, 1, 1, 1) * x0 + b.view(-1, 1, 1, 1) * nz
    out, onh = ctx.op_q_sample(x0, nz, a, b, want_nchw=True, cpad=64)
    assert torch.equal(out, ref) or (out - ref).abs().max().item() <= 1e-6
    assert onh.shape == (B, H, W, 64) and float(onh[..., C:].abs().max()) == 0.0
    assert (onh[..., :C].float() - ref.permute(0, 2, 3, 1)).abs().max().item() <= 2 ** -8 * ref.abs().max().item()
    # squared-error loss + gradient
    eps = bf16_round(_rand((B, H, W, C), 5)).to(d, torch.bfloat16)
    coef = (torch.arange(B, dtype=torch.float32) * 0.1 + 0.05).to(d)
    se, deps = ctx.op_mse_loss(eps, nz, coef)
    diff = eps.float() - nz.permute(0, 2, 3, 1)
    assert (se - (diff ** 2).mean(dim=(1, 2, 3))).abs().max().item() <= 1e-5
    assert (deps.float() - coef.view(-1, 1, 1, 1) * diff).abs().max().item() <= 2 ** -8 * float((coef.view(-1, 1, 1, 1) * diff).abs().max())
    se2, none = ctx.op_mse_loss(eps, nz)
    assert none is None and torch.equal(se2, se)
    # conditioning switch
    r, sig = _rand((B, 4, 512), 6).to(d), _rand((B, 4, 512), 7).to(d)
    mask = torch.tensor([1, 0, 0, 1, 1], dtype=torch.bool)
    got = ctx.op_where_rows(mask, sig, r)
    assert torch.equal(got, torch.where(mask.to(d).view(-1, 1, 1), sig, r))
    # timestep embedding (ldm: [cos | sin] of t * exp(-ln(10000) i / half))
    t = torch.tensor([0, 1, 37, 999, 481]).to(d)
    te = ctx.op_timestep_embedding(t, 192)
    fr = torch.exp(-np.log(10000.0) * torch.arange(96, dtype=torch.float32) / 96)
    args = t.cpu().float()[:, None] * fr[None]
    assert (te.float().cpu() - torch.cat([torch.cos(args), torch.sin(args)], -1)).abs().max().item() <= 2 ** -8 + 2e-3   # bf16 + fp32 sin/cos of arguments up to 999
    # per-sample column sums, 2x expansions
    xs = bf16_round(_rand((B, 40, 192), 8)).to(d, torch.bfloat16)
    cs = ctx.op_colsum_samples(xs)
    assert (cs.float() - xs.float().sum(1)).abs().max().item() <= 2 ** -7 * float(xs.float().sum(1).abs().max())
    xl = bf16_round(_rand((3, 1000, 320), 18)).to(d, torch.bfloat16)           # the two-stage path: ragged pixel chunks, two column blocks
    cl = ctx.op_colsum_samples(xl)
    assert (cl.float() - xl.float().sum(1)).abs().max().item() <= 2 ** -7 * float(xl.float().sum(1).abs().max())
    xe = bf16_round(_rand((2, 3, 5, 64), 9)).to(d, torch.bfloat16)
    z0, z1 = ctx.op_expand2(xe, 0), ctx.op_expand2(xe, 1)
    ref0 = torch.zeros((2, 6, 10, 64), device=d, dtype=torch.bfloat16); ref0[:, ::2, ::2] = xe
    assert torch.equal(z0, ref0) and torch.equal(z1, xe.repeat_interleave(2, 1).repeat_interleave(2, 2))


def _tiny_model(ctx, p_uncond=0.0, seed=21):
    from oracle import unet as ounet, vqdecoder as ovq
    from rdm_amd.models.diffusion.ddpm import MinimalRETRODiffusion
    # first stage with a 32 x 32 latent (two levels): the tiny UNet's attention then runs at 32^2 / 16^2 / 8^2 tokens (the backward's batched
    # GEMMs contract over the token count, a multiple of 64 at every shipped resolution)
    spec, vs = ounet.tiny_spec(), ovq.VQSpec(n_embed=512, ch=64, ch_mult=(1, 2), num_res_blocks=1, resolution=64)
    sd = {k: bf16_round(v) if v.dim() >= 2 else v for k, v in ounet.synth_state_dict(ounet.param_shapes(spec), seed=seed).items()}
    vsd = ounet.synth_state_dict({**ovq.vq_param_shapes(vs), **ovq.vq_encoder_param_shapes(vs)}, seed=seed + 1)
    up = dict(in_channels=spec.in_channels, out_channels=spec.out_channels, model_channels=spec.model_channels, num_res_blocks=spec.num_res_blocks,
              attention_resolutions=spec.attention_resolutions, channel_mult=spec.channel_mult, num_head_channels=spec.num_head_channels,
              context_dim=spec.context_dim)
    fs = {"params": {"embed_dim": vs.embed_dim, "n_embed": vs.n_embed, "mid_attn": vs.mid_attn,
                     "ddconfig": {"z_channels": vs.z_channels, "ch": vs.ch, "ch_mult": vs.ch_mult, "num_res_blocks": vs.num_res_blocks, "out_ch": vs.out_ch,
                                  "resolution": vs.resolution}}}
    m = MinimalRETRODiffusion(unet_config={"params": up}, first_stage_config=fs, k_nn=4, image_size=vs.z_res, channels=3, ctx=ctx)
    m.load_unet_state_dict(sd)
    m.load_first_stage_state_dict(vsd)
    m.p_uncond = p_uncond
    return m, spec, sd, vs, vsd


def test_training_step_from_images_tracks_torch(ctx):
    """Three optimisation steps through the REFERENCE SURFACE -- training_step(batch) with batch['image'] [B,H,W,3] in [-1,1] and
    batch['nn_embeddings'] [B,1,k,512] (rdm/models/diffusion/ddpm.py:390-443): first-stage encode, q_sample, Bernoulli(p_uncond)
    conditioning switch, UNet forward / backward, AdamW -- beside torch autograd + torch.optim.AdamW on the oracle's restatement."""
    from oracle import unet as ounet, vqdecoder as ovq
    m, spec, sd, vs, vsd = _tiny_model(ctx, p_uncond=0.5)
    B, k = 4, 4
    img = _rand((B, vs.resolution, vs.resolution, 3), 30, 0.5).clamp(-1, 1)
    nns = bf16_round(_rand((B, 1, k, 512), 31, 0.5))
    mask = torch.tensor([False, True, False, True])
    tsteps = torch.tensor([12, 480, 733, 999])
    noise = bf16_round(_rand((B, 3, vs.z_res, vs.z_res), 32))
    m.unconditional_guidance_vex = torch.zeros(512, device=ctx.device)            # the zero "no neighbours" signal of the shipped models
    m.configure_optimizers(lr=2e-4, weight_decay=1e-2, use_ema=True)
    # reference: the same pipeline in fp32 torch
    with torch.no_grad():
        z = ovq.vq_encode(vsd, vs, img.permute(0, 3, 1, 2).contiguous())
    a = m.sqrt_alphas_cumprod[tsteps].view(-1, 1, 1, 1); b = m.sqrt_one_minus_alphas_cumprod[tsteps].view(-1, 1, 1, 1)
    r = torch.where(mask.view(-1, 1, 1), torch.zeros(B, k, 512), nns.reshape(B, k, 512))
    ref_sd = {kk: v.clone().requires_grad_(True) for kk, v in sd.items()}
    opt = torch.optim.AdamW(list(ref_sd.values()), lr=2e-4, weight_decay=1e-2)
    curve = []
    zg = ctx.vq_encode(img.permute(0, 3, 1, 2).contiguous()).cpu()
    print(f"training surface: encoder latent rel L2 {rel_l2(zg, z):.2e}")
    for step in range(3):
        opt.zero_grad()
        loss_ref = ((ounet.unet_forward(ref_sd, spec, a * z + b * noise, tsteps, r) - noise) ** 2).mean()
        loss_ref.backward(); opt.step()
        loss = m.training_step({"image": img, "nn_embeddings": nns}, step, t=tsteps, noise=noise, uncond_mask=mask)
        curve.append((float(loss), loss_ref.item()))
    print(f"training surface: loss curve native vs torch {[(round(x, 4), round(y, 4)) for x, y in curve]}; keys {sorted(m.last_loss_dict)}")
    assert all(abs(x - y) <= 3e-2 * y for x, y in curve), curve
    assert curve[2][0] < curve[0][0] and curve[2][1] < curve[0][1], curve
    assert m.train_state.step == 3 and m.train_state.ema.num_updates == 3
    assert set(m.last_loss_dict) == {"train/loss_simple", "train/loss_vlb", "train/loss"}
    # validation on the trained weights: the sampler's copy follows sync_sampling_weights (EMA or live)
    v0, _ = m.validation_step({"image": img, "nn_embeddings": nns}, 0, t=tsteps, noise=noise, uncond_mask=mask)
    m.sync_sampling_weights(use_ema=False)
    v1, _ = m.validation_step({"image": img, "nn_embeddings": nns}, 0, t=tsteps, noise=noise, uncond_mask=mask)
    with torch.no_grad():
        ref_after = ((ounet.unet_forward(ref_sd, spec, a * z + b * noise, tsteps, r) - noise) ** 2).mean().item()
    print(f"training surface: validation loss before / after syncing the live weights {float(v0):.4f} / {float(v1):.4f}; torch after 3 steps {ref_after:.4f}")
    assert abs(float(v0) - curve[0][1]) <= 3e-2 * curve[0][1]             # still the initial weights
    assert abs(float(v1) - ref_after) <= 3e-2 * ref_after
    ctx.release_scratch()                                                 # training scratch handed back; sampling re-creates what it needs
    v2, _ = m.validation_step({"image": img, "nn_embeddings": nns}, 0, t=tsteps, noise=noise, uncond_mask=mask)
    assert float(v2) == float(v1)


def test_training_step_draws_like_the_reference_when_nothing_is_given(ctx):
    """t ~ randint, noise ~ randn, mask ~ Bernoulli(p_uncond) from torch's generators (ddpm.py:393-396, 406-413): seeded runs repeat."""
    m, spec, sd, vs, vsd = _tiny_model(ctx, p_uncond=0.3, seed=23)
    m.unconditional_guidance_vex = torch.zeros(512, device=ctx.device)
    img = _rand((2, vs.resolution, vs.resolution, 3), 40, 0.5).clamp(-1, 1)
    batch = {"image": img, "nn_embeddings": _rand((2, 1, 4, 512), 41, 0.5)}
    losses = []
    for _ in range(2):
        torch.manual_seed(5); torch.cuda.manual_seed_all(5)
        l, d = m.shared_step(batch)                      # forward-only (validation) form
        losses.append(float(l))
    assert losses[0] == losses[1] and np.isfinite(losses[0])


def test_whole_unet_gradients_shipped_topology(ctx):
    """The SHIPPED 400.9 M-parameter topology (models/rdm/imagenet/config.yaml:36-59) at B = 2, 64 x 64, k = 4: loss and the gradient of
    every parameter of input_blocks.{1,4,7,10}, middle_block, output_blocks.{0,5,11}, out and time_embed against CPU autograd through
    the oracle (fixture tests/golden/unet_shipped_grads.npz: norms + 1024 sampled elements per tensor, tools/gen_golden_grads.py)."""
    from oracle import unet as ounet
    from rdm_amd import training_unet as TU
    g = golden("unet_shipped_grads.npz")
    dev = ctx.device
    spec = ounet.shipped_spec()
    sd = {k: bf16_round(v) if v.dim() >= 2 else v for k, v in ounet.synth_state_dict(ounet.param_shapes(spec), seed=int(g["seed_w"])).items()}
    rng = np.random.default_rng(int(g["seed_x"]))
    x = bf16_round(torch.from_numpy(rng.standard_normal((2, 3, 64, 64)).astype(np.float32)))
    cx = bf16_round(torch.from_numpy((rng.standard_normal((2, 4, 512)) * 0.45).astype(np.float32)))
    noise = bf16_round(torch.from_numpy(rng.standard_normal((2, 3, 64, 64)).astype(np.float32)))
    t = torch.tensor([481, 37])
    P = TU.params_from_state_dict(sd, dev)
    to_nhwc = lambda v: v.permute(0, 2, 3, 1).contiguous().to(dev, torch.bfloat16)
    loss, grads, _ = TU.unet_loss_and_grads(ctx, P, TU.TrainSpec(spec_to_unet_cfg(spec)), to_nhwc(x), t.to(dev), cx.to(dev, torch.bfloat16), noise.to(dev))
    gs = TU.grads_to_state_dict_layout({k: v.cpu() for k, v in grads.items()}, sd)
    names = [k[2:] for k in g.files if k.startswith("n:")]
    assert len(names) >= 200
    errs, nerr = {}, {}
    for k in names:
        flat = gs[k].reshape(-1)
        pos = np.random.default_rng(zlib.crc32(k.encode())).integers(0, flat.numel(), size=min(int(g["nsamp"]), flat.numel()))
        ref = torch.from_numpy(g["v:" + k].astype(np.float32)) * float(g["s:" + k])
        errs[k] = rel_l2(flat[torch.from_numpy(pos)], ref)
        nerr[k] = abs(float(flat.double().norm()) / float(g["n:" + k]) - 1.0)
    worst = sorted(errs, key=errs.get)[-3:]
    wn = max(nerr, key=nerr.get)
    print(f"shipped-topology gradients: loss {loss:.5f} vs {float(g['loss']):.5f}; {len(names)} tensors; worst sampled rel L2 " +
          ", ".join(f"{k} {errs[k]:.2e}" for k in worst) + f"; worst norm mismatch {wn} {nerr[wn]:.2e}")
    assert abs(loss - float(g["loss"])) <= 2e-2 * float(g["loss"])
    assert errs[worst[-1]] <= 5e-2 + 2e-3 and nerr[wn] <= 5e-2        # + the fixture's fp16 storage of the samples
