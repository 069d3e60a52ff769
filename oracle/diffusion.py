"""fp32 CPU restatement of the diffusion schedule and the DDIM / DDPM sampling loops
(oracle — test infrastructure only).

Follows:
  rdm/models/diffusion/ddim.py:27-56    make_schedule
  rdm/models/diffusion/ddim.py:142-215  ddim_sampling (loop order, intermediates rule)
  rdm/models/diffusion/ddim.py:217-268  p_sample_ddim (CFG batch doubling, update)
  [ldm, un-vendored, parity unpinned — SURVEY.md appendix A.2]
      DDPM.register_schedule, make_ddim_timesteps, make_ddim_sampling_parameters,
      LatentDiffusion.p_sample / p_sample_loop
Known-answer constants: SURVEY.md appendix C (checked in tests/test_oracle_schedule.py).
"""
import numpy as np
import torch


class Schedule:
    """[ldm] DDPM.register_schedule with beta_schedule='linear' (A.2). fp32 buffers."""

    def __init__(self, timesteps=1000, linear_start=0.0015, linear_end=0.0195, v_posterior=0.0):
        betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        self.num_timesteps = timesteps
        self.betas = f32(betas)
        self.alphas_cumprod = f32(ac)
        self.alphas_cumprod_prev = f32(ac_prev)
        self.sqrt_alphas_cumprod = f32(np.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = f32(np.sqrt(1.0 - ac))
        self.sqrt_recip_alphas_cumprod = f32(np.sqrt(1.0 / ac))
        self.sqrt_recipm1_alphas_cumprod = f32(np.sqrt(1.0 / ac - 1))
        pv = (1 - v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + v_posterior * betas
        self.posterior_variance = f32(pv)
        self.posterior_log_variance_clipped = f32(np.log(np.maximum(pv, 1e-20)))
        self.posterior_mean_coef1 = f32(betas * np.sqrt(ac_prev) / (1.0 - ac))
        self.posterior_mean_coef2 = f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))
        # [ldm] eps-parameterisation ELBO weights: betas^2 / (2 * posterior_variance * alphas * (1 - alphas_cumprod)), [0] <- [1]
        pv32, b32, a32 = self.posterior_variance, self.betas, f32(alphas)
        lv = b32 ** 2 / (2 * pv32 * a32 * (1 - self.alphas_cumprod))
        lv[0] = lv[1]
        self.lvlb_weights = lv


def make_ddim_timesteps(S, T=1000):
    """[ldm] make_ddim_timesteps('uniform'): arange(0,T,T//S)+1."""
    c = T // S
    return np.asarray(list(range(0, T, c))) + 1


def make_ddim_sampling_parameters(alphacums: torch.Tensor, ddim_timesteps, eta):
    """[ldm] make_ddim_sampling_parameters; alphacums is the model's fp32 buffer
    (ddim.py:44-46 passes alphas_cumprod.cpu()).  Returns torch fp32 alphas / sigmas and a
    numpy alphas_prev, mirroring ldm's mixed types (SURVEY A.2 'Precision')."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    a32 = alphas.numpy()            # fp32 values promoted to float64 by the float64 alphas_prev array, as in ldm
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - a32) * (1 - a32 / alphas_prev))
    return sigmas, alphas, alphas_prev


def ddim_schedule(sched: Schedule, S, eta):
    ts = make_ddim_timesteps(S, sched.num_timesteps)
    sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(sched.alphas_cumprod, ts, eta)
    # per-step fp32 scalars as materialised by torch.full_like(e_t, value) (ddim.py:253-256)
    f = lambda v: torch.as_tensor(np.asarray(v), dtype=torch.float32).reshape(-1)
    a_t = f(alphas)
    a_prev = f(alphas_prev)
    sigma = f(sigmas)
    sqrt_1m = f(np.sqrt(1.0 - np.asarray(alphas, dtype=np.float32)))  # ddim.py:52 np.sqrt(1-ddim_alphas)
    return ts, a_t, a_prev, sigma, sqrt_1m


def p_sample_ddim(apply_model, x, c, t, index, sch, *, scale=1.0, uc=None, noise=None, temperature=1.0, quantize=None):
    """ddim.py:217-268 with the options the native path supports.  quantize: callable z -> z_q, the
    `pred_x0, _, *_ = self.model.first_stage_model.quantize(pred_x0)` of quantize_denoised (:260-261)."""
    ts, a_t, a_prev, sigma, sqrt_1m = sch
    b = x.shape[0]
    assert scale >= 1.0
    if noise is None:
        noise = torch.zeros_like(x)
    if scale > 1.0:
        out = apply_model(torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([c, uc]))
        e_t, e_u = out[:b], out[b:]
        e_t = e_u + scale * (e_t - e_u)
    else:
        e_t = apply_model(x, t, c)
    at = torch.full_like(e_t, float(a_t[index]))
    ap = torch.full_like(e_t, float(a_prev[index]))
    sg = torch.full_like(e_t, float(sigma[index]))
    s1m = torch.full_like(e_t, float(sqrt_1m[index]))
    pred_x0 = (x - s1m * e_t) / at.sqrt()
    if quantize is not None:
        pred_x0 = quantize(pred_x0)
    dir_xt = (1.0 - ap - sg ** 2).sqrt() * e_t
    nz = sg * noise * temperature
    x_prev = ap.sqrt() * pred_x0 + dir_xt + nz
    return x_prev, pred_x0


def ddim_sample(apply_model, sched: Schedule, S, x_T, cond, *, eta=0.0, scale=1.0, uncond=None,
                noise=None, log_every_t=100, temperature=1.0, mask=None, x0=None, q_noise=None, content_cond=None,
                style_cond=None, timesteps=None, score_corrector=None, quantize=None):
    """ddim.py:142-215. `noise` is an optional [S,B,...] stack consumed in loop order.  Optional loop-body options of the
    reference: inpainting (`mask`, `x0`; `q_noise` = explicit stack for the q_sample draw at :187), style / content conditioning
    by SNR band (:179-184), timestep subset (:158-160), score corrector (:239-241; a callable e_t, x, t, c -> e_t)."""
    sch = ddim_schedule(sched, S, eta)
    ts = sch[0]
    if timesteps is not None:
        subset_end = int(min(timesteps / ts.shape[0], 1) * ts.shape[0]) - 1
        ts = ts[:subset_end]
    img = x_T
    inter = {"x_inter": [img], "pred_x0": [img]}
    total = ts.shape[0]
    a_t = sch[1]
    for i, step in enumerate(np.flip(ts)):
        index = total - i - 1
        t = torch.full((x_T.shape[0],), int(step), dtype=torch.long)
        nz = None if noise is None else noise[i]
        snr = float(a_t[index]) / (1.0 - float(a_t[index]))
        c_in = cond
        if style_cond is not None and snr < 5.e-2:
            c_in = style_cond
        if content_cond is not None and snr >= 5.e-2 and snr < 1.:
            c_in = content_cond
        if mask is not None:
            qn = torch.zeros_like(x0) if q_noise is None else q_noise[i]
            ac = sched.alphas_cumprod[t].reshape(-1, 1, 1, 1)
            img_orig = ac.sqrt() * x0 + (1.0 - ac).sqrt() * qn       # ldm q_sample with sqrt_alphas_cumprod buffers
            img = img_orig * mask + (1. - mask) * img
        am = apply_model
        if score_corrector is not None:                               # corrector sees the guided e_t: wrap p_sample_ddim's pieces
            img, pred_x0 = _p_sample_ddim_corrected(apply_model, img, c_in, t, index, sch, scale, uncond, nz, temperature, score_corrector)
        else:
            img, pred_x0 = p_sample_ddim(am, img, c_in, t, index, sch, scale=scale, uc=uncond, noise=nz, temperature=temperature, quantize=quantize)
        if index % log_every_t == 0 or index == total - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter


def _p_sample_ddim_corrected(apply_model, x, c, t, index, sch, scale, uc, noise, temperature, corrector):
    ts, a_t, a_prev, sigma, sqrt_1m = sch
    b = x.shape[0]
    if noise is None:
        noise = torch.zeros_like(x)
    if scale > 1.0:
        out = apply_model(torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([c, uc]))
        e_t = out[b:] + scale * (out[:b] - out[b:])
    else:
        e_t = apply_model(x, t, c)
    e_t = corrector(e_t, x, t, c)
    at = torch.full_like(e_t, float(a_t[index])); ap = torch.full_like(e_t, float(a_prev[index]))
    sg = torch.full_like(e_t, float(sigma[index])); s1m = torch.full_like(e_t, float(sqrt_1m[index]))
    pred_x0 = (x - s1m * e_t) / at.sqrt()
    x_prev = ap.sqrt() * pred_x0 + (1.0 - ap - sg ** 2).sqrt() * e_t + sg * noise * temperature
    return x_prev, pred_x0


def p_sample_ddpm(apply_model, sched: Schedule, x, c, t, noise, clip_denoised=True, temperature=1.0):
    """[ldm] LatentDiffusion.p_sample / p_mean_variance / q_posterior (A.2)."""
    ti = int(t[0])
    eps = apply_model(x, t, c)
    x0 = sched.sqrt_recip_alphas_cumprod[ti] * x - sched.sqrt_recipm1_alphas_cumprod[ti] * eps
    if clip_denoised:
        x0 = x0.clamp(-1.0, 1.0)
    mean = sched.posterior_mean_coef1[ti] * x0 + sched.posterior_mean_coef2[ti] * x
    logvar = sched.posterior_log_variance_clipped[ti]
    nonzero = 0.0 if ti == 0 else 1.0
    return mean + nonzero * (0.5 * logvar).exp() * noise * temperature


def ddpm_sample(apply_model, sched: Schedule, x_T, cond, noise, timesteps=None, clip_denoised=True):
    """[ldm] p_sample_loop: for i in reversed(range(timesteps or T)). noise [T',B,...]."""
    T = timesteps or sched.num_timesteps
    img = x_T
    for n, i in enumerate(reversed(range(T))):
        t = torch.full((x_T.shape[0],), i, dtype=torch.long)
        img = p_sample_ddpm(apply_model, sched, img, cond, t, noise[n], clip_denoised)
    return img


def shared_step_loss(apply_model, sched: Schedule, x, nns, t, noise, *, uncond_mask=None, uncond_signal=None, l_simple_weight=1.0,
                     original_elbo_weight=0.0, prefix="val"):
    """MinimalRETRODiffusion.shared_step / forward (rdm/models/diffusion/ddpm.py:390-443; shipped configs: nn_encoder None,
    retrieval_encoder Identity, no second conditioning) + [ldm] LatentDiffusion.p_losses (loss_type l2, eps parameterisation,
    logvar = 0, learn_logvar False), forward only -- what validation_step logs.  x: latent [B,C,H,W]; nns [B,n,k,D] neighbour
    embeddings of the batch (ddpm.py:363-365: 'b n k d -> b (n k) d'); uncond_mask [B] bool = the Bernoulli(p_uncond) draw (:393-396)."""
    r = nns.reshape(nns.shape[0], -1, nns.shape[-1]).float()
    if uncond_mask is not None:
        r = torch.where(uncond_mask.reshape(-1, 1, 1), uncond_signal, r)
    a = sched.sqrt_alphas_cumprod[t].reshape(-1, 1, 1, 1); b = sched.sqrt_one_minus_alphas_cumprod[t].reshape(-1, 1, 1, 1)
    x_noisy = a * x + b * noise
    out = apply_model(x_noisy, t, r)
    loss_simple = ((out - noise) ** 2).mean(dim=(1, 2, 3))
    d = {f"{prefix}/loss_simple": loss_simple.mean()}
    logvar_t = torch.zeros_like(loss_simple)
    loss = (loss_simple / torch.exp(logvar_t) + logvar_t)
    loss = l_simple_weight * loss.mean()
    loss_vlb = (sched.lvlb_weights[t] * ((out - noise) ** 2).mean(dim=(1, 2, 3))).mean()
    d[f"{prefix}/loss_vlb"] = loss_vlb
    loss = loss + original_elbo_weight * loss_vlb
    d[f"{prefix}/loss"] = loss
    return loss, d
