"""SAMPLING ONLY — native counterpart of rdm/models/diffusion/ddim.py:14-268 (DDIMSampler).

Same constructor, `make_schedule`, `sample`, `ddim_sampling`, `p_sample_ddim` signatures and return values; the
S-step loop itself runs inside librdm_hip (rdm_ddim_sample): UNet forward with CFG batch doubling + fused update
per step, K/V of the neighbours projected once per call.  Options that change the loop body per step (callbacks, mask / x0
inpainting, style_cond / content_cond by SNR band, timestep subset, noise_dropout, score_corrector) take the per-step path:
native UNet forward, torch elementwise update.  quantize_x0 (round 4) also takes the per-step path: pred_x0 is snapped to the
first stage's codebook by the native quantiser (rdm_vq_quantize).  ddim_use_original_steps raises NotImplementedError: in the
reference that branch reads `self.model.ddim_sigmas_for_original_num_steps` (ddim.py:249), a buffer make_schedule registers on the
SAMPLER (:52), so it ends in AttributeError there -- dead code, not a feature to mirror.  Unlike the reference, nothing is forced
onto a "cuda" device string (ddim.py:21-25) and `--seed` style RNG comes from the caller's torch generator.
"""
import numpy as np
import torch


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method != "uniform":
        raise NotImplementedError(f'ddim discretization "{ddim_discr_method}"')
    c = num_ddpm_timesteps // num_ddim_timesteps
    ts = np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1      # NB: S that does not divide T yields > S steps (ldm quirk)
    if ts[-1] >= num_ddpm_timesteps:
        raise ValueError(f"S={num_ddim_timesteps} gives ddim timestep {ts[-1]} >= T={num_ddpm_timesteps} "
                         "(the reference indexes alphas_cumprod out of range for this S)")
    return ts


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    a = np.asarray(alphacums, dtype=np.float32)
    alphas = a[ddim_timesteps]
    alphas_prev = np.asarray([a[0]] + a[ddim_timesteps[:-1]].tolist())          # float64 array of fp32 values
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        ac = self.model.alphas_cumprod.detach().float().cpu()
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        self.register_buffer('alphas_cumprod', ac)
        sig, al, alp = make_ddim_sampling_parameters(ac.numpy(), self.ddim_timesteps, ddim_eta, verbose)
        self.register_buffer('ddim_sigmas', sig)
        self.register_buffer('ddim_alphas', al)
        self.register_buffer('ddim_alphas_prev', alp)
        self.register_buffer('ddim_sqrt_one_minus_alphas', np.sqrt(1. - al))

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, random_guiding='none', r_shape=None, retro_cond=None,
               return_neighbors=False, k_nn=None, ignore_noising=False, content_cond=None, style_cond=None,
               intermediates_to_cpu=False, **kwargs):
        if conditioning is not None and not isinstance(conditioning, (dict, list)):
            if conditioning.shape[0] != batch_size:
                print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        size = (batch_size,) + tuple(shape)
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, noise_dropout=noise_dropout,
                                  temperature=temperature, score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, random_guiding=random_guiding,
                                  content_cond=content_cond, style_cond=style_cond,
                                  intermediates_to_cpu=intermediates_to_cpu, S=S, eta=eta, noise=kwargs.get("noise"),
                                  q_noise=kwargs.get("q_noise"))

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, random_guiding='none', content_cond=None, style_cond=None,
                      intermediates_to_cpu=False, S=None, eta=0., **kwargs):
        if ddim_use_original_steps:
            raise NotImplementedError("ddim_use_original_steps: dead in the reference (ddim.py:249 reads a buffer that lives on the sampler, "
                                      "not on the model: AttributeError)")
        # options that change the loop body step by step run through the per-step path (native UNet forward per step)
        per_step = (quantize_denoised or mask is not None or x0 is not None or noise_dropout > 0. or score_corrector is not None or
                    content_cond is not None or style_cond is not None or random_guiding != 'none' or timesteps is not None)
        if isinstance(cond, dict):
            cond = cond[list(cond.keys())[0]]
        if isinstance(cond, list):
            if len(cond) != 1:
                raise NotImplementedError("native DDIM loop takes a single cross-attention conditioning tensor")
            cond = cond[0]
        if isinstance(unconditional_conditioning, list):
            unconditional_conditioning = unconditional_conditioning[0]
        assert unconditional_guidance_scale >= 1.
        if unconditional_guidance_scale > 1.:
            assert unconditional_conditioning is not None
        device = self.model.device
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device)
        total_steps = self.ddim_timesteps.shape[0]
        print(f"Running DDIM Sampling with {total_steps} timesteps")
        if callback is not None or img_callback is not None or per_step:
            unwrap = lambda c: c[0] if isinstance(c, (list, tuple)) else (c[list(c.keys())[0]] if isinstance(c, dict) else c)
            return self._python_loop(cond, img, callback, img_callback, log_every_t, temperature, eta,
                                     unconditional_guidance_scale, unconditional_conditioning, intermediates_to_cpu,
                                     mask=mask, x0=x0, noise_dropout=noise_dropout, score_corrector=score_corrector,
                                     corrector_kwargs=corrector_kwargs, random_guiding=random_guiding, timesteps=timesteps,
                                     content_cond=None if content_cond is None else unwrap(content_cond),
                                     style_cond=None if style_cond is None else unwrap(style_cond),
                                     noise=kwargs.get("noise"), q_noise=kwargs.get("q_noise"), quantize_denoised=quantize_denoised)
        noise = kwargs.get("noise")             # [native] optional explicit per-step noise stack [S, B, C, H, W] (consumed in loop order)
        if eta != 0. and noise is None:
            noise = torch.randn((total_steps,) + tuple(shape), device=device)
        z, xi, pi = self.model.ctx.ddim_sample(S, img, cond, unconditional_conditioning if unconditional_guidance_scale > 1. else None,
                                               self.alphas_cumprod, eta=eta, scale=unconditional_guidance_scale, noise=noise,
                                               log_every_t=log_every_t, temperature=temperature, want_intermediates=True)
        mv = (lambda t: t.detach().cpu()) if intermediates_to_cpu else (lambda t: t)
        intermediates = {'x_inter': [img] + [mv(t) for t in xi], 'pred_x0': [img] + [mv(t) for t in pi]}
        return z.detach(), intermediates

    def _python_loop(self, cond, img, callback, img_callback, log_every_t, temperature, eta, scale, uc, to_cpu, mask=None,
                     x0=None, noise_dropout=0., score_corrector=None, corrector_kwargs=None, random_guiding='none',
                     timesteps=None, content_cond=None, style_cond=None, noise=None, q_noise=None, quantize_denoised=False):
        """Per-step path of ddim.py:143-209 (callbacks, inpainting mask, style / content conditioning by SNR band, timestep
        subset, noise dropout, score corrector): native UNet forward per step, torch elementwise update.  `noise` / `q_noise`
        [native]: optional explicit stacks [steps, B, C, H, W] for the update noise and for q_sample of the masked region."""
        ts_all = self.ddim_timesteps
        if timesteps is not None:                                   # ddim.py:158-160
            subset_end = int(min(timesteps / ts_all.shape[0], 1) * ts_all.shape[0]) - 1
            ts_all = ts_all[:subset_end]
        total_steps = ts_all.shape[0]
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        b = img.shape[0]
        random_guider = None
        if random_guiding != 'none':                                # drawn (RNG parity) but unused, as in the reference (:228)
            random_guider = torch.clamp(torch.randn(img.shape, device=img.device), -1., 1.)
        for i, step in enumerate(np.flip(ts_all)):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=img.device, dtype=torch.long)
            snr = self.ddim_alphas[index] / (1 - self.ddim_alphas[index])
            input_cond = cond
            if style_cond is not None and snr < 5.e-2:
                input_cond = style_cond
            if content_cond is not None and snr >= 5.e-2 and snr < 1.:
                input_cond = content_cond
            if mask is not None:
                assert x0 is not None
                img_orig = self.model.q_sample(x0, ts, noise=None if q_noise is None else q_noise[i])
                img = img_orig * mask + (1. - mask) * img
            if random_guiding == 'sampled':
                random_guider = torch.clamp(torch.randn(img.shape, device=img.device), -1., 1.)
            img, pred_x0 = self.p_sample_ddim(img, input_cond, ts, index=index, temperature=temperature, quantize_denoised=quantize_denoised,
                                              noise_dropout=noise_dropout, score_corrector=score_corrector,
                                              corrector_kwargs=corrector_kwargs, unconditional_guidance_scale=scale,
                                              unconditional_conditioning=uc, noise=None if noise is None else noise[i],
                                              random_guider=random_guider)
            if callback: callback(i)
            if img_callback: img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img.cpu() if to_cpu else img)
                intermediates['pred_x0'].append(pred_x0.cpu() if to_cpu else pred_x0)
        return img, intermediates

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, use_original_steps=False, quantize_denoised=False, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, noise=None, random_guider=None):
        b = x.shape[0]
        assert unconditional_guidance_scale >= 1.
        if use_original_steps:
            raise NotImplementedError("use_original_steps: dead in the reference (ddim.py:249)")
        if noise is None:
            noise = torch.randn(x.shape, device=x.device)
        if unconditional_guidance_scale > 1.:
            assert unconditional_conditioning is not None
            out = self.model.apply_model(torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([c, unconditional_conditioning]))
            e_t, e_u = out[:b], out[b:]
            e_t = e_u + unconditional_guidance_scale * (e_t - e_u)
        else:
            e_t = self.model.apply_model(x, t, c)
        if score_corrector is not None:
            assert getattr(self.model, "parameterization", "eps") == "eps"
            e_t = score_corrector.modify_score(self.model, e_t, x, t, c, **(corrector_kwargs or {}))
        a_t = torch.full_like(e_t, float(self.ddim_alphas[index]))
        a_prev = torch.full_like(e_t, float(self.ddim_alphas_prev[index]))
        sigma_t = torch.full_like(e_t, float(self.ddim_sigmas[index]))
        sqrt_one_minus_at = torch.full_like(e_t, float(self.ddim_sqrt_one_minus_alphas[index]))
        pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
        if quantize_denoised:                     # ddim.py:260-261: pred_x0, _, *_ = self.model.first_stage_model.quantize(pred_x0)
            pred_x0 = self.model.quantize_first_stage(pred_x0)
        dir_xt = (1. - a_prev - sigma_t ** 2).sqrt() * e_t
        noise = sigma_t * noise * temperature
        if noise_dropout > 0.:
            noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        x_prev = a_prev.sqrt() * pred_x0 + dir_xt + noise
        return x_prev, pred_x0
