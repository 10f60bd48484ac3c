"""ORACLE (test infrastructure only).  PARITY UNPINNED (see blocks.py).

CPU restatement of diffusers==0.27.2 `EulerDiscreteScheduler` with the SVD-XT scheduler_config.json values
 and of the sampling-loop body of
/root/reference/src/ctrlv/pipelines/pipeline_video_control.py:287-343.

Cross-checked against the reference's own training-side formulas:
  input scaling 1/sqrt(sigma^2+1) ....... tools/train_video_controlnet.py:410
  c_out = -sigma/sqrt(sigma^2+1), c_skip = 1/(sigma^2+1) ... tools/train_video_controlnet.py:468-470
"""
import numpy as np
import torch

SVD_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    prediction_type="v_prediction", interpolation_type="linear", use_karras_sigmas=True,
    sigma_min=0.002, sigma_max=700.0, timestep_spacing="leading", timestep_type="continuous", steps_offset=1,
)


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, **overrides):
        self.config = dict(SVD_SCHEDULER_CONFIG, **overrides)
        c = self.config
        betas = torch.linspace(c["beta_start"] ** 0.5, c["beta_end"] ** 0.5, c["num_train_timesteps"],
                               dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).flip(0)
        self.timesteps = torch.Tensor([0.25 * s.log() for s in sigmas])
        self.sigmas = torch.cat([sigmas, torch.zeros(1)])
        self.num_inference_steps = None
        self._step_index = None

    @property
    def init_noise_sigma(self):
        max_sigma = max(self.sigmas) if isinstance(self.sigmas, list) else self.sigmas.max()
        if self.config["timestep_spacing"] in ("linspace", "trailing"):
            return max_sigma
        return (max_sigma ** 2 + 1) ** 0.5

    def _convert_to_karras(self, in_sigmas, num_inference_steps):
        sigma_min = self.config.get("sigma_min", None)
        sigma_max = self.config.get("sigma_max", None)
        sigma_min = sigma_min if sigma_min is not None else in_sigmas[-1].item()
        sigma_max = sigma_max if sigma_max is not None else in_sigmas[0].item()
        rho = 7.0
        ramp = np.linspace(0, 1, num_inference_steps)
        min_inv_rho = sigma_min ** (1 / rho)
        max_inv_rho = sigma_max ** (1 / rho)
        return (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        self.num_inference_steps = num_inference_steps
        step_ratio = c["num_train_timesteps"] // num_inference_steps          # "leading"
        timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.float32)
        timesteps += c["steps_offset"]
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)      # interpolation_type == "linear"
        sigmas = self._convert_to_karras(in_sigmas=sigmas, num_inference_steps=num_inference_steps)
        sigmas = torch.from_numpy(sigmas).to(dtype=torch.float32, device=device)
        # timestep_type == "continuous" and prediction_type == "v_prediction"
        self.timesteps = torch.Tensor([0.25 * s.log() for s in sigmas]).to(device=device)
        self.sigmas = torch.cat([sigmas, torch.zeros(1, device=sigmas.device)])
        self._step_index = None

    def _init_step_index(self, timestep):
        idx = (self.timesteps == timestep).nonzero()
        self._step_index = idx[1 if len(idx) > 1 else 0].item()

    def scale_model_input(self, sample, timestep):
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self._step_index]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample):
        if self._step_index is None:
            self._init_step_index(timestep)
        sample = sample.to(torch.float32)
        sigma = self.sigmas[self._step_index]
        sigma_hat = sigma                                                    # s_churn = 0 -> gamma = 0
        pred_original_sample = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + (sample / (sigma ** 2 + 1))
        derivative = (sample - pred_original_sample) / sigma_hat
        dt = self.sigmas[self._step_index + 1] - sigma_hat
        prev_sample = (sample + derivative * dt).to(model_output.dtype)
        self._step_index += 1
        return prev_sample


def guidance_scale_tensor(min_guidance_scale, max_guidance_scale, num_frames, batch, dtype=torch.float32):
    """pipeline_video_control.py:287-290."""
    g = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames).unsqueeze(0).to(dtype)
    g = g.repeat(batch, 1)
    return g[:, :, None, None, None]


@torch.no_grad()
def sample_loop(unet, controlnet, scheduler, latents, image_latents, image_embeddings, added_time_ids, cond_em,
                num_inference_steps, min_guidance_scale=1.0, max_guidance_scale=3.0, control_condition_scale=1.0,
                record=None):
    """Denoising loop of StableVideoControlPipeline.__call__ (pipeline_video_control.py:298-343) on prepared
    inputs; `controlnet=None` gives the VideoDiffusionPipeline loop (pipeline_video_diffusion.py:259-293).
    `image_latents`, `image_embeddings`, `added_time_ids`, `cond_em` already carry the CFG doubling."""
    scheduler.set_timesteps(num_inference_steps)
    do_cfg = max_guidance_scale > 1.0
    guidance = guidance_scale_tensor(min_guidance_scale, max_guidance_scale, latents.shape[1], latents.shape[0],
                                     latents.dtype)
    for t in scheduler.timesteps:
        latent_model_input = torch.cat([latents] * 2) if do_cfg else latents
        latent_model_input = scheduler.scale_model_input(latent_model_input, t)
        latent_model_input = torch.cat([latent_model_input, image_latents], dim=2)
        down = mid = None
        if controlnet is not None:
            down, mid = controlnet(latent_model_input, t, encoder_hidden_states=image_embeddings,
                                   added_time_ids=added_time_ids, control_cond=cond_em,
                                   conditioning_scale=control_condition_scale)
        noise_pred = unet(latent_model_input, t, encoder_hidden_states=image_embeddings,
                          added_time_ids=added_time_ids, down_block_additional_residuals=down,
                          mid_block_additional_residuals=mid)[0]
        if do_cfg:
            noise_pred_uncond, noise_pred_cond = noise_pred.chunk(2)
            noise_pred = noise_pred_uncond + guidance * (noise_pred_cond - noise_pred_uncond)
        latents = scheduler.step(noise_pred, t, latents)
        if record is not None:
            record.append(latents.clone())
    return latents
