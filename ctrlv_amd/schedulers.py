"""EulerDiscreteScheduler (SVD-XT configuration) -- host-side scheduling for the MI355X pipelines.

Counterpart of diffusers==0.27.2 `EulerDiscreteScheduler` as used by
/root/reference/src/ctrlv/pipelines/pipeline_video_control.py:259,301,332 (spec: SURVEY.md A.8; the update formulas
are cross-checked by the reference's training code tools/train_video_controlnet.py:405-410,468-471).
sigmas are kept as Python floats on the host so the sampling loop never synchronises with the device; the
per-step update itself is the fused HIP kernel `ctrlv_cfg_euler_step` (see pipelines/).
"""
import math

import numpy as np
import torch


class EulerDiscreteSchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class EulerDiscreteScheduler:
    order = 1
    _defaults = dict(
        num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
        prediction_type="v_prediction", interpolation_type="linear", use_karras_sigmas=True,
        sigma_min=0.002, sigma_max=700.0, timestep_spacing="leading", timestep_type="continuous", steps_offset=1,
    )

    def __init__(self, **kwargs):
        cfg = dict(self._defaults)
        unknown = set(kwargs) - set(cfg)
        if unknown:
            raise ValueError(f"unknown scheduler options {sorted(unknown)}")
        cfg.update(kwargs)
        if cfg["beta_schedule"] != "scaled_linear" or cfg["prediction_type"] != "v_prediction" or \
                cfg["timestep_type"] != "continuous" or not cfg["use_karras_sigmas"]:
            raise NotImplementedError("only the SVD configuration (scaled_linear, v_prediction, continuous, karras) "
                                      "is implemented")
        self.config = type("Config", (dict,), {"__getattr__": dict.__getitem__})(cfg)
        betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, cfg["num_train_timesteps"],
                               dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).flip(0)
        self.timesteps = 0.25 * sigmas.log()
        self.sigmas = torch.cat([sigmas, torch.zeros(1)])
        self.num_inference_steps = None
        self._step_index = None

    # ---- diffusers-style (de)serialisation: scheduler/scheduler_config.json -----------------------------------------
    @classmethod
    def from_config(cls, config, **overrides):
        """Accepts a diffusers scheduler config dict; keys this scheduler does not model (`_class_name`,
        `trained_betas`, `rescale_betas_zero_snr`, ...) are dropped, unsupported VALUES still raise in __init__."""
        cfg = {k: v for k, v in dict(config).items() if k in cls._defaults}
        cfg.update({k: v for k, v in overrides.items() if k in cls._defaults})
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, **kwargs):
        import json
        import os
        root = str(pretrained_model_name_or_path)
        if subfolder:
            root = os.path.join(root, subfolder)
        path = os.path.join(root, "scheduler_config.json")
        if not os.path.isfile(path):
            raise EnvironmentError(f"{path} not found (ctrlv_amd loads schedulers from a local directory only)")
        with open(path) as f:
            return cls.from_config(json.load(f), **kwargs)

    def save_pretrained(self, save_directory, **_):
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        cfg = {"_class_name": "EulerDiscreteScheduler", "_diffusers_version": "0.27.2"}
        cfg.update(self.config)
        with open(os.path.join(save_directory, "scheduler_config.json"), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)

    @property
    def init_noise_sigma(self):
        max_sigma = float(self.sigmas.max())
        if self.config["timestep_spacing"] in ("linspace", "trailing"):
            return max_sigma
        return (max_sigma ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        self.num_inference_steps = num_inference_steps
        rho = 7.0
        ramp = np.linspace(0, 1, num_inference_steps)
        min_inv_rho, max_inv_rho = c["sigma_min"] ** (1 / rho), c["sigma_max"] ** (1 / rho)
        sig = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
        sigmas = torch.from_numpy(sig).to(dtype=torch.float32)
        self.timesteps = (0.25 * sigmas.log()).to(device=device)
        self.sigmas = torch.cat([sigmas, torch.zeros(1)])          # kept on the host
        self._sigmas_host = [float(s) for s in self.sigmas]
        self._timesteps_host = [float(t) for t in self.timesteps.cpu()]
        self._step_index = None

    def _init_step_index(self, timestep):
        t = float(timestep)
        idx = [i for i, v in enumerate(self._timesteps_host) if v == t]
        if not idx:
            idx = [int(np.argmin([abs(v - t) for v in self._timesteps_host]))]
        self._step_index = idx[1] if len(idx) > 1 else idx[0]

    def sigma_at(self, i):
        return self._sigmas_host[i]

    def scale_model_input(self, sample, timestep):
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self._sigmas_host[self._step_index]
        return sample / math.sqrt(sigma ** 2 + 1)

    def step(self, model_output, timestep, sample, return_dict=True):
        """Generic (non-fused) Euler update on tensors of any device; the pipelines use the fused HIP kernel."""
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self._sigmas_host[self._step_index]
        sigma_next = self._sigmas_host[self._step_index + 1]
        x = sample.to(torch.float32)
        pred_original_sample = model_output.to(torch.float32) * (-sigma / math.sqrt(sigma ** 2 + 1)) + x / (sigma ** 2 + 1)
        derivative = (x - pred_original_sample) / sigma
        prev_sample = (x + derivative * (sigma_next - sigma)).to(model_output.dtype)
        self._step_index += 1
        if not return_dict:
            return (prev_sample,)
        return EulerDiscreteSchedulerOutput(prev_sample, pred_original_sample)
