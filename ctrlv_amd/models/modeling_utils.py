"""Minimal diffusers-compatible model surface (config.json + diffusion_pytorch_model.{safetensors,bin}).

Covers what the reference's callers touch on the two models (SURVEY.md 8b "other module surface"):
`from_pretrained(path, subfolder=..., variant=..., low_cpu_mem_usage=..., **config_overrides)`
(tools/train_video_controlnet.py:106-109, tools/eval_video_controlnet.py:114-115), `save_pretrained`
(tools/train_video_controlnet.py:151-179), `.config.<attr>`, `register_to_config`, `.dtype`, `.device`.
State-dict keys are the diffusers names, so real SVD-XT / Ctrl-V checkpoints load unchanged.
"""
import json
import os

import torch
from torch import nn

WEIGHTS_NAME = "diffusion_pytorch_model.bin"
SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
CONFIG_NAME = "config.json"


class FrozenConfig(dict):
    """dict with attribute access (diffusers' FrozenDict behaviour that callers rely on: `unet.config.num_frames`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        raise AttributeError("config is frozen; use register_to_config(...)")


class HipModelMixin(nn.Module):
    config_name = CONFIG_NAME
    _class_name = None

    def register_to_config(self, **kwargs):
        cfg = dict(getattr(self, "_config", {}))
        cfg.update(kwargs)
        object.__setattr__(self, "_config", FrozenConfig(cfg))

    @property
    def config(self):
        return self._config

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    # --- (de)serialisation ------------------------------------------------------------------------------------
    def save_pretrained(self, save_directory, safe_serialization=True, variant=None, **_):
        os.makedirs(save_directory, exist_ok=True)
        cfg = {"_class_name": self._class_name or type(self).__name__, "_diffusers_version": "0.27.2"}
        cfg.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()})
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        name = SAFETENSORS_WEIGHTS_NAME if safe_serialization else WEIGHTS_NAME
        if variant:
            stem, ext = name.rsplit(".", 1)
            name = f"{stem}.{variant}.{ext}"
        path = os.path.join(save_directory, name)
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, path, metadata={"format": "pt"})
        else:
            torch.save(sd, path)

    @classmethod
    def from_config(cls, config, **overrides):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(overrides)
        import inspect
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        return cls(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in accepted})

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, variant=None, torch_dtype=None,
                        low_cpu_mem_usage=None, **config_overrides):
        root = str(pretrained_model_name_or_path)
        if subfolder:
            root = os.path.join(root, subfolder)
        cfg_path = os.path.join(root, CONFIG_NAME)
        if not os.path.isfile(cfg_path):
            raise EnvironmentError(
                f"{cfg_path} not found.  ctrlv_amd loads models from a local diffusers-layout directory only "
                "(there is no hub download).")
        with open(cfg_path) as f:
            config = json.load(f)
        model = cls.from_config(config, **config_overrides)
        cands = []
        for base in (SAFETENSORS_WEIGHTS_NAME, WEIGHTS_NAME):
            if variant:
                stem, ext = base.rsplit(".", 1)
                cands.append(f"{stem}.{variant}.{ext}")
            cands.append(base)
        for name in cands:
            path = os.path.join(root, name)
            if os.path.isfile(path):
                if name.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd = load_file(path)
                else:
                    sd = torch.load(path, map_location="cpu", weights_only=True)
                break
        else:
            raise EnvironmentError(f"no weights file ({' / '.join(cands)}) under {root}")
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing or unexpected:
            raise ValueError(f"{cls.__name__}.from_pretrained: missing keys {missing[:5]}... unexpected {unexpected[:5]}...")
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        model.eval()
        return model

    # --- training-side hooks the reference calls --------------------------------------------------------------------
    gradient_checkpointing = False

    def enable_gradient_checkpointing(self):
        """tools/train_video_controlnet.py:185-186.  ctrlv_amd.training builds this model's training forward with its GEGLU
        feed-forwards CHECKPOINTED: their 4C / 8C-wide intermediates (31 GB of a 101 GB step at the reference's size) are
        recomputed in the backward instead of kept (ctrlv_amd/autograd.py)."""
        self.gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self.gradient_checkpointing = False

    def enable_xformers_memory_efficient_attention(self, *_, **__):
        pass  # attention is always the fused HIP kernel
