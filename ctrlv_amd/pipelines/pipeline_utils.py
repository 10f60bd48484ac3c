"""Host-side pipeline plumbing shared by the two sampling pipelines.

Restates the helpers the reference inherits from diffusers' `StableVideoDiffusionPipeline` (the
reference vendors copies of `_encode_image` / `_encode_vae_image` / `_resize_with_antialiasing` at
src/ctrlv/bbox_generator_baseline/utils/image_encoder.py:111-291, which serve as the spec) and implements the
denoising loop body of pipeline_video_control.py:298-343 / pipeline_video_diffusion.py:259-293 on top of the HIP
models: the CFG combine + Euler update is one fused kernel, latents stay fp32 on the device, sigmas stay on the
host, so a step is `ControlNet fwd + UNet fwd + 1 kernel + 2 small copies` with no host synchronisation.
VAE and CLIP are caller-supplied PyTorch-ROCm modules (out of the hot path, once per clip).
"""
import math
import os
from dataclasses import dataclass
from typing import Union

import numpy as np
import torch
import torch.nn.functional as Fnn

from .. import ops

# The VAE encoder / CLIP run through MIOpen.  Its default find mode benchmarks every solver the first time a conv shape is
# seen -- minutes for the 576x1024 VAE shapes on this stack (tools/vae_decode_bench.py) -- for kernels that run once per
# clip; the heuristic ("FAST") choice costs milliseconds.  Only a default: an exported MIOPEN_FIND_MODE wins.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

try:                                  # PIL is optional: tensors are the primary input type
    import PIL.Image
except Exception:                     # pragma: no cover
    PIL = None


@dataclass
class StableVideoDiffusionPipelineOutput:
    frames: Union[list, np.ndarray, torch.Tensor]


def _append_dims(x, target_dims):
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """Generator-device aware randn (a CPU generator draws on the CPU and the sample is moved)."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    if isinstance(generator, (list, tuple)):
        if len(generator) != shape[0]:
            raise ValueError(f"got {len(generator)} generators for batch size {shape[0]}")
        return torch.cat([randn_tensor((1,) + tuple(shape[1:]), g, device, dtype) for g in generator], 0)
    gdev = generator.device if generator is not None else device
    return torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)


def _gaussian_kernel1d(ks, sigma, device, dtype):
    x = torch.arange(ks, device=device, dtype=dtype) - ks // 2
    if ks % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return g / g.sum()


def _resize_with_antialiasing(image, size, interpolation="bicubic", align_corners=True):
    """Gaussian pre-blur (sigma from the down-scale factor) followed by bicubic resize."""
    h, w = image.shape[-2:]
    factors = (h / size[0], w / size[1])
    sigmas = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2 * s, 3)) for s in sigmas]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]
    ky = _gaussian_kernel1d(ks[0], sigmas[0], image.device, image.dtype)
    kx = _gaussian_kernel1d(ks[1], sigmas[1], image.device, image.dtype)
    c = image.shape[1]
    x = Fnn.pad(image, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
    x = Fnn.conv2d(x, ky.view(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    x = Fnn.conv2d(x, kx.view(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)
    return Fnn.interpolate(x, size=size, mode=interpolation, align_corners=align_corners)


class VaeImageProcessor:
    """The subset of diffusers' VaeImageProcessor the SVD pipelines use."""

    def __init__(self, vae_scale_factor=8):
        self.vae_scale_factor = vae_scale_factor

    @staticmethod
    def pil_to_numpy(images):
        if not isinstance(images, list):
            images = [images]
        return np.stack([np.array(im).astype(np.float32) / 255.0 for im in images], axis=0)

    @staticmethod
    def numpy_to_pt(images):
        if images.ndim == 3:
            images = images[..., None]
        return torch.from_numpy(images.transpose(0, 3, 1, 2))

    @staticmethod
    def pt_to_numpy(images):
        return images.cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def numpy_to_pil(images):
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        return [PIL.Image.fromarray(im.squeeze() if im.shape[-1] == 1 else im) for im in images]

    def preprocess(self, image, height=None, width=None):
        if torch.is_tensor(image):
            x = image if image.ndim == 4 else image[None]
            x = x.float()
            if height is not None and tuple(x.shape[-2:]) != (height, width):
                x = Fnn.interpolate(x, size=(height, width), mode="bilinear", align_corners=False)
            return x if x.min() < 0 else 2.0 * x - 1.0      # tensors already in [-1, 1] are passed through
        if not isinstance(image, list):
            image = [image]
        if height is not None:
            image = [im.resize((width, height), resample=PIL.Image.LANCZOS) for im in image]
        return 2.0 * self.numpy_to_pt(self.pil_to_numpy(image)) - 1.0

    def postprocess(self, image, output_type="pil"):
        image = (image / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return image
        image = self.pt_to_numpy(image)
        return image if output_type == "np" else self.numpy_to_pil(image)


def tensor2vid(video, processor, output_type="np"):
    """(B, C, F, H, W) in [-1, 1] -> per-clip list / array / tensor of frames."""
    outputs = []
    for b in range(video.shape[0]):
        outputs.append(processor.postprocess(video[b].permute(1, 0, 2, 3), output_type))
    if output_type == "np":
        return np.stack(outputs)
    if output_type == "pt":
        return torch.stack(outputs)
    if output_type != "pil":
        raise ValueError(f"{output_type} does not exist. Please choose one of ['np', 'pt', 'pil']")
    return outputs


class _ProgressBar:
    def __init__(self, total, disable):
        self.total, self.disable, self.n = total, disable, 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, n=1):
        self.n += n


def _load_vae(path, **kw):
    """diffusers' AutoencoderKLTemporalDecoder when diffusers is installed, else ctrlv_amd's own PyTorch-ROCm module of
    the same architecture and state-dict layout (models/autoencoder_kl_temporal_decoder.py)."""
    try:
        from diffusers import AutoencoderKLTemporalDecoder
    except ImportError:
        from ..models.autoencoder_kl_temporal_decoder import AutoencoderKLTemporalDecoder
    return AutoencoderKLTemporalDecoder.from_pretrained(path, **kw)


def _load_image_encoder(path, **kw):
    from transformers import CLIPVisionModelWithProjection
    return CLIPVisionModelWithProjection.from_pretrained(path, **kw)


def _load_feature_extractor(path, **kw):
    from transformers import CLIPImageProcessor
    return CLIPImageProcessor.from_pretrained(path)


def resolve_pretrained_dir(name_or_path):
    """A local diffusers-layout directory for `name_or_path`: the path itself, or -- for a hub id such as
    "stabilityai/stable-video-diffusion-img2vid-xt" -- a copy under $CTRLV_MODEL_ROOT/<id> or the local Hugging Face
    cache (no download: there is no network path in ctrlv_amd).  None if nothing is found."""
    import os
    p = str(name_or_path)
    if os.path.isdir(p):
        return p
    root = os.environ.get("CTRLV_MODEL_ROOT")
    if root:
        for cand in (os.path.join(root, p), os.path.join(root, p.split("/")[-1])):
            if os.path.isdir(cand):
                return cand
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(p, local_files_only=True)
    except Exception:       # noqa: BLE001  (not cached / hub library absent)
        return None


class ClipJob:
    """Everything one pipeline call has prepared before its denoising loop (SVDPipelineBase.prepare_clip)."""
    __slots__ = ("height", "width", "num_frames", "decode_chunk", "clips", "device", "clip_embeds", "cond_latents",
                 "time_ids", "latents", "vae_was_fp16", "cfg")


class SVDPipelineBase:
    """Component registry + the inherited helpers + the HIP denoising loop."""

    _component_names = ("vae", "image_encoder", "unet", "controlnet", "scheduler", "feature_extractor")
    # HIP-graph replay of the two model forwards is the default execution mode of the loop (first step eager, second
    # step captured, the rest replayed; a new pipeline call = new shapes = new capture).  `pipe.use_hip_graph = False`
    # or CTRLV_HIP_GRAPH=0 selects eager launches.
    use_hip_graph = os.environ.get("CTRLV_HIP_GRAPH", "1") != "0"

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
        self._components = list(kw)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, variant=None, component_loaders=None,
                        **kwargs):
        """Build the pipeline from a diffusers-layout directory (`model_index.json`, `unet/`, `controlnet/`, `vae/`,
        `image_encoder/`, `feature_extractor/`, `scheduler/scheduler_config.json`) with component overrides, as the
        reference's tools do: `StableVideoControlPipeline.from_pretrained(id, controlnet=ctrlnet, unet=unet)`
        (tools/eval_video_controlnet.py:116-118, tools/eval_overall.py:203-217) and with every module passed in
        (tools/train_video_controlnet.py:344-353,543-552).

        * Components given as keyword arguments are used as they are.
        * `unet` / `controlnet` otherwise load through the HIP models' own `from_pretrained(dir, subfolder=...)`.
        * `scheduler` loads `scheduler/scheduler_config.json` (SVD defaults when the directory has none).
        * `vae`, `image_encoder`, `feature_extractor` are PyTorch-ROCm modules outside the hot path; they load through
          `component_loaders[name](subfolder_path, torch_dtype=..., variant=...)`, by default diffusers'
          AutoencoderKLTemporalDecoder and transformers' CLIPVisionModelWithProjection / CLIPImageProcessor.
        * A hub id resolves to a local copy only ($CTRLV_MODEL_ROOT or the local HF cache); nothing is downloaded.
        `revision`, `use_safetensors`, `local_files_only`, `cache_dir` and similar hub keywords are accepted and ignored.
        """
        import inspect
        import json
        names = [n for n in inspect.signature(cls.__init__).parameters if n != "self"]
        given = {n: kwargs.pop(n) for n in list(kwargs) if n in names}
        loaders = {"vae": _load_vae, "image_encoder": _load_image_encoder, "feature_extractor": _load_feature_extractor}
        loaders.update(component_loaders or {})
        missing = [n for n in names if n not in given]
        root = resolve_pretrained_dir(pretrained_model_name_or_path) if missing else None
        if missing and root is None and missing != ["scheduler"]:
            raise EnvironmentError(
                f"{cls.__name__}.from_pretrained: '{pretrained_model_name_or_path}' is not a local directory (and not in "
                f"$CTRLV_MODEL_ROOT / the local Hugging Face cache) but the components {missing} were not passed in; "
                "ctrlv_amd never downloads")
        if root is not None and os.path.isfile(os.path.join(root, "model_index.json")):
            with open(os.path.join(root, "model_index.json")) as f:
                index = json.load(f)
        else:
            index = {}
        comps = dict(given)
        for n in missing:
            sub = os.path.join(root, n) if root is not None else None
            if n == "scheduler":
                from ..schedulers import EulerDiscreteScheduler
                comps[n] = (EulerDiscreteScheduler.from_pretrained(root, subfolder="scheduler")
                            if sub and os.path.isdir(sub) else EulerDiscreteScheduler())
                continue
            if not os.path.isdir(sub):
                raise EnvironmentError(f"{cls.__name__}.from_pretrained: no '{n}/' under {root} and no `{n}=` override"
                                       + (f" (model_index.json lists {index[n]})" if n in index else ""))
            if n in ("unet", "controlnet"):
                from ..models import ControlNetModel, UNetSpatioTemporalConditionModel
                mcls = UNetSpatioTemporalConditionModel if n == "unet" else ControlNetModel
                comps[n] = mcls.from_pretrained(root, subfolder=n, variant=variant, torch_dtype=torch_dtype)
            else:
                comps[n] = loaders[n](sub, torch_dtype=torch_dtype, variant=variant)
        pipe = cls(**comps)
        if torch_dtype is not None:
            for n in ("vae", "image_encoder"):
                m = comps.get(n)
                if n in missing and isinstance(m, torch.nn.Module):
                    m.to(torch_dtype)
        return pipe

    # ---- device / dtype plumbing -----------------------------------------------------------------------------
    def to(self, *args, **kwargs):
        for name in self._components:
            m = getattr(self, name, None)
            if isinstance(m, torch.nn.Module):
                m.to(*args, **kwargs)
        return self

    @property
    def device(self):
        return self.unet.device

    _execution_device = device

    @property
    def dtype(self):
        return self.unet.dtype

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    def progress_bar(self, total=None):
        return _ProgressBar(total, getattr(self, "_progress_bar_config", {}).get("disable", False))

    def maybe_free_model_hooks(self):
        pass

    def save_pretrained(self, save_directory, safe_serialization=True, **kw):
        """Writes the HIP models in diffusers layout (`unet/`, `controlnet/`), as pipeline.save_pretrained does
        at tools/train_video_controlnet.py:553; VAE / CLIP are the caller's modules and keep their own savers."""
        for name in ("unet", "controlnet"):
            m = getattr(self, name, None)
            if m is not None:
                m.save_pretrained(os.path.join(save_directory, name), safe_serialization=safe_serialization)
        for name in ("vae", "image_encoder", "feature_extractor", "scheduler"):
            m = getattr(self, name, None)
            if m is not None and hasattr(m, "save_pretrained"):
                m.save_pretrained(os.path.join(save_directory, name))
        import json
        index = {"_class_name": type(self).__name__, "_diffusers_version": "0.27.2"}
        for name in self._components:
            m = getattr(self, name, None)
            if m is not None:
                index[name] = [type(m).__module__.split(".")[0], type(m).__name__]
        with open(os.path.join(save_directory, "model_index.json"), "w") as f:
            json.dump(index, f, indent=2)

    # ---- guidance ------------------------------------------------------------------------------------------------
    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        if isinstance(self.guidance_scale, (int, float)):
            return self.guidance_scale > 1
        return self.guidance_scale.max() > 1

    @property
    def num_timesteps(self):
        return self._num_timesteps

    # ---- encoders (once per clip; PyTorch-ROCm modules) ----------------------------------------------------------
    def _encode_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance):
        dtype = next(self.image_encoder.parameters()).dtype
        if not isinstance(image, torch.Tensor):
            image = self.image_processor.pil_to_numpy(image)
            image = self.image_processor.numpy_to_pt(image)
            # normalise before resizing, un-normalise after (matches the original implementation)
            image = image * 2.0 - 1.0
            image = _resize_with_antialiasing(image, (224, 224))
            image = (image + 1.0) / 2.0
        image = self.feature_extractor(images=image, do_normalize=True, do_center_crop=False, do_resize=False,
                                       do_rescale=False, return_tensors="pt").pixel_values
        image = image.to(device=device, dtype=dtype)
        image_embeddings = self.image_encoder(image).image_embeds
        image_embeddings = image_embeddings.unsqueeze(1)
        bs_embed, seq_len, _ = image_embeddings.shape
        image_embeddings = image_embeddings.repeat(1, num_videos_per_prompt, 1)
        image_embeddings = image_embeddings.view(bs_embed * num_videos_per_prompt, seq_len, -1)
        if do_classifier_free_guidance:
            image_embeddings = torch.cat([torch.zeros_like(image_embeddings), image_embeddings])
        return image_embeddings

    def _encode_vae_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance):
        image = image.to(device=device)
        image_latents = self.vae.encode(image).latent_dist.mode()          # NOT multiplied by scaling_factor
        if do_classifier_free_guidance:
            image_latents = torch.cat([torch.zeros_like(image_latents), image_latents])
        return image_latents.repeat(num_videos_per_prompt, 1, 1, 1)

    def _encode_vae_condition(self, cond_image, device, num_videos_per_prompt, do_classifier_free_guidance):
        """pipeline_video_control.py:71-101: bbox frames (3 channels) go through the VAE, latents (4) pass through."""
        video_length = cond_image.shape[1]
        cond_image = cond_image.to(device=device).to(dtype=self.vae.dtype)
        if cond_image.shape[2] == 3:
            b = cond_image.shape[0]
            cond_em = self.vae.encode(cond_image.flatten(0, 1)).latent_dist.mode()
            cond_em = cond_em.reshape(b, video_length, *cond_em.shape[1:])
        else:
            assert cond_image.shape[2] == 4, \
                "The input tensor should have 3 or 4 channels. 3 for frames and 4 for latents."
            cond_em = cond_image
        cond_em = cond_em.repeat(num_videos_per_prompt, 1, 1, 1, 1)
        if do_classifier_free_guidance:
            cond_em = torch.cat([torch.zeros_like(cond_em), cond_em])
        return cond_em

    def _get_add_time_ids(self, fps, motion_bucket_id, noise_aug_strength, dtype, batch_size, num_videos_per_prompt,
                          do_classifier_free_guidance):
        add_time_ids = [fps, motion_bucket_id, noise_aug_strength]
        passed_add_embed_dim = self.unet.config.addition_time_embed_dim * len(add_time_ids)
        expected_add_embed_dim = self.unet.add_embedding.linear_1.in_features
        if expected_add_embed_dim != passed_add_embed_dim:
            raise ValueError(
                f"Model expects an added time embedding vector of length {expected_add_embed_dim}, but a vector of "
                f"{passed_add_embed_dim} was created. The model has an incorrect config. Please check "
                "`unet.config.time_embedding_type` and `text_encoder_2.config.projection_dim`.")
        add_time_ids = torch.tensor([add_time_ids], dtype=dtype)
        add_time_ids = add_time_ids.repeat(batch_size * num_videos_per_prompt, 1)
        if do_classifier_free_guidance:
            add_time_ids = torch.cat([add_time_ids, add_time_ids])
        return add_time_ids

    def prepare_latents(self, batch_size, num_frames, num_channels_latents, height, width, dtype, device, generator,
                        latents=None):
        shape = (batch_size, num_frames, num_channels_latents // 2, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                             f"effective batch size of {batch_size}. Make sure the batch size matches the length of "
                             "the generators.")
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def decode_latents(self, latents, num_frames, decode_chunk_size=14):
        latents = latents.flatten(0, 1)
        latents = 1 / self.vae.config.scaling_factor * latents
        frames = []
        for i in range(0, latents.shape[0], decode_chunk_size):
            num_frames_in = latents[i:i + decode_chunk_size].shape[0]
            frames.append(self.vae.decode(latents[i:i + decode_chunk_size], num_frames=num_frames_in).sample)
        frames = torch.cat(frames, dim=0)
        frames = frames.reshape(-1, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4)
        return frames.float()

    # ---- the hot loop --------------------------------------------------------------------------------------------
    # ------------------------------------------------------------------------------------------- call scaffolding
    def prepare_clip(self, image, check, *, height, width, num_frames, decode_chunk_size, fps, motion_bucket_id,
                     noise_aug_strength, num_videos_per_prompt, generator, latents, latent_channels,
                     min_guidance_scale, max_guidance_scale, num_inference_steps):
        """The part of a pipeline call that precedes the loop, shared by both pipelines: resolve sizes, validate
        (`check(height, width)` = the pipeline's check_inputs), CLIP-embed the first frame, noise-augment and VAE-encode it,
        micro-conditioning ids, timestep schedule, initial noise and the per-frame guidance ramp.  Behaviour follows
        pipeline_video_control.py:196-292 / pipeline_video_diffusion.py:131-228; in particular the two random draws happen
        in the reference's order (augmentation noise first, initial latents second), so a seeded generator gives the
        reference's sample.  Returns a ClipJob."""
        job = ClipJob()
        unet_cfg = self.unet.config
        job.height = height or unet_cfg.sample_size * self.vae_scale_factor
        job.width = width or unet_cfg.sample_size * self.vae_scale_factor
        job.num_frames = unet_cfg.num_frames if num_frames is None else num_frames
        job.decode_chunk = job.num_frames if decode_chunk_size is None else decode_chunk_size
        check(job.height, job.width)
        if PIL is not None and isinstance(image, PIL.Image.Image):
            job.clips = 1
        else:
            job.clips = len(image) if isinstance(image, list) else image.shape[0]
        dev = job.device = self._execution_device
        self._guidance_scale = max_guidance_scale              # scalar first: do_classifier_free_guidance reads it
        # ONE classifier-free-guidance decision per call, taken from the scalar like the reference's encoders do
        # (pipeline_video_control.py:206-284 run before the per-frame ramp is installed, :287-290): every conditioning
        # tensor and the loop use it -- no device read of ramp.max(), no half-doubled inputs when min > 1 >= max
        cfg = job.cfg = bool(self.do_classifier_free_guidance)
        job.clip_embeds = self._encode_image(image, dev, num_videos_per_prompt, cfg)
        dt = job.clip_embeds.dtype
        # first frame -> (noise-augmented) VAE latent, repeated over the frames
        pixels = self.image_processor.preprocess(image, height=job.height, width=job.width).to(dev)
        pixels = pixels + noise_aug_strength * randn_tensor(pixels.shape, generator=generator, device=dev,
                                                             dtype=pixels.dtype)                  # random draw 1
        job.vae_was_fp16 = self.vae.dtype == torch.float16 and getattr(self.vae.config, "force_upcast", False)
        if job.vae_was_fp16:
            self.vae.to(dtype=torch.float32)
        first = self._encode_vae_image(pixels.to(self.vae.dtype), device=dev, num_videos_per_prompt=num_videos_per_prompt,
                                       do_classifier_free_guidance=cfg)
        job.cond_latents = first.to(dt).unsqueeze(1).repeat(1, job.num_frames, 1, 1, 1)
        # SVD is conditioned on fps - 1
        job.time_ids = self._get_add_time_ids(fps - 1, motion_bucket_id, noise_aug_strength, dt, job.clips,
                                              num_videos_per_prompt, cfg).to(dev)
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        self._num_timesteps = len(self.scheduler.timesteps)
        n = job.clips * num_videos_per_prompt
        job.latents = self.prepare_latents(n, job.num_frames, latent_channels, job.height, job.width, dt, dev, generator,
                                           latents)                                               # random draw 2
        ramp = torch.linspace(min_guidance_scale, max_guidance_scale, job.num_frames, device="cpu")
        self._guidance_scale = ramp.to(dev, job.latents.dtype).unsqueeze(0).repeat(n, 1)[:, :, None, None, None]
        return job

    def finish_clip(self, job, latents, output_type, return_dict, clamp=False):
        """Decode (unless latents are asked for), post-process, restore the VAE dtype, wrap the output."""
        frames = latents
        if output_type != "latent":
            frames = self.decode_latents(latents.to(self.vae.dtype), job.num_frames, job.decode_chunk)
            if clamp:
                frames = torch.clamp(frames, -1, 1)
            frames = tensor2vid(frames, self.image_processor, output_type=output_type)
        if job.vae_was_fp16:
            self.vae.to(dtype=torch.float16)
        self.maybe_free_model_hooks()
        return StableVideoDiffusionPipelineOutput(frames=frames) if return_dict else frames

    def _denoise(self, latents, image_latents, image_embeddings, added_time_ids, cond_em, num_inference_steps,
                 min_guidance_scale, max_guidance_scale, control_condition_scale, callback_on_step_end,
                 callback_on_step_end_tensor_inputs, progress_bar, do_cfg=None):
        """Loop of pipeline_video_control.py:298-343.  `latents` (B, F, 4, h, w) any float dtype."""
        model_dtype = image_embeddings.dtype
        controlnet = getattr(self, "controlnet", None) if cond_em is not None else None
        stepper = DenoiseStepper(self.unet, controlnet, self.scheduler, latents, image_latents, image_embeddings,
                                 added_time_ids, cond_em, min_guidance_scale, max_guidance_scale,
                                 control_condition_scale,
                                 do_cfg=bool(self.do_classifier_free_guidance) if do_cfg is None else bool(do_cfg),
                                 use_hip_graph=bool(self.use_hip_graph))
        for i, t in enumerate(self.scheduler.timesteps):
            noise_pred = stepper.step(i)
            if callback_on_step_end is not None:
                latents_cb = stepper.latents.to(model_dtype)
                local = {"latents": latents_cb, "noise_pred": noise_pred, "image_latents": image_latents}
                callback_kwargs = {k: local[k] for k in callback_on_step_end_tensor_inputs}
                callback_outputs = callback_on_step_end(self, i, t, callback_kwargs)
                new = callback_outputs.pop("latents", latents_cb)
                if new is not latents_cb:
                    stepper.set_latents(new, i + 1)
            progress_bar.update()
        return stepper.latents.to(model_dtype)


class DenoiseStepper:
    """One scheduler iteration for a batch of clips = the unit of the headline metric ("denoising step"):
    scale + concat (pipeline_video_control.py:300-304) -> ControlNet forward (:305-313) -> UNet forward (:316-324)
    -> CFG combine + Euler update (:327-332, one fused kernel).  Latents stay fp32 on the device and sigmas on the
    host, so a step never synchronises.  With `use_hip_graph` the two model forwards are captured once into a HIP
    graph (static input buffers, timestep read from device memory) and replayed every step."""

    def __init__(self, unet, controlnet, scheduler, latents, image_latents, image_embeddings, added_time_ids,
                 cond_em, min_guidance_scale=1.0, max_guidance_scale=3.0, control_condition_scale=1.0, do_cfg=True,
                 use_hip_graph=False):
        self.unet, self.controlnet, self.scheduler = unet, controlnet, scheduler
        device = latents.device
        self.model_dtype = image_embeddings.dtype
        B, F = latents.shape[:2]
        self.B, self.F, self.do_cfg = B, F, do_cfg
        self.guidance = torch.linspace(min_guidance_scale, max_guidance_scale, F, dtype=torch.float32, device=device)
        self.latents = latents.to(torch.float32).contiguous().clone()
        nb = 2 * B if do_cfg else B
        self.c_lat = self.latents.shape[2]
        # scaled model input | image latents on the channel dim (pipeline_video_control.py:300-304)
        self.lmi = torch.empty(nb, F, self.c_lat + image_latents.shape[2], *self.latents.shape[3:],
                               dtype=self.model_dtype, device=device)
        self.lmi[:, :, self.c_lat:] = image_latents.to(self.model_dtype)
        self.image_embeddings, self.added_time_ids, self.cond_em = image_embeddings, added_time_ids, cond_em
        self.control_scale = control_condition_scale
        # next scaled model input, written by the fused Euler kernel in the models' element type
        self.scaled = torch.empty(B, F, *self.latents.shape[2:], device=device,
                                  dtype=torch.float16 if self.model_dtype == torch.float16 else torch.bfloat16)
        self.set_latents(self.latents, 0)
        self.t_dev = torch.zeros((), dtype=torch.float32, device=device)
        self.use_hip_graph = use_hip_graph
        self.overlap_controlnet = os.environ.get("CTRLV_OVERLAP", "1") != "0"
        # measured on MI355X: four concurrent half-batch forwards are 9 % SLOWER than two full-batch ones (258.8 ->
        # 282.0 ms/step): twice the kernels, every weight tile fetched twice, L2 shared four ways.  Opt-in only.
        self.split_cfg = (self.overlap_controlnet and os.environ.get("CTRLV_SPLIT_CFG", "0") == "1"
                          and getattr(unet, "time_context_order", "sb") == "bs")
        self._side = None
        self._graph = None
        self._noise_pred = None
        self._eager_runs = 0

    def set_latents(self, latents, sigma_index):
        """Copies into the stepper's own fp32 buffer: the caller's tensor is never aliased (the fused Euler kernel
        updates `self.latents` in place) and the buffer address stays stable for HIP-graph replay."""
        if latents is not self.latents:
            self.latents.copy_(latents.to(device=self.latents.device, dtype=torch.float32))
        s = self.scheduler.sigma_at(sigma_index)
        self.scaled.copy_(self.latents / math.sqrt(s ** 2 + 1))

    def _forward(self, t, overlap=False, split=False):
        """ControlNet forward then UNet forward.

        `overlap` (HIP-graph mode): the ControlNet runs on a side stream concurrently with the UNet's down / mid blocks
        -- the two are independent until the residual adds of unet_spatio_temporal_condition.py:119-127 -- and the UNet
        joins the side stream right before it consumes the residuals.  `split` (experiment, off by default): the two
        classifier-free-guidance halves of the batch run as separate forwards, with `overlap` on their own streams.
        NOTE: that is bit-identical to the batched forward only with time_context_order == "bs"; under the default
        diffusers-0.27.2 ordering quirk ("sb") the temporal cross-attention of one half reads the other half's
        embedding, which a split forward cannot reproduce.  The GEMMs are persistent kernels with one
        workgroup per CU, so when one forward's kernel reaches its last, partially filled round of tiles (12 % of the
        N = C layers: 1800 tiles on 256 CUs) the idle CUs pick up another forward's workgroups instead of waiting."""
        nb = self.lmi.shape[0]
        parts = [(0, nb)]
        if split and nb >= 2 and nb % 2 == 0:
            parts = [(0, nb // 2), (nb // 2, nb)]
        main = torch.cuda.current_stream()
        if overlap and self._side is None:
            self._side = [torch.cuda.Stream(device=self.latents.device) for _ in range(3)]
        outs = []
        try:
            for k, (a, b) in enumerate(parts):
                su = main if (k == 0 or not overlap) else self._side[0]           # UNet stream of this part
                sc = su if not overlap else self._side[1 + k]                     # its ControlNet stream
                if overlap:
                    if su is not main:
                        su.wait_stream(main)                                      # fork (inputs written on main)
                    sc.wait_stream(main)
                down = mid = None
                if self.controlnet is not None:
                    self.controlnet._lane = k
                    with torch.cuda.stream(sc):
                        down, mid = self.controlnet(
                            self.lmi[a:b], timestep=t, encoder_hidden_states=self.image_embeddings[a:b],
                            added_time_ids=self.added_time_ids[a:b], control_cond=self.cond_em[a:b],
                            conditioning_scale=self.control_scale, return_dict=False)
                self.unet._lane = k
                ev = None
                if sc is not su:      # the UNet waits for the ControlNet right before it reads the residuals
                    ev = torch.cuda.Event()
                    ev.record(sc)
                self.unet._residual_event = ev
                with torch.cuda.stream(su):
                    outs.append(self.unet(sample=self.lmi[a:b], timestep=t,
                                          encoder_hidden_states=self.image_embeddings[a:b],
                                          added_time_ids=self.added_time_ids[a:b],
                                          down_block_additional_residuals=down, mid_block_additional_residuals=mid,
                                          return_dict=False)[0])
                if su is not main:
                    main.wait_stream(su)                                          # join
        finally:
            self.unet._residual_event = None
            self.unet._lane = 0
            if self.controlnet is not None:
                self.controlnet._lane = 0
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    def step(self, i):
        B, c = self.B, self.c_lat
        self.lmi[:B, :, :c] = self.scaled
        if self.do_cfg:
            self.lmi[B:, :, :c] = self.scaled
        t = self.scheduler.timesteps[i]
        if not self.use_hip_graph:
            noise_pred = self._forward(t)
        else:
            self.t_dev.copy_(t)
            if self._graph is None and self._eager_runs < 1:
                # warm-up: sizes the workspaces of every lane, packs weights
                noise_pred = self._forward(self.t_dev, split=self.split_cfg)
                self._eager_runs += 1
            else:
                if self._graph is None:
                    torch.cuda.synchronize()
                    self._graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph):
                        self._noise_pred = self._forward(self.t_dev, overlap=self.overlap_controlnet,
                                                         split=self.split_cfg)
                self._graph.replay()
                noise_pred = self._noise_pred
        ops.cfg_euler_step(self.latents, noise_pred.contiguous(), self.guidance, self.scheduler.sigma_at(i),
                           self.scheduler.sigma_at(i + 1), self.scaled)
        return noise_pred
