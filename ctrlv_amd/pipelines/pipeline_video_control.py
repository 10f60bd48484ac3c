"""StableVideoControlPipeline (Box2Video sampling) on MI355X.

Counterpart of /root/reference/src/ctrlv/pipelines/pipeline_video_control.py:25-360 with the same constructor
(:27-46), `check_inputs` (:51-68), `_encode_vae_condition` (:71-101) and `__call__` signature / defaults / output
types (:105-127, :345-360).  The denoising loop (:298-343) runs on the HIP models; see pipeline_utils._denoise.
"""
from typing import Callable, Dict, List, Optional, Union

import torch

from .pipeline_utils import (PIL, StableVideoDiffusionPipelineOutput, SVDPipelineBase, VaeImageProcessor,
                             randn_tensor, tensor2vid)


class StableVideoControlPipeline(SVDPipelineBase):
    model_cpu_offload_seq = "image_encoder->unet->vae"
    _callback_tensor_inputs = ["latents"]

    def __init__(self, vae, image_encoder, unet, controlnet, scheduler, feature_extractor):
        self.register_modules(vae=vae, image_encoder=image_encoder, controlnet=controlnet, unet=unet,
                              scheduler=scheduler, feature_extractor=feature_extractor)
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor)

    def check_inputs(self, image, cond_images, height, width):
        pil_ok = PIL is not None and isinstance(image, PIL.Image.Image)
        if not isinstance(image, torch.Tensor) and not pil_ok and not isinstance(image, list):
            raise ValueError(
                "`image` has to be of type `torch.FloatTensor` or `PIL.Image.Image` or `List[PIL.Image.Image]` but is"
                f" {type(image)}")
        if not isinstance(cond_images, torch.Tensor):
            raise ValueError(
                "`cond_images` has to be of type `torch.FloatTensor` or `PIL.Image.Image` or `List[PIL.Image.Image]` but is"
                f" {type(cond_images)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

    @torch.no_grad()
    def __call__(
        self,
        image,
        cond_images: torch.FloatTensor = None,
        height: int = 576,
        width: int = 1024,
        num_frames: Optional[int] = None,
        num_inference_steps: int = 25,
        min_guidance_scale: float = 1.0,
        max_guidance_scale: float = 3.0,
        control_condition_scale: float = 1.0,
        fps: int = 7,
        motion_bucket_id: int = 127,
        noise_aug_strength: float = 0.02,
        decode_chunk_size: Optional[int] = None,
        num_videos_per_prompt: Optional[int] = 1,
        generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
        latents: Optional[torch.FloatTensor] = None,
        output_type: Optional[str] = "pil",
        callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
        callback_on_step_end_tensor_inputs: List[str] = ["latents"],
        return_dict: bool = True,
    ):
        # 0. defaults (:196-200)
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        num_frames = num_frames if num_frames is not None else self.unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else num_frames
        # 1. (:203)
        self.check_inputs(image, cond_images, height, width)
        # 2. (:206-217)
        if PIL is not None and isinstance(image, PIL.Image.Image):
            batch_size = 1
        elif isinstance(image, list):
            batch_size = len(image)
        else:
            batch_size = image.shape[0]
        device = self._execution_device
        self._guidance_scale = max_guidance_scale
        # 3. CLIP (:220)
        image_embeddings = self._encode_image(image, device, num_videos_per_prompt, self.do_classifier_free_guidance)
        fps = fps - 1                                  # SVD was conditioned on fps - 1 (:224)
        # 4. VAE (:227-245)
        image = self.image_processor.preprocess(image, height=height, width=width).to(device)
        noise = randn_tensor(image.shape, generator=generator, device=device, dtype=image.dtype)
        image = image + noise_aug_strength * noise
        needs_upcasting = self.vae.dtype == torch.float16 and getattr(self.vae.config, "force_upcast", False)
        if needs_upcasting:
            self.vae.to(dtype=torch.float32)
        image_latents = self._encode_vae_image(image.to(self.vae.dtype), device=device,
                                               num_videos_per_prompt=num_videos_per_prompt,
                                               do_classifier_free_guidance=self.do_classifier_free_guidance)
        image_latents = image_latents.to(image_embeddings.dtype)
        image_latents = image_latents.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
        # 5. (:247-256)
        added_time_ids = self._get_add_time_ids(fps, motion_bucket_id, noise_aug_strength, image_embeddings.dtype,
                                                batch_size, num_videos_per_prompt, self.do_classifier_free_guidance)
        added_time_ids = added_time_ids.to(device)
        # 6. (:259-260)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        # 7a. (:263-274)
        num_channels_latents = self.unet.config.out_channels * 2
        latents = self.prepare_latents(batch_size * num_videos_per_prompt, num_frames, num_channels_latents, height,
                                       width, image_embeddings.dtype, device, generator, latents)
        # 7b. (:277-284)
        if cond_images is not None:
            cond_em = self._encode_vae_condition(cond_images, device, num_videos_per_prompt,
                                                 self.do_classifier_free_guidance)
            cond_em = cond_em.to(image_embeddings.dtype)
        else:
            cond_em = None
        # 8. (:287-292)
        guidance_scale = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames).unsqueeze(0)
        guidance_scale = guidance_scale.to(device, latents.dtype)
        guidance_scale = guidance_scale.repeat(batch_size * num_videos_per_prompt, 1)
        self._guidance_scale = guidance_scale[:, :, None, None, None]
        # 9. denoising loop (:295-343)
        self._num_timesteps = len(timesteps)
        with self.progress_bar(total=num_inference_steps) as progress_bar:
            latents = self._denoise(latents, image_latents, image_embeddings, added_time_ids, cond_em,
                                    num_inference_steps, min_guidance_scale, max_guidance_scale,
                                    control_condition_scale, callback_on_step_end,
                                    callback_on_step_end_tensor_inputs, progress_bar)
        if not output_type == "latent":                # (:345-349)
            frames = self.decode_latents(latents.to(self.vae.dtype), num_frames, decode_chunk_size)
            frames = tensor2vid(frames, self.image_processor, output_type=output_type)
        else:
            frames = latents
        if needs_upcasting:
            self.vae.to(dtype=torch.float16)
        self.maybe_free_model_hooks()
        if not return_dict:
            return frames
        return StableVideoDiffusionPipelineOutput(frames=frames)
