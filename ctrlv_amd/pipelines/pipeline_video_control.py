"""StableVideoControlPipeline (Box2Video sampling) on MI355X.

Counterpart of /root/reference/src/ctrlv/pipelines/pipeline_video_control.py:25-360 with the same constructor
(:27-46), `check_inputs` (:51-68), `_encode_vae_condition` (:71-101) and `__call__` signature / defaults / output
types (:105-127, :345-360).  The denoising loop (:298-343) runs on the HIP models; see pipeline_utils._denoise.
"""
from typing import Callable, Dict, List, Optional, Union

import torch

from .pipeline_utils import PIL, SVDPipelineBase, VaeImageProcessor


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
        job = self.prepare_clip(
            image, lambda h, w: self.check_inputs(image, cond_images, h, w), height=height, width=width,
            num_frames=num_frames, decode_chunk_size=decode_chunk_size, fps=fps, motion_bucket_id=motion_bucket_id,
            noise_aug_strength=noise_aug_strength, num_videos_per_prompt=num_videos_per_prompt, generator=generator,
            latents=latents, latent_channels=self.unet.config.out_channels * 2, min_guidance_scale=min_guidance_scale,
            max_guidance_scale=max_guidance_scale, num_inference_steps=num_inference_steps)
        # bbox frames -> VAE latents: the ControlNet's conditioning (:277-284); encoded after the initial noise is drawn
        control = None
        if cond_images is not None:
            control = self._encode_vae_condition(cond_images, job.device, num_videos_per_prompt,
                                                 job.cfg).to(job.clip_embeds.dtype)
        with self.progress_bar(total=num_inference_steps) as bar:          # the loop (:295-343) on the HIP models
            out = self._denoise(job.latents, job.cond_latents, job.clip_embeds, job.time_ids, control,
                                num_inference_steps, min_guidance_scale, max_guidance_scale, control_condition_scale,
                                callback_on_step_end, callback_on_step_end_tensor_inputs, bar, do_cfg=job.cfg)
        return self.finish_clip(job, out, output_type, return_dict)
