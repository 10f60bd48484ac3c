"""ctrlv_amd -- MI355X (gfx950) native implementation of Ctrl-V's denoising hot path.

Drop-in counterparts of the reference's callables:

  ctrlv_amd.models.UNetSpatioTemporalConditionModel   (src/ctrlv/models/unet_spatio_temporal_condition.py:13)
  ctrlv_amd.models.ControlNetModel                    (src/ctrlv/models/controlnet.py:20)
  ctrlv_amd.pipelines.StableVideoControlPipeline      (src/ctrlv/pipelines/pipeline_video_control.py:25)
  ctrlv_amd.pipelines.VideoDiffusionPipeline          (src/ctrlv/pipelines/pipeline_video_diffusion.py:18)

All arithmetic runs in hand-written HIP kernels from `libctrlv_hip.so` (C ABI: include/ctrlv_hip.h) reached through
ctypes; PyTorch only owns device memory, streams and `torch.distributed`.  There is no CPU / eager fallback: if the
shared library is missing every forward raises.
"""
__version__ = "0.1.0"
