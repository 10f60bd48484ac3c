"""Sampling pipelines with the reference's class names (src/ctrlv/pipelines/__init__.py)."""
from .pipeline_video_control import StableVideoControlPipeline  # noqa: F401
from .pipeline_video_diffusion import VideoDiffusionPipeline  # noqa: F401
