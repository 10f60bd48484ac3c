"""Model shells with the reference's class names (src/ctrlv/models/__init__.py exports the same two)."""
from .controlnet import ControlNetModel, ControlNetOutput  # noqa: F401
from .unet_spatio_temporal_condition import (UNetSpatioTemporalConditionModel,  # noqa: F401
                                             UNetSpatioTemporalConditionOutput)
from .autoencoder_kl_temporal_decoder import AutoencoderKLTemporalDecoder  # noqa: F401,E402  (PyTorch-ROCm module)
