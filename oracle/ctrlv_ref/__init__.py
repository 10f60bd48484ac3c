"""ORACLE -- test infrastructure only.  PARITY UNPINNED.

CPU (plain PyTorch) restatement of the reference's denoising hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package; the product
(`ctrlv_amd`) never does.  See blocks.py for the provenance statement.
"""
from .blocks import *            # noqa: F401,F403
from .blocks import storage_rounding, store, store_absmax  # noqa: F401
from .models import (SVD_CONFIG, TINY_CONFIG, ControlNetModel, UNetSpatioTemporalConditionModel,  # noqa: F401
                     seeded_init_, zero_module)
from .scheduler import (SVD_SCHEDULER_CONFIG, EulerDiscreteScheduler, guidance_scale_tensor,  # noqa: F401
                        sample_loop)

from . import vae  # noqa: F401,E402  (SVD VAE restatement: checker of the HIP VAE paths)
