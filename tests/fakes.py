"""Stand-in VAE / CLIP / feature extractor (the modules the reference's callers supply from diffusers / transformers;
they sit outside the hot path) for the pipeline tests."""
import types

import torch
import torch.nn.functional as F


class _Dist:
    def __init__(self, m):
        self._m = m

    def mode(self):
        return self._m


class FakeVAE(torch.nn.Module):
    """Deterministic stand-in for AutoencoderKLTemporalDecoder: 8x average pool + fixed 3->4 channel mix."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.mix = torch.nn.Parameter(torch.randn(4, 3, generator=g) * 0.5, requires_grad=False)
        self.config = types.SimpleNamespace(block_out_channels=(1, 1, 1, 1), scaling_factor=0.18215, force_upcast=False)

    @property
    def dtype(self):
        return self.mix.dtype

    def encode(self, x):
        lat = torch.einsum("oc,nchw->nohw", self.mix.to(x.dtype), F.avg_pool2d(x, 8))
        return types.SimpleNamespace(latent_dist=_Dist(lat))

    def decode(self, z, num_frames=None):
        img = torch.einsum("oc,nohw->nchw", self.mix.to(z.dtype), z)
        return types.SimpleNamespace(sample=F.interpolate(img, scale_factor=8.0, mode="nearest"))


class FakeCLIP(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        g = torch.Generator().manual_seed(6)
        self.proj = torch.nn.Parameter(torch.randn(dim, 3, generator=g), requires_grad=False)

    def forward(self, pixel_values):
        return types.SimpleNamespace(image_embeds=pixel_values.mean(dim=(2, 3)) @ self.proj.T.to(pixel_values.dtype))


def fake_feature_extractor(images, **kw):
    return types.SimpleNamespace(pixel_values=images)
