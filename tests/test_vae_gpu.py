"""VAE temporal decoder on the HIP kernels (ctrlv_amd/models/vae_decoder_hip.py) against the plain-torch
module it executes (fp32 on the CPU: the same parameters rounded to bf16), production widths (128/256/512/512), 3 frames
of a 16x24 latent -> 128x192 pixels.  Tolerance: 18 res blocks + attention of bf16-stored activations, parity_err <= 2.5e-2
(measured value printed).  The torch module is itself an unpinned restatement of diffusers' AutoencoderKLTemporalDecoder."""
import pytest
import torch

from tests.parity_utils import parity_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vae(hip_lib):
    from ctrlv_amd.models import AutoencoderKLTemporalDecoder
    torch.manual_seed(5)
    m = AutoencoderKLTemporalDecoder().eval()
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("mix_factor"):
                p.fill_(0.4)                          # exercise both branches of the AlphaBlender
            p.copy_(p.to(torch.bfloat16).float())
    return m


@pytest.mark.parametrize("n,h,w", [(3, 16, 24), (1, 8, 8)])
def test_vae_decode_hip_matches_torch_module(vae, n, h, w):
    import copy
    from ctrlv_amd.models import vae_decoder_hip as vh
    z = torch.randn(n, 4, h, w, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).float()
    with torch.no_grad():
        ref = vae.decoder(z, n)                                   # torch modules, fp32, CPU
    dev_vae = copy.deepcopy(vae).to(DEV, torch.bfloat16)
    zd = z.to(DEV, torch.bfloat16)
    assert vh.supports(zd, n)
    with torch.no_grad():
        got = dev_vae.decode(zd, num_frames=n).sample             # dispatches to the HIP executor
        direct = vh.decode(dev_vae.decoder, zd, n)
    torch.cuda.synchronize()
    assert got.shape == (n, 3, 8 * h, 8 * w) and got.dtype == torch.bfloat16
    assert torch.equal(got, direct)
    assert parity_err(got.float().cpu(), ref, "vae decode") < 2.5e-2


def test_vae_decode_two_clips_and_limits(vae):
    """Two clips in one call decode independently (temporal blocks do not mix them); oversize chunks are refused."""
    import copy
    from ctrlv_amd.models import vae_decoder_hip as vh
    dev_vae = copy.deepcopy(vae).to(DEV, torch.bfloat16)
    z = torch.randn(4, 4, 8, 8, generator=torch.Generator().manual_seed(9)).to(DEV, torch.bfloat16)
    with torch.no_grad():
        both = vh.decode(dev_vae.decoder, z, 2)
        first, second = vh.decode(dev_vae.decoder, z[:2], 2), vh.decode(dev_vae.decoder, z[2:], 2)
    assert torch.equal(both, torch.cat([first, second]))
    big = torch.empty(25, 4, 72, 128, device=DEV, dtype=torch.bfloat16)
    assert not vh.supports(big, 25) and vh.supports(big[:14], 14)
    with pytest.raises(ValueError):
        vh.decode(dev_vae.decoder, big, 25)


@pytest.mark.parametrize("n,H,W", [(2, 128, 192), (1, 64, 64)])
def test_vae_encode_hip_matches_torch_module(vae, n, H, W):
    """Encoder on the HIP kernels (asymmetric-padding down-samplers as stride-1 conv + odd-position subsampling) against the
    torch module in fp32; also through `AutoencoderKLTemporalDecoder.encode(...).latent_dist.mode()`."""
    import copy
    from ctrlv_amd.models import vae_encoder_hip as ve
    x = (torch.rand(n, 3, H, W, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(torch.bfloat16).float()
    with torch.no_grad():
        ref_m = vae.encoder(x)
        ref_lat = vae.encode(x).latent_dist.mode()
    dev_vae = copy.deepcopy(vae).to(DEV, torch.bfloat16)
    xd = x.to(DEV, torch.bfloat16)
    with torch.no_grad():
        got_m = ve.encode(dev_vae.encoder, xd)
        got_lat = dev_vae.encode(xd).latent_dist.mode()
    torch.cuda.synchronize()
    assert got_m.shape == (n, 8, H // 8, W // 8) and got_lat.shape == (n, 4, H // 8, W // 8)
    assert parity_err(got_m.float().cpu(), ref_m, "vae encoder moments") < 2.5e-2
    assert parity_err(got_lat.float().cpu(), ref_lat, "vae latents (mode)") < 2.5e-2
