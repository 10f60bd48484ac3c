"""VAE temporal decoder / encoder on the HIP kernels (ctrlv_amd/models/vae_decoder_hip.py, vae_encoder_hip.py; SURVEY 8 row
f4) against the CPU ORACLE of the VAE (oracle/ctrlv_ref/vae.py, fp32, the same parameters rounded to bf16), production
widths (128/256/512/512), e.g. 3 frames of a 16x24 latent -> 128x192 pixels.  Tolerance: 18 res blocks + attention of
bf16-stored activations, parity_err <= 2.5e-2 (measured value printed).  The oracle is an unpinned restatement of
diffusers' AutoencoderKLTemporalDecoder (tests/test_oracle.py checks it against the product's nn.Module form)."""
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
    import ctrlv_ref as R
    with torch.no_grad():
        ref = R.vae.decode({k: v.detach() for k, v in vae.state_dict().items()}, z, n)        # the oracle, fp32, CPU
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
    big = torch.empty(30, 4, 72, 128, device=DEV, dtype=torch.bfloat16)
    assert not vh.supports(big, 30) and vh.supports(big[:25], 25) and vh.supports(big[:25], 25, dev_vae.decoder)
    with pytest.raises(ValueError):
        vh.decode(dev_vae.decoder, big, 30)
    # frame-range path of the per-frame ops (forced by a tiny limit): identical bits to the whole-clip launches
    ref = vh.decode(dev_vae.decoder, z[:3], 3)
    old = vh._LIMIT
    try:
        vh._LIMIT = 3 * 64 * 64 * 128 * 2 + 1          # the clip's 128-channel tensors fit, its 256-channel ones do not
        assert len(vh._frame_batches(3, 64 * 64, 256)) == 3
        got = vh.decode(dev_vae.decoder, z[:3], 3)
    finally:
        vh._LIMIT = old
    assert torch.equal(got, ref)


@pytest.mark.parametrize("n,H,W", [(2, 128, 192), (1, 64, 64)])
def test_vae_encode_hip_matches_torch_module(vae, n, H, W):
    """Encoder on the HIP kernels (asymmetric-padding down-samplers as stride-1 conv + odd-position subsampling) against the
    torch module in fp32; also through `AutoencoderKLTemporalDecoder.encode(...).latent_dist.mode()`."""
    import copy
    from ctrlv_amd.models import vae_encoder_hip as ve
    x = (torch.rand(n, 3, H, W, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(torch.bfloat16).float()
    import ctrlv_ref as R
    sd = {k: v.detach() for k, v in vae.state_dict().items()}
    with torch.no_grad():
        ref_m = R.vae.encode_moments(sd, x, quant=False)          # the oracle, fp32, CPU
        ref_lat = R.vae.encode_moments(sd, x)[:, :4]              # latent_dist.mode() = the mean half
    dev_vae = copy.deepcopy(vae).to(DEV, torch.bfloat16)
    xd = x.to(DEV, torch.bfloat16)
    with torch.no_grad():
        got_m = ve.encode(dev_vae.encoder, xd)
        got_lat = dev_vae.encode(xd).latent_dist.mode()
    torch.cuda.synchronize()
    assert got_m.shape == (n, 8, H // 8, W // 8) and got_lat.shape == (n, 4, H // 8, W // 8)
    assert parity_err(got_m.float().cpu(), ref_m, "vae encoder moments") < 2.5e-2
    assert parity_err(got_lat.float().cpu(), ref_lat, "vae latents (mode)") < 2.5e-2


@torch.no_grad()
def test_pipeline_with_real_vae_hip_vs_torch(vae, monkeypatch):
    """`StableVideoControlPipeline.__call__` with the real VAE module (tiny UNet / ControlNet, 3 frames, 128x128): the HIP
    encoder / decoder dispatch inside `vae.encode` / `vae.decode` against the same call with CTRLV_VAE_HIP=0 (torch
    modules on the GPU).  Same seeded latents => the decoded frames agree to the VAE tolerance."""
    import copy
    import ctrlv_ref as R
    from ctrlv_amd.pipelines import StableVideoControlPipeline
    from ctrlv_amd.schedulers import EulerDiscreteScheduler
    from tests.fakes import FakeCLIP, fake_feature_extractor
    from tests.parity_utils import make_pair
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV)
    dev_vae = copy.deepcopy(vae).to(DEV, torch.bfloat16)
    clip = FakeCLIP(cfg["cross_attention_dim"]).to(DEV, torch.bfloat16)
    pipe = StableVideoControlPipeline(dev_vae, clip, hu, hc, EulerDiscreteScheduler(), fake_feature_extractor)
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(11)
    image = (torch.rand(1, 3, 128, 128, generator=g) * 2 - 1).to(DEV, torch.bfloat16)
    cond = (torch.rand(1, 3, 3, 128, 128, generator=g) * 2 - 1).to(DEV)
    lat = torch.randn(1, 3, 4, 16, 16, generator=g).to(DEV, torch.bfloat16)

    def run():
        return pipe(image, cond_images=cond, height=128, width=128, num_frames=3, num_inference_steps=2,
                    decode_chunk_size=2, latents=lat.clone(), noise_aug_strength=0.0, output_type="pt").frames.float().cpu()

    got = run()
    monkeypatch.setenv("CTRLV_VAE_HIP", "0")
    ref = run()
    assert got.shape == (1, 3, 3, 128, 128) and got.min() >= 0 and got.max() <= 1
    assert parity_err(got, ref, "pipeline frames, HIP VAE vs torch VAE") < 3e-2
