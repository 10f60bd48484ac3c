"""CPU tests of the host-side logic: weight packing, scheduler, model surface (config / state dict / save+load /
from_unet / validation), pipeline helpers, workspace.  No kernel is launched."""
import math
import os

import numpy as np
import pytest
import torch

import ctrlv_ref as R
from ctrlv_amd import packing
from ctrlv_amd._lib import CtrlvHipError
from ctrlv_amd.models import ControlNetModel, UNetSpatioTemporalConditionModel
from ctrlv_amd.schedulers import EulerDiscreteScheduler
from ctrlv_amd.workspace import Workspace

torch.set_grad_enabled(False)
TINY = dict(R.TINY_CONFIG)
CTINY = {k: v for k, v in TINY.items() if k not in ("out_channels", "up_block_types")}


# ------------------------------------------------------------------------------------------------ packing
def test_pack_conv3x3_and_temporal_index_order():
    w = torch.randn(40, 64, 3, 3)
    p = packing.pack_conv3x3(w)
    assert p.shape == (64, 9 * 64) and p.dtype == torch.bfloat16          # rows padded 40 -> 64
    for (n, ky, kx, c) in [(0, 0, 0, 0), (39, 2, 1, 63), (7, 1, 2, 5)]:
        assert p[n, (ky * 3 + kx) * 64 + c] == w[n, c, ky, kx].to(torch.bfloat16)
    assert p[40:].abs().max() == 0
    wt = torch.randn(64, 64, 3, 1, 1)
    pt = packing.pack_conv_temporal(wt)
    assert pt[5, 2 * 64 + 9] == wt[5, 9, 2, 0, 0].to(torch.bfloat16)


def test_pack_geglu_interleave_matches_chunk_semantics():
    inner, k = 128, 64
    w, b = torch.randn(2 * inner, k), torch.randn(2 * inner)
    wp, bp = packing.pack_geglu(w, b)
    x = torch.randn(5, k)
    proj = x @ wp.float().T + bp
    blocks = proj.reshape(5, inner // 16, 2, 16)
    a, gt = blocks[:, :, 0].reshape(5, inner), blocks[:, :, 1].reshape(5, inner)
    ref = x @ w.to(torch.bfloat16).float().T + b
    torch.testing.assert_close(a, ref[:, :inner], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gt, ref[:, inner:], rtol=1e-4, atol=1e-4)


def test_pack_linear_pads_and_conv_in_slots():
    p = packing.pack_linear(torch.randn(4, 96))
    assert p.shape == (32, 128) and p[:, 96:].abs().max() == 0 and p[4:].abs().max() == 0
    wa, wb = torch.randn(64, 8, 3, 3), torch.randn(64, 4, 3, 3)
    pc = packing.pack_conv_in([wa, wb], 16, 192)
    assert pc.shape == (64, 192)
    assert pc[3, 4 * 16 + 5] == wa[3, 5, 1, 1].to(torch.bfloat16)         # tap (1,1), conv_in channel 5
    assert pc[3, 4 * 16 + 8 + 2] == wb[3, 2, 1, 1].to(torch.bfloat16)     # control channel 2 sits in slot 8+2
    assert pc[:, 144:].abs().max() == 0 and pc[:, 12:16].abs().max() == 0
    assert packing.pad_bias(torch.ones(4)).shape == (32,)


# ------------------------------------------------------------------------------------------------ scheduler
@pytest.mark.parametrize("n", [25, 30, 50])
def test_scheduler_matches_oracle_and_golden(n):
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "scheduler_tables.npz"))
    s, o = EulerDiscreteScheduler(), R.EulerDiscreteScheduler()
    s.set_timesteps(n)
    o.set_timesteps(n)
    np.testing.assert_allclose(s.sigmas.numpy(), gold[f"sigmas_{n}"], rtol=1e-6)
    np.testing.assert_allclose(s.timesteps.numpy(), gold[f"timesteps_{n}"], rtol=1e-6, atol=1e-7)
    assert abs(s.init_noise_sigma - float(o.init_noise_sigma)) < 1e-3
    g = torch.Generator().manual_seed(0)
    x, v = torch.randn(1, 2, 4, 4, 4, generator=g) * 50, torch.randn(1, 2, 4, 4, 4, generator=g)
    for i in (0, 3, n - 1):
        s._step_index = o._step_index = None
        t = o.timesteps[i]
        torch.testing.assert_close(s.scale_model_input(x, s.timesteps[i]), o.scale_model_input(x, t), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(s.step(v, s.timesteps[i], x).prev_sample, o.step(v, t, x), rtol=1e-5, atol=1e-4)
    with pytest.raises(ValueError):
        EulerDiscreteScheduler(bogus=1)


# ------------------------------------------------------------------------------------------------ model surface
def test_models_share_diffusers_key_layout_and_param_counts():
    with torch.device("meta"):
        hu, hc = UNetSpatioTemporalConditionModel(), ControlNetModel()
        ou, oc = R.UNetSpatioTemporalConditionModel(), R.ControlNetModel()
    assert sum(p.numel() for p in hu.parameters()) == 1_524_623_082
    assert sum(p.numel() for p in hc.parameters()) == 680_946_897
    for h, o in ((hu, ou), (hc, oc)):
        hs, os_ = h.state_dict(), o.state_dict()
        assert set(hs) == set(os_)
        assert all(hs[k].shape == os_[k].shape for k in hs)
    assert hu.config.num_frames == 25 and hu.config.in_channels == 8 and hu.config.out_channels == 4
    assert hu.config.addition_time_embed_dim == 256 and hu.add_embedding.linear_1.in_features == 768
    assert hu.config["block_out_channels"] == (320, 640, 1280, 1280)


def test_config_validation_raises_like_the_reference():
    with pytest.raises(ValueError, match="block_out_channels"):
        ControlNetModel(block_out_channels=(320, 640))
    with pytest.raises(ValueError, match="num_attention_heads"):
        ControlNetModel(**dict(CTINY, num_attention_heads=(1, 2)))
    with pytest.raises(ValueError, match="layers_per_block"):
        ControlNetModel(**dict(CTINY, layers_per_block=(2, 2)))
    with pytest.raises(ValueError, match="head_dim 64"):
        UNetSpatioTemporalConditionModel(**dict(TINY, num_attention_heads=(2, 2, 2, 2)))
    with pytest.raises(ValueError):
        UNetSpatioTemporalConditionModel(**dict(TINY, down_block_types=("Bogus",) * 4))


def test_from_unet_copies_key_intersection_and_helpers():
    hu = UNetSpatioTemporalConditionModel(**TINY)
    hc = ControlNetModel.from_unet(hu)
    usd, csd = hu.state_dict(), hc.state_dict()
    inter = set(usd) & set(csd)
    assert len(inter) > 500 and all(torch.equal(usd[k], csd[k]) for k in inter)
    assert all(csd[k].abs().max() == 0 for k in csd if k.startswith(("controlnet_down_blocks", "controlnet_mid")))
    assert len(hc.controlnet_down_blocks) == 12
    params = hu.enable_grad(temporal_transformer_block=True)
    assert params and all(p.requires_grad for p in hu.get_parameters_with_grad())
    names = [n for n, p in hu.named_parameters() if p.requires_grad]
    assert all("temporal_transformer_block" in n for n in names)
    lat = torch.randn(2, 4, 8, 8)
    assert hu.encode_bbox_frame(lat, None).shape == (2, TINY["num_frames"], 4, 8, 8)


def test_save_and_load_pretrained_roundtrip(tmp_path):
    hu = UNetSpatioTemporalConditionModel(**TINY)
    for safe in (True, False):
        d = tmp_path / f"m{int(safe)}"
        hu.save_pretrained(str(d / "unet"), safe_serialization=safe)
        assert os.path.isfile(d / "unet" / "config.json")
        hu2 = UNetSpatioTemporalConditionModel.from_pretrained(str(d), subfolder="unet", num_frames=7)
        assert hu2.config.num_frames == 7                       # config override as tools/train_video_controlnet.py:106-109
        sd, sd2 = hu.state_dict(), hu2.state_dict()
        assert all(torch.equal(sd[k], sd2[k]) for k in sd)
    with pytest.raises(EnvironmentError):
        ControlNetModel.from_pretrained(str(tmp_path / "nope"))


def test_no_cpu_fallback():
    hu = UNetSpatioTemporalConditionModel(**TINY)
    x = torch.randn(1, 3, 8, 16, 16)
    with pytest.raises(CtrlvHipError, match="no CPU"):
        hu(x, torch.tensor(1.0), torch.randn(1, 1, 64), torch.tensor([[6.0, 127.0, 0.02]]))


def test_model_packing_tables_cover_every_block():
    hu = UNetSpatioTemporalConditionModel(**TINY)
    hu.pack()
    pk = hu._pk
    n_res = sum(1 for m in hu.modules() if type(m).__name__ == "SpatioTemporalResBlock")
    n_tr = sum(1 for m in hu.modules() if type(m).__name__ == "TransformerSpatioTemporalModel")
    assert (n_res, n_tr) == (22, 16)
    offs = sorted(o for m in hu.modules() if type(m).__name__ == "SpatioTemporalResBlock" for o in m.temb_off)
    assert len(offs) == 44 and offs[0] == 0 and len(set(offs)) == 44
    assert pk["temb_w"].shape[0] == pk["temb_n"] and len(pk["xattn_out"]) == 32
    # the fused to_v table row block of an attention equals its to_v weight
    tr = hu.down_blocks[0].attentions[0]
    off = tr.xattn_off[1]
    w = tr.temporal_transformer_blocks[0].attn2.to_v.weight
    assert torch.equal(pk["xv_w"][off:off + w.shape[0], :w.shape[1]], w.to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------ pipelines
class _FakeVAEConfig(dict):
    __getattr__ = dict.__getitem__


def _pipeline():
    from ctrlv_amd.pipelines import StableVideoControlPipeline

    class VAE(torch.nn.Module):
        config = _FakeVAEConfig(block_out_channels=(1, 1, 1, 1), scaling_factor=0.18215, force_upcast=False)
        dtype = torch.float32
    hu = UNetSpatioTemporalConditionModel(**TINY)
    hc = ControlNetModel.from_unet(hu)
    return StableVideoControlPipeline(VAE(), None, hu, hc, EulerDiscreteScheduler(), None)


def test_pipeline_input_checks_and_helpers():
    pipe = _pipeline()
    assert pipe.vae_scale_factor == 8
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe.check_inputs(torch.zeros(1, 3, 60, 64), torch.zeros(1, 3, 3, 60, 64), 60, 64)
    with pytest.raises(ValueError, match="cond_images"):
        pipe.check_inputs(torch.zeros(1, 3, 64, 64), None, 64, 64)
    with pytest.raises(ValueError, match="image"):
        pipe.check_inputs(3.0, torch.zeros(1), 64, 64)
    ids = pipe._get_add_time_ids(6, 127, 0.02, torch.float32, 1, 1, True)
    assert ids.shape == (2, 3) and ids[0].tolist() == pytest.approx([6, 127, 0.02])
    pipe.scheduler.set_timesteps(25)
    lat = pipe.prepare_latents(1, 3, 8, 64, 64, torch.float32, "cpu", torch.Generator().manual_seed(0))
    assert lat.shape == (1, 3, 4, 8, 8)
    assert abs(lat.std().item() / pipe.scheduler.init_noise_sigma - 1) < 0.2
    cond = torch.randn(1, 3, 4, 8, 8)
    em = pipe._encode_vae_condition(cond, "cpu", 1, True)          # 4-channel latents pass through, CFG half is zero
    assert em.shape == (2, 3, 4, 8, 8) and em[0].abs().max() == 0 and torch.equal(em[1], cond[0])
    pipe._guidance_scale = 3.0
    assert pipe.do_classifier_free_guidance
    pipe._guidance_scale = torch.ones(1, 3, 1, 1, 1)
    assert not pipe.do_classifier_free_guidance


def test_pipeline_from_pretrained_resolution_rules(tmp_path, monkeypatch):
    """from_pretrained as the reference's tools call it (tools/eval_video_controlnet.py:116-118,
    tools/train_video_controlnet.py:344-353): overrides win, unet/controlnet/scheduler load from the directory, the
    PyTorch-side components go through (overridable) loaders, a hub id resolves to a local copy or fails loudly."""
    from ctrlv_amd.pipelines import StableVideoControlPipeline, VideoDiffusionPipeline
    from ctrlv_amd.schedulers import EulerDiscreteScheduler
    from tests.fakes import FakeCLIP, FakeVAE, fake_feature_extractor
    pipe = _pipeline()
    pipe.vae, pipe.image_encoder, pipe.feature_extractor = FakeVAE(), FakeCLIP(64), fake_feature_extractor
    ck = tmp_path / "stabilityai" / "stable-video-diffusion-img2vid-xt"
    pipe.save_pretrained(str(ck))
    assert (ck / "model_index.json").is_file() and (ck / "scheduler" / "scheduler_config.json").is_file()
    sched = EulerDiscreteScheduler.from_pretrained(str(ck), subfolder="scheduler")
    assert sched.config["sigma_max"] == 700.0 and sched.config["prediction_type"] == "v_prediction"
    assert EulerDiscreteScheduler.from_config({"_class_name": "EulerDiscreteScheduler", "sigma_max": 500.0,
                                               "trained_betas": None}).config["sigma_max"] == 500.0
    with pytest.raises(NotImplementedError):
        EulerDiscreteScheduler.from_config({"prediction_type": "epsilon"})
    for n in ("vae", "image_encoder", "feature_extractor"):
        (ck / n).mkdir(exist_ok=True)
    seen = []
    loaders = {"vae": lambda p, **k: seen.append(("vae", p, k)) or FakeVAE(),
               "image_encoder": lambda p, **k: seen.append(("image_encoder", p, k)) or FakeCLIP(64),
               "feature_extractor": lambda p, **k: fake_feature_extractor}
    # (1) everything from the directory
    p1 = StableVideoControlPipeline.from_pretrained(str(ck), component_loaders=loaders, variant=None, revision=None)
    assert [s_[0] for s_ in seen] == ["vae", "image_encoder"] and seen[0][1] == str(ck / "vae")
    assert sorted(p1.unet.state_dict()) == sorted(pipe.unet.state_dict())
    assert all(torch.equal(a, b) for a, b in zip(p1.controlnet.state_dict().values(),
                                                 pipe.controlnet.state_dict().values()))
    # (2) the eval tools: hub id + model overrides; the id resolves through $CTRLV_MODEL_ROOT
    monkeypatch.setenv("CTRLV_MODEL_ROOT", str(tmp_path))
    p2 = StableVideoControlPipeline.from_pretrained("stabilityai/stable-video-diffusion-img2vid-xt",
                                                    controlnet=pipe.controlnet, unet=pipe.unet,
                                                    component_loaders=loaders)
    assert p2.unet is pipe.unet and p2.controlnet is pipe.controlnet and isinstance(p2.vae, FakeVAE)
    p2.to("cpu"); p2.set_progress_bar_config(disable=True)
    # (3) the training tool: every module passed in -> no directory needed at all (scheduler = SVD defaults)
    monkeypatch.delenv("CTRLV_MODEL_ROOT")
    p3 = StableVideoControlPipeline.from_pretrained("no/such-model", unet=pipe.unet, controlnet=pipe.controlnet,
                                                    vae=pipe.vae, image_encoder=pipe.image_encoder,
                                                    feature_extractor=fake_feature_extractor, revision=None,
                                                    variant=None, torch_dtype=torch.float32)
    assert isinstance(p3.scheduler, EulerDiscreteScheduler) and p3.scheduler.config["sigma_max"] == 700.0
    # (4) UNet-only pipeline takes no controlnet; a missing component without a directory fails loudly
    p4 = VideoDiffusionPipeline.from_pretrained(str(ck), unet=pipe.unet, component_loaders=loaders)
    assert not hasattr(p4, "controlnet") and p4.unet is pipe.unet
    with pytest.raises(EnvironmentError, match="never downloads"):
        StableVideoControlPipeline.from_pretrained("no/such-model", unet=pipe.unet, controlnet=pipe.controlnet)
    assert StableVideoControlPipeline.use_hip_graph      # graph replay is the default loop mode


def test_vae_temporal_decoder_module(tmp_path):
    """ctrlv_amd's PyTorch-side AutoencoderKLTemporalDecoder (outside the hot path): diffusers state-dict key
    layout, the known parameter count of the (Stable-Diffusion-identical) encoder, shapes of encode / decode in the way
    the pipelines call them (pipeline_video_control.py:71-101,235,346), save / load round trip, frame mixing."""
    from ctrlv_amd.models import AutoencoderKLTemporalDecoder
    with torch.device("meta"):
        full = AutoencoderKLTemporalDecoder()
    n_enc = sum(p.numel() for p in full.encoder.parameters())
    assert n_enc == 34_163_592                                   # SD VAE encoder (128/256/512/512, 2 layers per block)
    assert sum(p.numel() for p in full.quant_conv.parameters()) == 72
    keys = set(full.state_dict())
    for k in ("encoder.conv_in.weight", "encoder.down_blocks.0.resnets.1.norm2.bias",
              "encoder.down_blocks.1.resnets.0.conv_shortcut.weight", "encoder.down_blocks.2.downsamplers.0.conv.weight",
              "encoder.mid_block.attentions.0.to_q.bias", "encoder.mid_block.attentions.0.group_norm.weight",
              "encoder.mid_block.resnets.1.conv2.weight", "encoder.conv_norm_out.weight", "quant_conv.weight",
              "decoder.conv_in.weight", "decoder.mid_block.resnets.0.spatial_res_block.conv1.weight",
              "decoder.mid_block.resnets.1.temporal_res_block.conv2.bias", "decoder.mid_block.resnets.0.time_mixer.mix_factor",
              "decoder.mid_block.attentions.0.to_out.0.weight", "decoder.up_blocks.0.resnets.2.spatial_res_block.norm1.weight",
              "decoder.up_blocks.2.resnets.0.spatial_res_block.conv_shortcut.weight",
              "decoder.up_blocks.2.upsamplers.0.conv.weight", "decoder.conv_norm_out.bias", "decoder.conv_out.weight",
              "decoder.time_conv_out.weight"):
        assert k in keys, k
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in keys
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in keys
    assert full.state_dict()["decoder.time_conv_out.weight"].shape == (3, 3, 3, 1, 1)
    assert full.config.scaling_factor == 0.18215 and len(full.config.block_out_channels) == 4
    torch.manual_seed(0)
    vae = AutoencoderKLTemporalDecoder(block_out_channels=(32, 64, 64, 64), layers_per_block=1).eval()
    with torch.no_grad():
        x = torch.rand(3, 3, 64, 64) * 2 - 1
        dist = vae.encode(x).latent_dist
        z = dist.mode()
        assert z.shape == (3, 4, 8, 8) and torch.equal(z, vae.encode(x).latent_dist.mode())
        g = torch.Generator().manual_seed(1)
        assert dist.sample(g).shape == z.shape and not torch.equal(dist.sample(g), z)
        y = vae.decode(z, num_frames=3).sample
        assert y.shape == (3, 3, 64, 64) and torch.isfinite(y).all()
        # the temporal path mixes frames: decoding frame 0 together with other frames differs from decoding it alone,
        # once the mixers are opened (merge_factor 0.0 -> alpha = 0.5 at init already)
        y1 = vae.decode(z[:1], num_frames=1).sample
        assert (y[:1] - y1).abs().max() > 1e-6
        with pytest.raises(ValueError, match="multiple of num_frames"):
            vae.decode(z, num_frames=2)
        vae.save_pretrained(str(tmp_path / "vae"))
        v2 = AutoencoderKLTemporalDecoder.from_pretrained(str(tmp_path / "vae"))
        assert torch.equal(v2.decode(z, num_frames=3).sample, y)
    from ctrlv_amd.pipelines.pipeline_utils import _load_vae
    assert type(_load_vae(str(tmp_path / "vae"))).__name__ == "AutoencoderKLTemporalDecoder"     # no diffusers here


def test_image_processor_and_tensor2vid():
    from ctrlv_amd.pipelines.pipeline_utils import VaeImageProcessor, _resize_with_antialiasing, tensor2vid
    ip = VaeImageProcessor(8)
    x01 = torch.rand(1, 3, 16, 16)
    torch.testing.assert_close(ip.preprocess(x01, 16, 16), 2 * x01 - 1)
    xm = x01 * 2 - 1
    torch.testing.assert_close(ip.preprocess(xm, 16, 16), xm)
    v = torch.rand(2, 3, 5, 8, 8) * 2 - 1
    out = tensor2vid(v, ip, "pt")
    assert out.shape == (2, 5, 3, 8, 8) and out.min() >= 0 and out.max() <= 1
    assert tensor2vid(v, ip, "np").shape == (2, 5, 8, 8, 3)
    r = _resize_with_antialiasing(torch.rand(1, 3, 64, 96), (32, 32))
    assert r.shape == (1, 3, 32, 32) and torch.isfinite(r).all()


# ------------------------------------------------------------------------------------------------ workspace
def test_workspace_stack_discipline():
    ws = Workspace("cpu", chunk_bytes=1 << 16)
    a = ws.alloc((10, 8))
    m = ws.mark()
    b = ws.alloc((100, 64))
    c = ws.alloc((1000, 64))           # larger than a chunk -> own block
    base = ws.blocks[0].data_ptr()          # device allocations are >= 256-B aligned; offsets are what we control
    assert (a.data_ptr() - base) % 256 == 0 and (b.data_ptr() - base) % 256 == 0 and c.numel() == 64000
    ws.release(m)
    b2 = ws.alloc((100, 64))
    assert b2.data_ptr() == b.data_ptr()
    cap = ws.capacity()
    ws.reset()
    for _ in range(3):                 # steady state: no growth
        ws.alloc((10, 8)); mm = ws.mark(); ws.alloc((100, 64)); ws.alloc((1000, 64)); ws.release(mm); ws.reset()
    assert ws.capacity() == cap


def test_workspace_trunk_tensors_carry_a_one_byte_lo_plane_in_split_mode():
    """Workspace.trunk (the per-op executor's residual-trunk tensors, csrc/plan.hip `Trk`): plain mode -> one plane, `.lo` is
    None; split mode ("fp16x2") -> the fp16 hi plane carries a uint8 (e5m2) lo plane of the same shape as the attribute `.lo`,
    3 bytes per element, both from the arena (released together); ordinary allocations never carry one."""
    ws = Workspace("cpu", chunk_bytes=1 << 20)
    ws.el = torch.float16
    t = ws.trunk((64, 320))
    assert t.dtype == torch.float16 and t.lo is None
    ws.reset()
    ws.split = True
    m = ws.mark()
    t = ws.trunk((64, 320))
    assert t.dtype == torch.float16 and t.lo is not None and t.lo.dtype == torch.uint8 and tuple(t.lo.shape) == (64, 320)
    assert t.lo.data_ptr() - t.data_ptr() == 64 * 320 * 2                     # the lo plane follows the hi plane
    assert ws.off - m[1] == 64 * 320 * 3
    alias = t                                                                  # (h2 = h0: the alias keeps the plane)
    assert alias.lo is t.lo
    plain = ws.alloc((64, 320))
    assert getattr(plain, "lo", None) is None
    ws.release(m)
    assert ws.trunk((64, 320)).data_ptr() == t.data_ptr()
