"""ctrlv_amd's host-side helpers against vectors computed by the REFERENCE'S OWN CODE (tests/golden/make_ref_helpers.py ran
the reference's pure-torch definitions in the build container): the CLIP pre-processing resize
(bbox_generator_baseline/utils/image_encoder.py:184-291, called by the pipelines at pipeline_video_control.py:214-221) and
the EDM pre-conditioning / loss statements of the training step (tools/train_video_controlnet.py:410, 468-478).
These are the only reference-executed pins the path has (the model arithmetic lives in diffusers, absent here)."""
import os

import numpy as np
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_helpers.npz"))


def test_resize_with_antialiasing_matches_the_reference():
    from ctrlv_amd.pipelines.pipeline_utils import _resize_with_antialiasing
    for i in range(4):
        x, want = torch.from_numpy(G[f"resize{i}_in"]), torch.from_numpy(G[f"resize{i}_out"])
        got = _resize_with_antialiasing(x, tuple(int(v) for v in G[f"resize{i}_size"]))
        assert got.shape == want.shape
        assert float((got - want).abs().max()) < 2e-6, (i, float((got - want).abs().max()))


def test_edm_preconditioning_and_loss_match_the_reference():
    from ctrlv_amd.training import edm_loss, rows_of
    noisy, target, pred = (torch.from_numpy(G["edm_" + k]) for k in ("noisy_latents", "target_latents", "model_pred"))
    sig = torch.from_numpy(G["edm_sigmas"])
    # input scaling (train_video_controlnet.py:410): noisy / sqrt(sigma^2 + 1)
    s5 = sig.reshape(-1, 1, 1, 1, 1)
    assert torch.allclose(noisy / (s5 * s5 + 1) ** 0.5, torch.from_numpy(G["edm_inp_noisy_latents"]), rtol=1e-6, atol=1e-7)
    # loss (468-478) on the row layout the HIP path produces: rows (b, f, y, x) x 4 channels
    loss = edm_loss(rows_of(pred), noisy, target, sig)
    assert abs(float(loss) - float(G["edm_loss"])) <= 1e-5 * abs(float(G["edm_loss"])), (float(loss), float(G["edm_loss"]))
