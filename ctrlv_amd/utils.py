"""Small host utilities: synthetic weight initialisation of the HIP models directly on the device."""
import torch
from torch import nn


@torch.no_grad()
def random_init_(model, seed=0, zero_conv_std=None):
    """SURVEY.md 8(d) recipe, drawn on the model's device: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for every
    Conv/Linear weight and bias (PyTorch's default), norm affines (1, 0), mix_factor 0.5, ControlNet zero-convs
    either zero or N(0, zero_conv_std^2) so the residual path is exercised.  (The CPU oracle's `seeded_init_`
    is the bit-reproducible variant used for parity tests; this one only has to have the same statistics.)"""
    dev = next(model.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, mod in model.named_modules():
        if isinstance(mod, (nn.Conv2d, nn.Conv3d, nn.Linear)):
            fan_in = mod.weight[0].numel()
            bound = 1.0 / fan_in ** 0.5
            is_zero = name.startswith(("controlnet_down_blocks", "controlnet_mid_block"))
            for p in (mod.weight, mod.bias):
                if p is None:
                    continue
                if is_zero and zero_conv_std is None:
                    p.zero_()
                elif is_zero:
                    p.copy_((torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) * zero_conv_std).to(p.dtype))
                else:
                    p.copy_(((torch.rand(p.shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1) * bound).to(p.dtype))
        elif isinstance(mod, (nn.GroupNorm, nn.LayerNorm)):
            mod.weight.fill_(1.0)
            mod.bias.zero_()
    for name, p in model.named_parameters():
        if name.endswith("mix_factor"):
            p.fill_(0.5)
    if hasattr(model, "_packed"):
        model._packed = False
    return model


def build_on_device(cls, device, dtype=torch.bfloat16, **config):
    """Construct a model without materialising fp32 CPU weights (meta device -> empty device tensors)."""
    with torch.device("meta"):
        model = cls(**config)
    model = model.to_empty(device=device)
    model.to(dtype)
    return model.eval()
