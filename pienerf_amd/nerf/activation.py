"""nerf/activation.py:5-18 of the reference: exp with a clamped-gradient backward (training side)."""
import torch


class _trunc_exp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.to(torch.float32)  # custom_fwd(cast_inputs=torch.float32)
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _trunc_exp.apply
