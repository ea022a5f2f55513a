"""Drop-in mirror of the reference's ``shencoder`` package (degree 1-8 like the reference; forward on the render path with degree 4, dy_dx / backward for training).

/root/reference/shencoder/sphere_harmonics.py:14-37 (``_sh_encoder.forward``), :60-87 (``SHEncoder``).
"""
import torch
import torch.nn as nn

from .._lib import check, lib, ptr, require_gpu, stream_ptr


class _sh_encoder(torch.autograd.Function):
    """shencoder/sphere_harmonics.py:14-56: forward [B,3] -> [B, degree^2]; with ``calc_grad_inputs`` also dy_dx [B, 3*degree^2] and a
    backward to the directions (training side)."""

    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        inputs = inputs.to(torch.float32).contiguous()  # custom_fwd(cast_inputs=torch.float32)
        require_gpu(inputs)
        B, input_dim = inputs.shape
        outputs = torch.empty(B, degree ** 2, dtype=inputs.dtype, device=inputs.device)
        dy_dx = torch.empty(B, input_dim * degree ** 2, dtype=inputs.dtype, device=inputs.device) if calc_grad_inputs else None
        check(lib().pn_sh_encode_forward(ptr(inputs), ptr(outputs), B, input_dim, int(degree), ptr(dy_dx), stream_ptr()), "sh_encode_forward")
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = [B, input_dim, int(degree)]
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        B, input_dim, degree = ctx.dims
        grad = grad.to(torch.float32).contiguous()
        grad_inputs = torch.zeros_like(inputs)
        check(lib().pn_sh_encode_backward(ptr(grad), ptr(inputs), B, input_dim, degree, ptr(dy_dx), ptr(grad_inputs), stream_ptr()), "sh_encode_backward")
        return grad_inputs, None, None


def sh_encode(inputs, degree, calc_grad_inputs=False):
    """sphere_harmonics.py:58: inputs [B,3] in [-1,1] -> [B, degree^2] fp32."""
    return _sh_encoder.apply(inputs, degree, calc_grad_inputs)


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert self.degree > 0 and self.degree <= 8, "SHEncoder: degree must be in [1, 8]"   # sphere_harmonics.py:70

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix_shape + [self.output_dim])
