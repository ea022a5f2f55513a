"""Drop-in mirror of the reference's ``shencoder`` package (forward only, degree <= 4).

/root/reference/shencoder/sphere_harmonics.py:14-37 (``_sh_encoder.forward``), :60-87 (``SHEncoder``).
"""
import torch
import torch.nn as nn

from .._lib import check, lib, ptr, require_gpu, stream_ptr


def sh_encode(inputs, degree, calc_grad_inputs=False):
    """sphere_harmonics.py:14-37: inputs [B,3] in [-1,1] -> [B, degree^2] fp32."""
    if calc_grad_inputs:
        raise RuntimeError("sh_encode: dy_dx / backward are not part of the inference path")
    inputs = inputs.to(torch.float32).contiguous()  # custom_fwd(cast_inputs=torch.float32)
    require_gpu(inputs)
    B, input_dim = inputs.shape
    outputs = torch.empty(B, degree ** 2, dtype=inputs.dtype, device=inputs.device)
    check(lib().pn_sh_encode_forward(ptr(inputs), ptr(outputs), B, input_dim, int(degree), None, stream_ptr()), "sh_encode_forward")
    return outputs


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert self.degree > 0 and self.degree <= 4, "this build implements degree in [1, 4] (the reference goes to 8; the renderer uses 4)"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix_shape + [self.output_dim])
