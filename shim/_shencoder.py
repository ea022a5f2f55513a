"""Drop-in for the reference's ``_shencoder`` extension module (shencoder/src/bindings.cpp, shencoder.h:9-10)."""
from pienerf_amd._lib import check, lib, ptr, require_gpu, stream_ptr


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
    require_gpu(inputs, outputs)
    check(lib().pn_sh_encode_forward(ptr(inputs), ptr(outputs), int(B), int(D), int(C), ptr(dy_dx), stream_ptr()), "sh_encode_forward")


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    require_gpu(grad, inputs, dy_dx, grad_inputs)
    check(lib().pn_sh_encode_backward(ptr(grad), ptr(inputs), int(B), int(D), int(C), ptr(dy_dx), ptr(grad_inputs), stream_ptr()), "sh_encode_backward")
