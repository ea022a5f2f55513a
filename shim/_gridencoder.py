"""Drop-in for the reference's ``_gridencoder`` extension module (gridencoder/src/bindings.cpp, gridencoder.h:11-15): positional signatures kept,
outputs owned by the caller.  ``outputs`` is [L, B, C] like the reference kernel writes it (grid.py:47 permutes afterwards); half tables / outputs
select kernel_grid<at::Half> (AT_DISPATCH_FLOATING_TYPES_AND_HALF on embeddings.scalar_type(), gridencoder.cu:448-471), forward only."""
import torch

from pienerf_amd._lib import check, lib, ptr, require_gpu, stream_ptr

import weakref

_host_offsets = {}  # id(tensor) -> (weak reference, version, host copy)


def _offsets_host(offsets):
    """The reference hands `offsets` over as a device tensor; the launcher derives the level geometry on the host, so a host copy is kept per tensor
    OBJECT and version (a key made of the data pointer would be reused by the caching allocator for another table: stale geometry, out-of-range
    reads)."""
    key = id(offsets)
    hit = _host_offsets.get(key)
    if hit is not None and hit[0]() is offsets and hit[1] == offsets._version:
        return hit[2]
    host = offsets.detach().to("cpu", torch.int32).contiguous()
    _host_offsets[key] = (weakref.ref(offsets, lambda _, k=key: _host_offsets.pop(k, None)), offsets._version, host)
    return host


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
    require_gpu(inputs, embeddings, outputs)
    if not (inputs.is_contiguous() and embeddings.is_contiguous() and outputs.is_contiguous()):
        raise RuntimeError("inputs, embeddings and outputs must be contiguous tensors")  # CHECK_CONTIGUOUS (gridencoder.cu:449-465)
    off = _offsets_host(offsets)
    if embeddings.dtype == torch.float16:
        if dy_dx is not None:
            raise RuntimeError("grid_encode_forward: dy_dx with a half table (fp16 training) is not built")
        check(lib().pn_grid_encode_forward_half(ptr(inputs), ptr(embeddings), off.data_ptr(), ptr(outputs), int(B), int(D), int(C), int(L), float(S), int(H),
                                                int(gridtype), int(bool(align_corners)), int(interp), 0, stream_ptr()), "grid_encode_forward")
        return
    if embeddings.dtype != torch.float32 or outputs.dtype != torch.float32:
        raise RuntimeError("embeddings must be a float or half tensor")  # CHECK_IS_FLOATING; double is not built
    # the reference's dy_dx is [B, L*D*C]; the C ABI writes the same memory order [B, L, D, C]
    check(lib().pn_grid_encode_forward(ptr(inputs), ptr(embeddings), off.data_ptr(), ptr(outputs), int(B), int(D), int(C), int(L), float(S), int(H), ptr(dy_dx),
                                       int(gridtype), int(bool(align_corners)), int(interp), 0, stream_ptr()), "grid_encode_forward")


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp):
    require_gpu(grad, inputs, embeddings, grad_embeddings)
    check(lib().pn_grid_encode_backward(ptr(grad), ptr(inputs), ptr(embeddings), _offsets_host(offsets).data_ptr(), ptr(grad_embeddings), int(B), int(D), int(C), int(L),
                                        float(S), int(H), ptr(dy_dx), ptr(grad_inputs), int(gridtype), int(bool(align_corners)), int(interp), stream_ptr()),
          "grid_encode_backward")


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
    require_gpu(inputs, embeddings, grad)
    check(lib().pn_grad_total_variation(ptr(inputs), ptr(embeddings), ptr(grad), _offsets_host(offsets).data_ptr(), float(weight), int(B), int(D), int(C), int(L),
                                        float(S), int(H), int(gridtype), int(bool(align_corners)), stream_ptr()), "grad_total_variation")
