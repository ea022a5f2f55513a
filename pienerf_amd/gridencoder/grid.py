"""Mirror of the reference's ``gridencoder`` package surface.

Interface replaced: /root/reference/gridencoder/grid.py:24-63 (``_grid_encode.forward``) and
:96-161 (``GridEncoder``): same call signatures, same module attributes / state-dict keys
(``offsets`` buffer, ``embeddings`` parameter).  Forward is on the simulate-and-render path; backward, dy_dx and
``grad_total_variation`` (:65-92,168-190) are the training side.  Under autocast the forward runs on a half copy of the
table (grid.py:43-44: ``embeddings.to(torch.half)``) and returns half features; its backward is the half scatter-add of gridencoder.cu:324-331.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .._lib import check, lib, ptr, require_gpu, stream_ptr

GRIDTYPES = ("hash", "tiled")        # ids 0, 1 (grid.py:14-17)
INTERPOLATIONS = ("linear", "smoothstep")  # ids 0, 1 (grid.py:19-22)


def level_table_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """Entry offsets of the per-level tables (grid.py:116-127): level i holds min(2^log2_hashmap_size, (res_i [+1])^D)
    entries rounded up to a multiple of 8, res_i = ceil(base_resolution * per_level_scale^i)."""
    cap = 1 << log2_hashmap_size
    sizes = []
    for lvl in range(num_levels):
        side = int(np.ceil(base_resolution * per_level_scale ** lvl)) + (0 if align_corners else 1)
        sizes.append(8 * math.ceil(min(cap, side ** input_dim) / 8))
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)


class _grid_encode(torch.autograd.Function):
    """gridencoder/grid.py:24-92.  Forward: one HIP launch writes [B, L*C] directly (the reference writes [L,B,C] and permutes,
    grid.py:47,57), plus the dy_dx launch when ``calc_grad_inputs``.  Backward (training side): scatter-add into
    grad_embeddings with hardware fp32 atomics and, with dy_dx, the chain rule to the inputs.  Autocast (grid.py:43-44): when
    ``torch.is_autocast_enabled()`` and C is even, the table is cast to half, the kernel is kernel_grid<at::Half> and the features come
    back in half; the backward then accumulates a half grad_embeddings with packed half atomics (kernel_grid_backward<at::Half>); input gradients: fp32 path only."""

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False,
                interpolation=0, offsets_host=None, grad_mode=True):
        # grad_mode: torch.is_grad_enabled() at the call site (inside forward() it is always off, and ctx.needs_input_grad follows requires_grad alone)
        x = inputs.to(torch.float32).contiguous()
        B, D = x.shape
        n_levels = offsets.shape[0] - 1
        C = embeddings.shape[1]
        if offsets_host is None:
            offsets_host = offsets.detach().to("cpu", torch.int32).contiguous()
        log2_scale = float(np.float32(np.log2(per_level_scale)))  # the reference passes S = log2(per_level_scale) as a float (grid.py:37)
        if (torch.is_autocast_enabled() and C % 2 == 0) or embeddings.dtype == torch.float16:  # grid.py:41-44
            if calc_grad_inputs:
                raise RuntimeError("grid_encode: input gradients are not built on the half-precision path; disable autocast for them")
            table = embeddings.detach().to(torch.float16).contiguous()
            require_gpu(x, table)
            feats = torch.empty(B, n_levels * C, device=x.device, dtype=torch.float16)
            check(lib().pn_grid_encode_forward_half(ptr(x), ptr(table), offsets_host.data_ptr(), ptr(feats), B, D, C, n_levels, log2_scale,
                                                    int(base_resolution), int(gridtype), int(bool(align_corners)), int(interpolation), 1, stream_ptr()),
                  "grid_encode_forward_half")
            if grad_mode and ctx.needs_input_grad[0]:
                raise RuntimeError("grid_encode: the half-precision path has no gradient to its inputs (inputs.requires_grad under autocast)")
            if grad_mode and ctx.needs_input_grad[1]:   # fp16 training (trainer.py:561 with --fp16): the half backward below
                ctx.save_for_backward(x)
                ctx.half = True
                ctx.table_shape, ctx.table_dtype = tuple(embeddings.shape), embeddings.dtype
                ctx.dims = [B, D, C, n_levels, log2_scale, int(base_resolution), int(gridtype), int(interpolation)]
                ctx.align_corners = bool(align_corners)
                ctx.offsets_host = offsets_host
            else:
                ctx.mark_non_differentiable(feats)
            return feats
        table = embeddings.to(torch.float32).contiguous()
        require_gpu(x, table)
        feats = torch.empty(B, n_levels * C, device=x.device, dtype=torch.float32)
        dy_dx = torch.empty(B, n_levels * D * C, device=x.device, dtype=torch.float32) if calc_grad_inputs else None
        rc = lib().pn_grid_encode_forward(ptr(x), ptr(table), offsets_host.data_ptr(), ptr(feats), B, D, C, n_levels, log2_scale, int(base_resolution),
                                          ptr(dy_dx), int(gridtype), int(bool(align_corners)), int(interpolation), 1, stream_ptr())
        check(rc, "grid_encode_forward")
        ctx.save_for_backward(x, table, dy_dx)
        ctx.half = False
        ctx.dims = [B, D, C, n_levels, log2_scale, int(base_resolution), int(gridtype), int(interpolation)]
        ctx.align_corners = bool(align_corners)
        ctx.offsets_host = offsets_host
        return feats

    @staticmethod
    def backward(ctx, grad):
        if ctx.half:   # grid.py:65-92 with a half table: grad_embeddings in half (autograd casts it to the parameter's dtype)
            (x,) = ctx.saved_tensors
            B, D, C, L, S, H, gridtype, interpolation = ctx.dims
            grad = grad.to(torch.float16).view(B, L, C).permute(1, 0, 2).contiguous()
            grad_embeddings = torch.zeros(ctx.table_shape, device=x.device, dtype=torch.float16)
            check(lib().pn_grid_encode_backward_half(ptr(grad), ptr(x), ctx.offsets_host.data_ptr(), ptr(grad_embeddings), B, D, C, L, S, H, gridtype,
                                                     int(ctx.align_corners), interpolation, stream_ptr()), "grid_encode_backward_half")
            return None, grad_embeddings.to(ctx.table_dtype), None, None, None, None, None, None, None, None, None
        x, table, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation = ctx.dims
        grad = grad.to(torch.float32).view(B, L, C).permute(1, 0, 2).contiguous()  # [B, L*C] -> [L, B, C] (grid.py:73)
        grad_embeddings = torch.zeros_like(table)
        grad_inputs = torch.zeros_like(x) if dy_dx is not None else None
        rc = lib().pn_grid_encode_backward(ptr(grad), ptr(x), ptr(table), ctx.offsets_host.data_ptr(), ptr(grad_embeddings), B, D, C, L, S, H, ptr(dy_dx),
                                           ptr(grad_inputs), gridtype, int(ctx.align_corners), interpolation, stream_ptr())
        check(rc, "grid_encode_backward")
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None, None, None


def grid_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False,
                interpolation=0, offsets_host=None):
    """inputs [B,D] in [0,1], embeddings [sO,C], offsets [L+1] -> features [B, L*C] fp32 (differentiable w.r.t. embeddings, and w.r.t.
    inputs when ``calc_grad_inputs``)."""
    return _grid_encode.apply(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs, gridtype, align_corners, interpolation,
                              offsets_host, torch.is_grad_enabled())


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=None,
                 gridtype="hash", align_corners=False, interpolation="linear"):
        super().__init__()
        if desired_resolution is not None:  # finest resolution wins over per_level_scale (grid.py:100-102)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution, self.log2_hashmap_size = per_level_scale, base_resolution, log2_hashmap_size
        self.output_dim = num_levels * level_dim
        self.gridtype, self.gridtype_id = gridtype, GRIDTYPES.index(gridtype)
        self.interpolation, self.interp_id = interpolation, INTERPOLATIONS.index(interpolation)
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size
        table = level_table_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self._offsets_host = torch.from_numpy(table)  # host copy: the launcher derives the level geometry on the host
        self.register_buffer("offsets", self._offsets_host.clone())
        self.n_params = int(table[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(table[-1]), level_dim).uniform_(-1e-4, 1e-4))

    def extra_repr(self):
        return (f"levels={self.num_levels}x{self.level_dim} res={self.base_resolution}..x{self.per_level_scale:.4f} "
                f"params={tuple(self.embeddings.shape)} {self.gridtype}/{self.interpolation}")

    def forward(self, inputs, bound=1):
        unit = (inputs + bound) / (2 * bound)  # [-bound, bound] -> [0, 1] (grid.py:149)
        lead = list(unit.shape[:-1])
        feats = grid_encode(unit.view(-1, self.input_dim), self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                            unit.requires_grad, self.gridtype_id, self.align_corners, self.interp_id, offsets_host=self._offsets_host)
        return feats.view(lead + [self.output_dim])

    @torch.no_grad()
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """gridencoder/grid.py:168-190: adds the total-variation gradient at the cells of ``inputs`` (or of B uniform random points)
        to ``self.embeddings.grad``; call between ``loss.backward()`` and ``optimizer.step()``."""
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
            B = inputs.shape[0]
        if self.embeddings.grad is None:
            raise ValueError("grad is None, should be called after loss.backward() and before optimizer.step()!")
        inputs = inputs.to(torch.float32).contiguous()
        table, grad = self.embeddings.detach(), self.embeddings.grad
        require_gpu(inputs, table, grad)
        assert table.is_contiguous() and grad.is_contiguous() and grad.dtype == torch.float32
        rc = lib().pn_grad_total_variation(ptr(inputs), ptr(table), ptr(grad), self._offsets_host.data_ptr(), float(weight), B, self.input_dim,
                                           self.level_dim, self.num_levels, float(np.float32(np.log2(self.per_level_scale))), int(self.base_resolution),
                                           self.gridtype_id, int(bool(self.align_corners)), stream_ptr())
        check(rc, "grad_total_variation")
