"""Mirror of the reference's ``gridencoder`` package surface (inference only).

Interface replaced: /root/reference/gridencoder/grid.py:24-63 (``_grid_encode.forward``) and
:96-161 (``GridEncoder``): same call signatures, same module attributes / state-dict keys
(``offsets`` buffer, ``embeddings`` parameter).  Backward, total-variation and fp16 tables are
training-side and not part of the simulate-and-render path.
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


def grid_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False,
                interpolation=0, offsets_host=None):
    """inputs [B,D] in [0,1], embeddings [sO,C], offsets [L+1] -> features [B, L*C] fp32.

    One HIP launch writes [B, L*C] directly; the reference writes [L,B,C] and permutes (grid.py:47,57)."""
    if calc_grad_inputs:
        raise RuntimeError("grid_encode: dy_dx / backward are not part of the inference path")
    x = inputs.to(torch.float32).contiguous()
    table = embeddings.to(torch.float32).contiguous()
    require_gpu(x, table)
    n_levels = offsets.shape[0] - 1
    if offsets_host is None:
        offsets_host = offsets.detach().to("cpu", torch.int32).contiguous()
    feats = torch.empty(x.shape[0], n_levels * table.shape[1], device=x.device, dtype=torch.float32)
    log2_scale = float(np.float32(np.log2(per_level_scale)))  # the reference passes S = log2(per_level_scale) as a float (grid.py:37)
    rc = lib().pn_grid_encode_forward(ptr(x), ptr(table), offsets_host.data_ptr(), ptr(feats), x.shape[0], x.shape[1], table.shape[1], n_levels,
                                      log2_scale, int(base_resolution), None, int(gridtype), int(bool(align_corners)), int(interpolation), 1,
                                      stream_ptr())
    check(rc, "grid_encode_forward")
    return feats


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
