"""NeRFNetwork with the reference's constructor, state-dict keys and forward (nerf/network.py:14-127).

``forward(x, d)`` runs the fused HIP kernel (hash grid + SH + both MLPs on the matrix cores, bf16 three-way split at fp32 accuracy); ``forward_ops`` is the same
computation op by op (grid_encode -> torch Linear -> sh_encode -> ...) like the reference: the parity tests use it, and it is the differentiable path
``forward`` takes in train() mode.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import check, lib, ptr, require_gpu, stream_ptr
from .activation import trunc_exp
from .encoding import get_encoder
from .renderer import NeRFRenderer


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", encoding_bg="hashgrid", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, num_layers_bg=2, hidden_dim_bg=64, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        if self.bg_radius > 0:
            raise NotImplementedError("background model (bg_radius > 0) is not on the simulate-and-render path (main_gui.py uses -1)")
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        dims = [self.in_dim] + [hidden_dim] * (num_layers - 1) + [1 + geo_feat_dim]  # 1 sigma + 15 geo features
        self.sigma_net = nn.ModuleList([nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(num_layers)])
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir)
        cdims = [self.in_dim_dir + geo_feat_dim] + [hidden_dim_color] * (num_layers_color - 1) + [3]
        self.color_net = nn.ModuleList([nn.Linear(cdims[i], cdims[i + 1], bias=False) for i in range(num_layers_color)])
        self.bg_net = None
        self._net = None
        self._net_sig = None
        self._net_dev = None
        self._net_half = False

    # ---- packed device context for the fused kernel; rebuilt when the parameters change
    def _signature(self):
        ts = [self.encoder.embeddings] + [l.weight for l in self.sigma_net] + [l.weight for l in self.color_net]
        return tuple((t.data_ptr(), t._version, str(t.device)) for t in ts)

    def _net_handle(self, half=False):
        """The packed device context (pn_net).  Created once; when a parameter changed since (optimizer step, checkpoint load) the packed
        weights are refreshed IN PLACE (pn_net_update: the weights are read back to the host — which waits for the current stream — packed into
        pinned staging and uploaded asynchronously; no allocation).  The upload is ordered on the CURRENT stream only: renders in flight on other
        streams (harness.capture_pipelined) would read a half-written weight image, so a refresh is refused while the owning harness reports frames
        in flight (`_in_flight`, set by the harness): drain_pipeline() first.  half=True additionally makes sure the fp16 tables exist."""
        sig = self._signature()
        if self._net is not None and sig != self._net_sig and getattr(self, "_in_flight", None) is not None and self._in_flight() > 0:
            raise RuntimeError("the network's parameters changed while pipelined frames are in flight: drain_pipeline() before updating weights")
        if self._net is None or sig != self._net_sig:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the network's parameters changed and its packed weights cannot be refreshed while the stream is being captured into "
                                   "a HIP graph (the packing reads them on the host); evaluate the network once before capture")
            if (self.num_layers, self.hidden_dim, self.geo_feat_dim, self.num_layers_color, self.hidden_dim_color) != (2, 64, 15, 3, 64):
                raise RuntimeError("fused network kernel implements the reference architecture only (32->64->16 | 31->64->64->3)")
            emb = self.encoder.embeddings
            require_gpu(emb)
            assert emb.dtype == torch.float32 and emb.is_contiguous()
            Ws = [np.ascontiguousarray(l.weight.detach().cpu().numpy(), dtype=np.float32) for l in list(self.sigma_net) + list(self.color_net)]
            wp = [w.ctypes.data for w in Ws]
            if self._net is None or self._net_dev != str(emb.device):
                if self._net is not None:
                    lib().pn_net_destroy(self._net)
                off = self.encoder._offsets_host
                h = C.c_void_p()
                check(lib().pn_net_create(C.byref(h), ptr(emb), off.data_ptr(), self.encoder.num_levels, self.encoder.level_dim,
                                          float(np.float32(np.log2(self.encoder.per_level_scale))), int(self.encoder.base_resolution), float(self.bound),
                                          wp[0], wp[1], wp[2], wp[3], wp[4], stream_ptr()), "net_create")
                self._net, self._net_dev, self._net_half = h, str(emb.device), False
            else:
                check(lib().pn_net_update(self._net, ptr(emb), wp[0], wp[1], wp[2], wp[3], wp[4], stream_ptr()), "net_update")
            self._net_sig = sig
        if half and not self._net_half:
            check(lib().pn_net_enable_half(self._net, stream_ptr()), "net_enable_half")
            self._net_half = True
        return self._net

    @staticmethod
    def _autocast_half():
        """True when the caller runs under torch.cuda.amp.autocast with fp16 (trainer.py:561, Trainer(fp16=True)): the reference then casts
        the hash table to half (gridencoder/grid.py:43-44) and every nn.Linear computes in half."""
        if not torch.is_autocast_enabled("cuda"):
            return False
        if torch.get_autocast_dtype("cuda") != torch.float16:
            # the grid encoder follows the reference and casts its table to half under ANY autocast (grid.py:43-44) while the layers would run
            # in the other type: a combination the reference never uses and nothing here implements
            raise RuntimeError("autocast with a dtype other than float16 is not implemented (the reference's Trainer uses fp16, trainer.py:561)")
        return True

    def _wants_grad(self, *inputs):
        """Differentiable path only in train() mode with autograd recording (Trainer.train_one_epoch calls model.train(), evaluate /
        test call model.eval(): trainer.py:655,747); everything else — the simulate-and-render path — takes the fused kernel."""
        return self.training and torch.is_grad_enabled()

    def forward(self, x, d):
        """x [N,3] in [-bound,bound], d [N,3] unit -> sigma [N], color [N,3] (network.py:98-127).  Inference (no_grad, the
        simulate-and-render path): one fused launch.  With autograd recording (training): the differentiable op
        sequence of ``forward_ops`` — HIP encoders with their backward kernels, rocBLAS for the five small GEMMs."""
        if self._wants_grad(x, d):
            return self.forward_ops(x, d)
        x = x.to(torch.float32).contiguous().view(-1, 3)
        d = d.to(torch.float32).contiguous().view(-1, 3)
        require_gpu(x, d)
        M = x.shape[0]
        sigma = torch.empty(M, dtype=torch.float32, device=x.device)
        color = torch.empty(M, 3, dtype=torch.float32, device=x.device)
        if self._autocast_half():
            # fp16 tables + fp16 MFMA layers with half-rounded activations; sigma stays float (trunc_exp casts, activation.py:7), the colour
            # is a half tensor like the reference's sigmoid of a half input (the kernel writes the half values exactly representable in fp32)
            check(lib().pn_nerf_forward_half(self._net_handle(half=True), ptr(x), ptr(d), M, 1.0, ptr(sigma), ptr(color), stream_ptr()), "nerf_forward_half")
            return sigma, color.to(torch.float16)
        check(lib().pn_nerf_forward(self._net_handle(), ptr(x), ptr(d), M, 1.0, ptr(sigma), ptr(color), stream_ptr()), "nerf_forward")
        return sigma, color

    def density(self, x):
        """x [N,3] in [-bound,bound] -> {'sigma': [N], 'geo_feat': [N,15]} (network.py:129-146), the fused kernel stopped after the
        sigma net."""
        if self._wants_grad(x):
            h = self.encoder(x, bound=self.bound)
            for i, layer in enumerate(self.sigma_net):
                h = layer(h)
                if i != self.num_layers - 1:
                    h = F.relu(h, inplace=True)
            return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}
        x = x.to(torch.float32).contiguous().view(-1, 3)
        require_gpu(x)
        M = x.shape[0]
        sigma = torch.empty(M, dtype=torch.float32, device=x.device)
        geo = torch.empty(M, 15, dtype=torch.float32, device=x.device)
        if self._autocast_half():
            check(lib().pn_nerf_density_half(self._net_handle(half=True), ptr(x), M, ptr(sigma), ptr(geo), stream_ptr()), "nerf_density_half")
            return {"sigma": sigma, "geo_feat": geo.to(torch.float16)}
        check(lib().pn_nerf_density(self._net_handle(), ptr(x), M, ptr(sigma), ptr(geo), stream_ptr()), "nerf_density")
        return {"sigma": sigma, "geo_feat": geo}

    def forward_ops(self, x, d):
        """The reference's op sequence: GridEncoder -> Linear/ReLU -> exp | SHEncoder, cat -> Linear/ReLU x3 -> sigmoid."""
        h = self.encoder(x, bound=self.bound)
        for i, layer in enumerate(self.sigma_net):
            h = layer(h)
            if i != self.num_layers - 1:
                h = F.relu(h, inplace=True)
        sigma = trunc_exp(h[..., 0])  # nerf/activation.py:5-18
        h = torch.cat([self.encoder_dir(d), h[..., 1:]], dim=-1)
        for i, layer in enumerate(self.color_net):
            h = layer(h)
            if i != self.num_layers_color - 1:
                h = F.relu(h, inplace=True)
        return sigma, torch.sigmoid(h)

    def get_params(self, lr):
        """network.py:196-207: optimizer parameter groups."""
        return [{"params": self.encoder.parameters(), "lr": lr}, {"params": self.sigma_net.parameters(), "lr": lr},
                {"params": self.encoder_dir.parameters(), "lr": lr}, {"params": self.color_net.parameters(), "lr": lr}]

    def load_checkpoint_dict(self, ck):
        """Loads the synthetic checkpoint dict of pienerf_amd.scene.make_checkpoint (same tensors as the reference's
        state dict: encoder.embeddings, sigma_net.{0,1}.weight, color_net.{0,1,2}.weight, density_bitfield)."""
        dev = self.encoder.embeddings.device
        with torch.no_grad():
            assert tuple(self.encoder.offsets.cpu().numpy()) == tuple(np.asarray(ck["offsets"])), "hash-grid geometry mismatch"
            self.encoder.embeddings.copy_(torch.from_numpy(ck["embeddings"]).to(dev))
            for layer, key in zip(list(self.sigma_net) + list(self.color_net), ("W0", "W1", "W2", "W3", "W4")):
                layer.weight.copy_(torch.from_numpy(ck[key]).to(dev))
            self.density_bitfield.copy_(torch.from_numpy(ck["density_bitfield"]).to(dev))
        self._net_sig = None
        return self.eval()  # a loaded checkpoint is used for inference; training code calls .train() itself
