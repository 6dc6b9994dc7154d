"""Drop-in for the ``spconv`` 1.0 Python surface GeoFormer uses, on MI355X.

Exports exactly what the reference touches (geoformer.py:4,42-53,398;
geoformer_modules.py:2,6,15-35,58-109): ``SparseConvTensor``, ``SparseSequential``,
``SubMConv3d``, ``SparseConv3d``, ``SparseInverseConv3d`` and ``modules.SparseModule``.
Parameters keep spconv's name and layout (``weight`` of shape [k,k,k,Cin,Cout], no bias), so
reference checkpoints load unchanged (checkpoint.py:10-66).

Rulebooks are built by the HIP bitmap-rank builder and cached in the tensor's shared
``indice_dict`` under the layer's ``indice_key`` exactly like spconv does; all arithmetic runs
in libgeoformer_hip.so (``geoformer_amd.sparse``).  There is no CPU path.

To use under the unmodified reference:  ``import geoformer_amd.dropin as d; d.install()``
registers this package as ``spconv`` in ``sys.modules``.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn as nn

import os

from .. import pointops, sparse
from . import modules
from .modules import SparseModule

_fused_bn = os.environ.get("GF_FUSED_BN", "1") != "0"  # dev knob: the framework's BatchNorm / ReLU kernels instead

__all__ = ["SparseConvTensor", "SparseSequential", "SubMConv3d", "SparseConv3d", "SparseInverseConv3d", "modules"]


class SparseConvTensor:
    """features fp32 [M,C], indices int32 [M,4] (batch,x,y,z), spatial_shape (3), batch_size.
    Plain mutable attributes: the reference assigns ``.features`` in place
    (geoformer_modules.py:33,116,127)."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid
        self._index = None  # occupancy index of this voxel set (built lazily)

    @property
    def spatial_size(self):
        return int(self.spatial_shape[0] * self.spatial_shape[1] * self.spatial_shape[2])

    def find_indice_pair(self, key):
        return self.indice_dict.get(key) if key is not None else None

    def _coords(self):
        c = self.indices
        if c.dtype != torch.int32 or not c.is_contiguous():
            c = c.int().contiguous()
            self.indices = c
        return c

    def _level_index(self):
        if self._index is None:
            self._index = sparse.build_index(self._coords(), self.batch_size, self.spatial_shape)
        return self._index

    def dense(self, channels_first=True):
        c = self.indices.long()
        out = torch.zeros([self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]],
                          dtype=self.features.dtype, device=self.features.device)
        out[c[:, 0], c[:, 1], c[:, 2], c[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out


class SparseSequential(SparseModule):
    """Accepts positional modules or one OrderedDict (geoformer_modules.py:58-63).  Sparse
    modules get the tensor; any other nn.Module is applied to ``.features`` and assigned back on
    the SAME object (skipped when the tensor is empty)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for i, module in enumerate(args):
                self.add_module(str(i), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError(f"index {idx} is out of range")
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def forward(self, input):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            i += 1
            if isinstance(module, SparseModule):
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    # training: BatchNorm1d + ReLU (the pre-activation pair in front of every convolution,
                    # geoformer_modules.py:15-27) as three launches per direction (csrc/bn_train.hip)
                    if (i < len(mods) and isinstance(module, nn.BatchNorm1d) and type(mods[i]) is nn.ReLU
                            and _fused_bn and pointops.bn_relu_train_supported(module, input.features)):
                        input.features = pointops.bn_relu_train(module, input.features.contiguous())
                        i += 1
                    else:
                        input.features = module(input.features)
                elif getattr(module, "sync_across_ranks", False) and module.training:
                    # spconv skips dense modules on an empty tensor; a layer whose statistics span ranks must still be
                    # entered (count 0): the other ranks are waiting in its collective (parallel.SyncBatchNorm1d)
                    input.features = module(input.features)
            else:
                input = module(input)
        return input


class _GatherConv(torch.autograd.Function):
    """out = gather-GEMM(feats, weight) through a neighbour table; backward through the transposed
    table (same kernel) for the input gradient and gf_conv_wgrad for the weight gradient."""

    @staticmethod
    def forward(ctx, feats, weight, fwd, bwd):
        tbl, gmask, K, M_out, ld = fwd[:5]
        out = sparse.conv_fwd(feats.contiguous(), weight, tbl, gmask, K, M_out, ld,
                              **({"steps": fwd[5]} if len(fwd) > 5 and fwd[5] is not None else {}))
        ctx.save_for_backward(feats, weight)
        ctx.fwd, ctx.bwd = fwd, bwd
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feats, weight = ctx.saved_tensors
        tbl, gmask, K, M_out, ld = ctx.fwd[:5]
        g = grad_out.contiguous()
        d_feats = d_weight = None
        if ctx.needs_input_grad[0]:
            d_feats = sparse.conv_dgrad(g, weight, ctx.bwd, feats.shape[0])
        if ctx.needs_input_grad[1]:
            d_weight = sparse.conv_wgrad(feats.contiguous(), g, tbl, K, M_out, ld, gmask=gmask).view_as(weight)
        return d_feats, d_weight, None, None


class _SparseConvBase(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, bias, indice_key):
        super().__init__()
        ks = kernel_size if isinstance(kernel_size, int) else int(kernel_size[0])
        if not isinstance(kernel_size, int) and any(int(k) != ks for k in kernel_size):
            raise NotImplementedError("only cubic kernels are implemented")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = [ks, ks, ks]
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(ks, ks, ks, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] ** 3
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def _finish(self, out_tensor, feats):
        if self.bias is not None:
            feats = feats + self.bias
        out_tensor.features = feats
        return out_tensor

    def _new_like(self, input, indices=None, spatial_shape=None):
        out = SparseConvTensor(None, input.indices if indices is None else indices,
                               input.spatial_shape if spatial_shape is None else spatial_shape, input.batch_size)
        out.indice_dict = input.indice_dict  # shared dict object: how inverse convs find their rules
        out.grid = input.grid
        return out

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, indice_key={self.indice_key}"


class SubMConv3d(_SparseConvBase):
    """Submanifold conv: output sites == input sites.  k=1 is a plain GEMM on the features
    (geoformer_modules.py:17-19); k=3/padding=1 uses the 27-offset neighbour table."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, bias, indice_key)
        ks = self.kernel_size[0]
        pad = padding if isinstance(padding, int) else int(padding[0])
        if ks not in (1, 3) or (ks == 3 and pad != 1) or stride != 1 or dilation != 1 or groups != 1:
            raise NotImplementedError("SubMConv3d: implemented for k=1 and k=3/padding=1, stride 1 (what GeoFormer uses)")

    def get_rules(self, input):
        """The 27-offset neighbour table of this tensor's voxel set, cached under ``indice_key``."""
        rules = input.find_indice_pair(self.indice_key)
        if rules is None:
            rules = sparse.subm_rules(input._coords(), input._level_index())
            if self.indice_key is not None:
                input.indice_dict[self.indice_key] = rules
        return rules

    def forward(self, input):
        out = self._new_like(input)
        out._index = input._index
        M = input.indices.shape[0]
        if self.kernel_size[0] == 1:
            w2 = self.weight.view(self.in_channels, self.out_channels)
            f = input.features
            if f.is_cuda and torch.is_grad_enabled() and f.shape[0] >= (1 << 14):
                # 1x1x1 convolution = a GEMM whose weight gradient reduces over every voxel: split over row chunks
                # (the library runs the 523 008-long reduction on a handful of workgroups, 0.9 ms per layer)
                from ..model.layers import _SplitKLinearFn

                return self._finish(out, _SplitKLinearFn.apply(f, w2.t(), None))
            return self._finish(out, torch.mm(f, w2))
        if M == 0:
            return self._finish(out, input.features.new_zeros((0, self.out_channels)))
        rules = self.get_rules(input)
        out._index = input._index
        if (rules.steps is not None and self.in_channels < 16 and self.out_channels == 16 and self.bias is None
                and not torch.is_grad_enabled() and input.features.is_cuda):
            # inference, narrow input (the 6 -> 16 input conv of the U-Net): rows and weights zero-padded to one
            # 16-channel chunk so that the launch takes the counted-loop kernel (36 -> ~19 us at S150k)
            key = (self.weight.data_ptr(), self.weight._version)
            hit = self.__dict__.get("_gf_w16")
            if hit is None or hit[0] != key:
                w16 = torch.nn.functional.pad(self.weight.detach().reshape(27, self.in_channels, 16),
                                              (0, 0, 0, 16 - self.in_channels)).contiguous()
                hit = self.__dict__["_gf_w16"] = (key, w16)
            x16 = torch.nn.functional.pad(input.features, (0, 16 - self.in_channels))
            return self._finish(out, sparse.conv_fwd(x16, hit[1], rules.nbr, rules.gmask, 27, M, rules.ld,
                                                     steps=rules.steps))
        spec = (rules.nbr, rules.gmask, 27, M, rules.ld, rules.steps)
        return self._finish(out, _GatherConv.apply(input.features, self.weight, spec, ("subm", spec)))


class SparseConv3d(_SparseConvBase):
    """Strided conv; implemented for kernel_size=2, stride=2, padding=0 (geoformer_modules.py:77-84).
    Output voxels come in ascending linearised (batch,x,y,z) order."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, bias, indice_key)
        st = stride if isinstance(stride, int) else int(stride[0])
        pad = padding if isinstance(padding, int) else int(padding[0])
        if self.kernel_size[0] != 2 or st != 2 or pad != 0 or dilation != 1 or groups != 1:
            raise NotImplementedError("SparseConv3d: implemented for kernel_size=2, stride=2, padding=0")

    def get_rules(self, input):
        hook = input.indice_dict.pop("_prebuild", None)
        if hook is not None:
            hook(input)  # a caller that knows the whole down-sampling chain builds it in one go (one host sync)
        pre = input.find_indice_pair(self.indice_key)
        if isinstance(pre, sparse.DownRules) and getattr(pre, "prebuilt_for", None) == input.indices.data_ptr() \
                and pre.M_in == input.indices.shape[0]:
            # rulebook chain prebuilt for exactly this voxel set (sparse.down_rules_chain): no host sync here
            pre.in_coords, pre.in_shape, pre.in_index = input.indices, list(input.spatial_shape), input._index
            return pre
        rules = sparse.down_rules(input._coords(), input.batch_size, input.spatial_shape)
        rules.in_coords, rules.in_shape, rules.in_index = input.indices, list(input.spatial_shape), input._index
        if self.indice_key is not None:
            input.indice_dict[self.indice_key] = rules
        return rules

    def output_tensor(self, input, rules):
        out = self._new_like(input, rules.out_coords, list(rules.out_shape))
        out._index = rules.index_out
        return out

    def forward(self, input):
        rules = self.get_rules(input)
        out = self.output_tensor(input, rules)
        fwd = (rules.child, rules.gmask_down, 8, rules.M_out, rules.ld)
        bwd = ("table", (rules.up, rules.gmask_up, 8, rules.M_in, rules.ld_up))
        return self._finish(out, _GatherConv.apply(input.features, self.weight, fwd, bwd))


class SparseInverseConv3d(_SparseConvBase):
    """Inverse of the SparseConv3d that shares its ``indice_key``: restores that conv's input voxel
    set and row order (geoformer_modules.py:91-97); rows that had no output cell stay zero."""

    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, bias, indice_key)
        if self.kernel_size[0] != 2:
            raise NotImplementedError("SparseInverseConv3d: implemented for kernel_size=2")

    def get_rules(self, input):
        rules = input.find_indice_pair(self.indice_key)
        if rules is None or not isinstance(rules, sparse.DownRules):
            raise RuntimeError(f"SparseInverseConv3d: no SparseConv3d rules under indice_key={self.indice_key!r}")
        return rules

    def output_tensor(self, input, rules):
        out = self._new_like(input, rules.in_coords, rules.in_shape)
        out._index = rules.in_index
        return out

    def forward(self, input):
        rules = self.get_rules(input)
        out = self.output_tensor(input, rules)
        fwd = (rules.up, rules.gmask_up, 8, rules.M_in, rules.ld_up)
        bwd = ("table", (rules.child, rules.gmask_down, 8, rules.M_out, rules.ld))
        return self._finish(out, _GatherConv.apply(input.features, self.weight, fwd, bwd))
