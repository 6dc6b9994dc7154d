"""Module shims that let the REFERENCE's own Python (imported from /root/reference, in the build
container only) run on CPU: its four native dependencies (spconv, PG_OP, pointnet2._ext, faiss)
are stood in for by the CPU oracle.  Used only by make_golden.py to generate fixtures -- test
infrastructure, never imported by the product package and never shipped reference code.
"""
from __future__ import annotations

import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


# ------------------------------------------------------------------------------------------
# spconv (semantics: SURVEY.md Appendix A)
# ------------------------------------------------------------------------------------------
class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid

    def find_indice_pair(self, key):
        return self.indice_dict.get(key) if key is not None else None


class SparseModule(nn.Module):
    pass


def _is_sparse(m):
    return isinstance(m, SparseModule)


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def forward(self, input):
        for m in self._modules.values():
            if _is_sparse(m):
                input = m(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input.features = m(input.features)
            else:
                input = m(input)
        return input


class _ConvFn(torch.autograd.Function):
    """out[o] = sum_k in[nbr[k][o]] @ W[k] over an output-stationary table, with the backward the
    training golden needs: dIn[i] += dOut[o] @ W[k]^T, dW[k] = sum in[i]^T dOut[o] (oracle operators)."""

    @staticmethod
    def forward(ctx, feats, W, nbr, M_out):
        ctx.save_for_backward(feats, W)
        ctx.nbr = nbr
        return torch.from_numpy(orc.conv_fwd(feats.detach().numpy(), W.detach().numpy(), nbr, M_out))

    @staticmethod
    def backward(ctx, g):
        feats, W = ctx.saved_tensors
        g = np.ascontiguousarray(g.detach().numpy())
        d_in = torch.from_numpy(orc.conv_dgrad(g, W.detach().numpy(), ctx.nbr, feats.shape[0]))
        d_w = torch.from_numpy(orc.conv_wgrad(feats.detach().numpy(), g, ctx.nbr, W.shape[0]))
        return d_in, d_w, None, None


class _Conv(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, bias=False, indice_key=None,
                 kind="subm"):
        super().__init__()
        ks = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.in_channels, self.out_channels, self.ks, self.kind, self.indice_key = in_channels, out_channels, ks, kind, indice_key
        self.weight = nn.Parameter(torch.empty(ks, ks, ks, in_channels, out_channels))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        assert not bias

    def forward(self, input):
        feats = input.features
        W = self.weight.reshape(-1, self.in_channels, self.out_channels)
        coords = input.indices.numpy().astype(np.int32)
        shape = input.spatial_shape
        out = SparseConvTensor(None, input.indices, shape, input.batch_size)
        out.indice_dict = input.indice_dict
        if self.ks == 1:
            out.features = feats @ W[0]
            return out
        if self.kind == "subm":
            nbr = input.indice_dict.get(self.indice_key)
            if nbr is None:
                nbr = orc.rules_subm3(coords, shape)
                input.indice_dict[self.indice_key] = nbr
            out.features = _ConvFn.apply(feats, W, nbr, coords.shape[0])
        elif self.kind == "down":
            oc, child, parent, koff = orc.rules_down2(coords, shape)
            input.indice_dict[self.indice_key] = (coords, shape, parent, koff, input.indices)
            out.features = _ConvFn.apply(feats, W, child, oc.shape[0])
            out.indices = torch.from_numpy(oc)
            out.spatial_shape = [(s - 2) // 2 + 1 for s in shape]
        else:  # inverse
            coords_f, shape_f, parent, koff, ind_f = input.indice_dict[self.indice_key]
            up = orc.up_table(parent, koff)
            out.features = _ConvFn.apply(feats, W, up, coords_f.shape[0])
            out.indices = ind_f
            out.spatial_shape = list(shape_f)
        return out


class SubMConv3d(_Conv):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key, kind="subm")


class SparseConv3d(_Conv):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key, kind="down")


class SparseInverseConv3d(_Conv):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key, kind="inverse")


# ------------------------------------------------------------------------------------------
# PG_OP / pointnet2._ext / faiss
# ------------------------------------------------------------------------------------------
def _pg_voxelize_idx(coords, output_coords, input_map, output_map, batch_size, mode):
    oc, im, om = orc.voxelize_idx(coords.numpy(), mode)
    output_coords.resize_(oc.shape).copy_(torch.from_numpy(oc))
    input_map.copy_(torch.from_numpy(im))
    output_map.resize_(om.shape).copy_(torch.from_numpy(om))


def _pg_voxelize_fp(feats, out, rules, mode, M, maxActive, C):
    out.copy_(torch.from_numpy(orc.voxelize_fp(feats.detach().numpy(), rules.numpy(), mode == 4)))


def _pg_voxelize_bp(d_out, d_feats, rules, mode, M, maxActive, C):
    d_feats.copy_(torch.from_numpy(orc.voxelize_bp(d_out.detach().numpy(), rules.numpy(), d_feats.shape[0], mode == 4)))


def _ext_gather_grad(grad_out, idx, n):
    return torch.from_numpy(orc.gather_points_grad(grad_out.detach().numpy(), idx.numpy(), n))


def _ext_group_grad(grad_out, idx, n):
    return torch.from_numpy(orc.group_points_grad(np.ascontiguousarray(grad_out.detach().numpy()), idx.numpy(), n))


def _ext_fps(xyz, m):
    return torch.from_numpy(orc.fps(xyz.detach().numpy(), m))


def _ext_gather(points, idx):
    return torch.from_numpy(orc.gather_points(points.detach().numpy(), idx.numpy()))


def _ext_ball_query(new_xyz, xyz, radius, nsample):
    return torch.from_numpy(orc.ball_query(new_xyz.detach().numpy(), xyz.detach().numpy(), radius, nsample))


def _ext_group(points, idx):
    return torch.from_numpy(orc.group_points(points.detach().numpy(), idx.numpy()))


class _FaissIndex:
    """GpuIndexFlatL2 stand-in: exact fp32 squared L2, ties by lower index (oracle definition)."""

    def __init__(self, res, dim, cfg):
        self.base = None

    def add(self, x):
        self.base = x.detach().numpy().copy()

    def search(self, x, k, D, I):
        d, i = orc.knn(self.base, x.detach().numpy(), k)
        D.copy_(torch.from_numpy(d))
        I.copy_(torch.from_numpy(i))

    def reset(self):
        self.base = None


def install(reference_root="/root/reference"):
    """Register the shim modules and make CUDA-only idioms of the reference no-ops on CPU."""
    sp = types.ModuleType("spconv")
    for name in ("SparseConvTensor", "SparseSequential", "SubMConv3d", "SparseConv3d", "SparseInverseConv3d"):
        setattr(sp, name, globals()[name])
    spm = types.ModuleType("spconv.modules")
    spm.SparseModule = SparseModule
    sp.modules = spm
    sys.modules["spconv"], sys.modules["spconv.modules"] = sp, spm

    pg = types.ModuleType("PG_OP")
    pg.voxelize_idx, pg.voxelize_fp, pg.voxelize_bp = _pg_voxelize_idx, _pg_voxelize_fp, _pg_voxelize_bp
    sys.modules["PG_OP"] = pg

    p2 = types.ModuleType("pointnet2")
    ext = types.ModuleType("pointnet2._ext")
    ext.furthest_point_sampling, ext.gather_points = _ext_fps, _ext_gather
    ext.ball_query, ext.group_points = _ext_ball_query, _ext_group
    ext.gather_points_grad, ext.group_points_grad = _ext_gather_grad, _ext_group_grad
    p2._ext = ext
    sys.modules["pointnet2"], sys.modules["pointnet2._ext"] = p2, ext

    fa = types.ModuleType("faiss")
    fa.GpuIndexFlatConfig = lambda: types.SimpleNamespace(useFloat16=False, device=0)
    fa.StandardGpuResources = lambda: object()
    fa.GpuIndexFlatL2 = _FaissIndex
    fc = types.ModuleType("faiss.contrib")
    ft = types.ModuleType("faiss.contrib.torch_utils")
    fa.contrib, fc.torch_utils = fc, ft
    sys.modules["faiss"], sys.modules["faiss.contrib"], sys.modules["faiss.contrib.torch_utils"] = fa, fc, ft
    for dummy in ("trimesh", "tensorboardX"):
        sys.modules.setdefault(dummy, types.ModuleType(dummy))

    # torch's AVX-512 CPU sqrt is NOT correctly rounded (1 ulp off on ~0.6 % of inputs, measured here),
    # while the sqrt the reference runs on a GPU (and HIP's / numpy's) is IEEE-correct.  The geodesic
    # distances are fp32 sums of sqrt'ed edge lengths compared bit for bit, so the generator evaluates
    # sqrt through numpy to reflect the reference's device behaviour rather than a CPU artefact.
    _sqrt = torch.sqrt
    torch.sqrt = lambda t, *a, **k: (torch.from_numpy(np.sqrt(t.detach().numpy())) if (not a and not k and t.dtype == torch.float32 and not t.requires_grad) else _sqrt(t, *a, **k))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = lambda *shape: torch.zeros(*shape, dtype=torch.float32)
    torch.cuda.IntTensor = lambda *shape: torch.zeros(*shape, dtype=torch.int32)
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
