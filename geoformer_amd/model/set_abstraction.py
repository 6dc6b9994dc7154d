"""Set-abstraction stage: FPS -> ball query -> grouping -> shared MLP -> pooling.

Mirrors ``PointnetSAModuleVotesSeparate`` (lib/pointnet2/pointnet2_modules.py:150-249) and the
autograd wrappers of lib/pointnet2/pointnet2_utils.py:40-356 over the HIP operators.
Parameter names: ``mlp_module.layer{i}.conv.weight`` / ``mlp_module.layer{i}.bn.bn.*``
(lib/pointnet2/pytorch_utils.py:9-32,59-104).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import pointops
from . import layers
from .layers import BatchNorm2d, PointwiseConv2d


# ---- autograd wrappers (pointnet2_utils.py:40-68, 71-104, 190-236, 239-269) -------------------
class _FPS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        idx = pointops.furthest_point_sampling(xyz.contiguous(), npoint)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return pointops.gather_points(features.contiguous(), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return pointops.gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return pointops.group_points(features.contiguous(), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return pointops.group_points_grad(grad_out.contiguous(), idx, ctx.n), None


class _BallQuery(torch.autograd.Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        idx = pointops.ball_query(new_xyz.contiguous(), xyz.contiguous(), radius, nsample)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


furthest_point_sample = _FPS.apply
gather_operation = _Gather.apply
grouping_operation = _Group.apply
ball_query = _BallQuery.apply


class QueryAndGroup(nn.Module):
    """ball query + grouping; grouped xyz are centred on the query point and, with normalize_xyz,
    divided by the radius; features are not centred (pointnet2_utils.py:303-356)."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz, self.normalize_xyz = ret_grouped_xyz, normalize_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)  # (B,3,npoint,nsample)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.radius
        if features is not None:
            grouped = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
        else:
            new_features = grouped_xyz
        return (new_features, grouped_xyz) if self.ret_grouped_xyz else new_features


class _BN2d(nn.Sequential):
    def __init__(self, c):
        super().__init__()
        self.add_module("bn", BatchNorm2d(c))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class _ConvBNReLU2d(nn.Sequential):
    def __init__(self, cin, cout, act):
        super().__init__()
        conv = PointwiseConv2d(cin, cout, kernel_size=(1, 1), bias=False)
        nn.init.kaiming_normal_(conv.weight)
        self.add_module("conv", conv)
        self.add_module("bn", _BN2d(cout))
        self.add_module("activation", act)

    def forward(self, x):
        bn = self.bn.bn
        if bn.training and isinstance(self.activation, nn.ReLU) and layers._FUSED_BN:
            y = self.conv(x)
            if pointops.bn_train_cl_supported(bn, y):
                # BatchNorm + ReLU of the layer as one pass over the [B, C, npoint, nsample] values per direction
                # (csrc/bn_train.hip) instead of the norm's launches plus a clamp / threshold pass each way
                bn._nbt_pending = getattr(bn, "_nbt_pending", 0) + 1
                return pointops.bn_train_cl(bn, y, relu=True)
            return self.activation(self.bn(y))
        return super().forward(x)


class SharedMLP(nn.Sequential):
    def __init__(self, dims, bn=True):
        super().__init__()
        assert bn
        act = nn.ReLU(inplace=True)
        for i in range(len(dims) - 1):
            self.add_module(f"layer{i}", _ConvBNReLU2d(dims[i], dims[i + 1], act))


class PointnetSAModuleVotesSeparate(nn.Module):
    def __init__(self, *, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True, pooling="max",
                 sigma=None, normalize_xyz=False, sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        if sample_uniformly or ret_unique_cnt or npoint is None:
            raise NotImplementedError("only the configuration GeoFormer instantiates is implemented")
        self.npoint, self.radius, self.nsample, self.pooling = npoint, radius, nsample, pooling
        self.use_xyz, self.normalize_xyz = use_xyz, normalize_xyz
        self.sigma = sigma if sigma is not None else radius / 2
        self.grouper = QueryAndGroup(radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                                     normalize_xyz=normalize_xyz)
        dims = list(mlp)
        if use_xyz and dims:
            dims[0] += 3
        self.mlp_module = SharedMLP(dims, bn=bn)

    def group_points(self, xyz, features, inds=None, npoint_new=None):
        npoint = self.npoint if npoint_new is None else npoint_new
        if inds is None:
            inds = furthest_point_sample(xyz, npoint)  # always asks for npoint, even when n < npoint
        else:
            assert inds.shape[1] == npoint
        new_xyz = gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        grouped_features, grouped_xyz = self.grouper(xyz, new_xyz, features)
        return new_xyz, grouped_features, grouped_xyz, inds

    def _fused_chain(self):
        """Folded Conv2d(1x1)+BN+ReLU stack for the fused inference kernel (csrc/pointwise_mlp.hip), cached."""
        from .. import pointops

        walk = self.__dict__.get("_gf_chain_walk")
        if walk is None:  # the module tree is walked once; per call only BatchNorm modes and version counters
            flat = [m for _, m in self.mlp_module.named_modules(remove_duplicate=False) if len(list(m.children())) == 0]
            convs = [m for m in flat if isinstance(m, nn.Conv2d)]
            ok = (1 <= len(convs) <= 4) and not any(m.kernel_size != (1, 1) or m.groups != 1 for m in convs) and \
                not any(m.out_channels % 16 or m.out_channels > 64 for m in convs) and convs[0].in_channels <= 64
            walk = self.__dict__["_gf_chain_walk"] = (
                flat, [m for m in flat if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)],
                [p for m in flat for p in list(m.parameters()) + list(m.buffers())], ok)
        flat, bns, params, ok = walk
        if not ok or any(m.training for m in bns):
            return None
        key = (params[0].data_ptr(), sum([p._version for p in params]))
        hit = self.__dict__.get("_gf_chain")
        if hit is None or hit[0] != key:
            hit = (key, pointops.PointwiseChain(flat))
            self.__dict__["_gf_chain"] = hit
        return hit[1]

    def fused_forward(self, xyz, features, inds, grid=None):
        """(new_xyz, pooled features [B,C,npoint]) of the whole stage in two launches -- inference with max pooling on
        the GPU and given sample indices; None when that path does not apply (the caller then uses group_points + mlp)."""
        if self.pooling != "max" or not xyz.is_cuda or torch.is_grad_enabled() or inds is None:
            return None
        chain = self._fused_chain()
        if chain is None:
            return None
        from .. import pointops

        new_xyz, _, pooled = pointops.sa_group_mlp_max(xyz.contiguous(), features.contiguous(), inds.contiguous(),
                                                       self.radius, self.nsample, self.use_xyz, self.normalize_xyz, chain,
                                                       grid=grid)
        return new_xyz, pooled

    def mlp(self, grouped_features, grouped_xyz, pooling=None):
        pooling = pooling or self.pooling
        if pooling == "max" and grouped_features.is_cuda and not torch.is_grad_enabled():
            chain = self._fused_chain()
            if chain is not None:
                from .. import pointops

                return pointops.group_mlp_max(grouped_features.contiguous(), chain)
        x = self.mlp_module(grouped_features)  # (B, C, npoint, nsample)
        if pooling == "max":
            x = F.max_pool2d(x, kernel_size=[1, x.size(3)])
        elif pooling == "avg":
            x = F.avg_pool2d(x, kernel_size=[1, x.size(3)])
        elif pooling == "rbf":
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (self.sigma ** 2) / 2)
            x = torch.sum(x * rbf.unsqueeze(1), -1, keepdim=True) / float(self.nsample)
        else:
            raise ValueError(pooling)
        return x.squeeze(-1)
