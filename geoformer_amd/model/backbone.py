"""Sparse-voxel U-Net blocks (geoformer_modules.py:10-35, 52-129) over the drop-in ``spconv``."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .. import spconv
from ..spconv.modules import SparseModule
from .layers import BackboneTransformer, BatchNorm1d, PointwiseConv1d, scene_counts


def bn_affine(bn):
    """Eval-mode BatchNorm1d as y = x * scale + shift (cached until a parameter or statistic changes)."""
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr())
    hit = getattr(bn, "_gf_affine", None)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
            shift = (bn.bias - bn.running_mean * scale).float().contiguous()
        hit = (key, scale, shift)
        bn._gf_affine = hit
    return hit[1], hit[2]


def _fusable(t, *bns):
    """Inference on the GPU with frozen statistics: BN+ReLU ride in the conv kernel's prologue and the
    residual add in its epilogue (csrc/spconv_conv.hip) instead of costing HBM round trips and launches."""
    return (not torch.is_grad_enabled()) and t.features.is_cuda and t.indices.shape[0] > 0 and \
        all(not bn.training for bn in bns)


def _train_transformer_ok(t, transformer):
    from .. import pointops

    return pointops.backbone_transformer_train_supported(t.features, t._coords(), transformer)


class ResidualBlock(SparseModule):
    """out = conv_branch(x) + i_branch(x); i_branch is Identity or a 1x1x1 conv when the widths differ.
    conv_branch = [BN, ReLU, SubM3, BN, ReLU, SubM3] (pre-activation)."""

    def __init__(self, in_channels, out_channels, norm_fn, indice_key=None):
        super().__init__()
        if in_channels == out_channels:
            self.i_branch = spconv.SparseSequential(nn.Identity())
        else:
            self.i_branch = spconv.SparseSequential(
                spconv.SubMConv3d(in_channels, out_channels, kernel_size=1, bias=False))
        self.conv_branch = spconv.SparseSequential(
            norm_fn(in_channels), nn.ReLU(),
            spconv.SubMConv3d(in_channels, out_channels, kernel_size=3, padding=1, bias=False, indice_key=indice_key),
            norm_fn(out_channels), nn.ReLU(),
            spconv.SubMConv3d(out_channels, out_channels, kernel_size=3, padding=1, bias=False, indice_key=indice_key),
        )

    def _fused_params(self):
        """Packed weights and folded BatchNorm of the block, rebuilt when any of them changes (one combined key)."""
        from .. import sparse

        # the tensors the derived copies depend on, looked up once (attribute access on a Module goes through
        # __getattr__; 26 blocks x a dozen lookups per forward is measurable in the launch-bound U-Net)
        cached = self.__dict__.get("_gf_block_t")
        if cached is None or cached[1]._parameters["weight"] is not cached[0][0]:
            bn0, _, conv0, bn1, _, conv1 = list(self.conv_branch._modules.values())
            ib = self.i_branch[0]
            wi = None if isinstance(ib, nn.Identity) else ib.weight
            tensors = [conv0.weight, conv1.weight, bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var, bn1.weight,
                       bn1.bias, bn1.running_mean, bn1.running_var] + ([] if wi is None else [wi])
            cached = self.__dict__["_gf_block_t"] = (tensors, conv0, (bn0, conv0, bn1, conv1, ib, wi))
        tensors, _, (bn0, conv0, bn1, conv1, ib, wi) = cached
        key = (tensors[0].data_ptr(), [t._version for t in tensors])
        hit = self.__dict__.get("_gf_block")
        if hit is None or hit[0] != key:
            s0, t0 = bn_affine(bn0)
            s1, t1 = bn_affine(bn1)
            wpi = None if wi is None else sparse.pack_weights(wi.view(1, ib.in_channels, ib.out_channels))
            hit = (key, sparse.pack_weights(conv0.weight), sparse.pack_weights(conv1.weight), wpi, s0, t0, s1, t1,
                   conv0, conv1)
            self.__dict__["_gf_block"] = hit
        return hit

    def _forward_fused(self, input):
        from .. import sparse

        _, wp0, wp1, wpi, s0, t0, s1, t1, conv0, conv1 = self._fused_params()
        rules = conv0.get_rules(input)
        M = input.indices.shape[0]
        out = conv1._new_like(input)
        out._index = input._index
        out.features = sparse.resblock_fwd(input.features.contiguous(), wp0, wp1, wpi, rules.nbr, rules.gmask, 27, M,
                                           rules.ld, conv0.in_channels, conv0.out_channels, s0, t0, s1, t1,
                                           steps=rules.steps)
        return out

    def forward(self, input):
        mods = list(self.conv_branch._modules.values())
        if _fusable(input, mods[0], mods[3]):
            return self._forward_fused(input)
        # snapshot BEFORE conv_branch: SparseSequential replaces input.features in place
        identity = spconv.SparseConvTensor(input.features, input.indices, input.spatial_shape, input.batch_size)
        output = self.conv_branch(input)
        output.features = output.features + self.i_branch(identity).features
        return output


class UBlock(nn.Module):
    """Recursive U-Net level: blocks -> [down conv -> UBlock -> inverse conv -> concat -> blocks_tail];
    the two deepest levels add a dense per-scene transformer (geoformer_modules.py:64-68,120-127)."""

    def __init__(self, nPlanes, norm_fn, block_reps, block, use_backbone_transformer=False, indice_key_id=1):
        super().__init__()
        self.nPlanes = nPlanes
        c = nPlanes[0]
        self.blocks = spconv.SparseSequential(OrderedDict(
            (f"block{i}", block(c, c, norm_fn, indice_key=f"subm{indice_key_id}")) for i in range(block_reps)))
        if len(nPlanes) <= 2 and use_backbone_transformer:
            self.before_transformer_linear = nn.Linear(c, 128)
            self.transformer = BackboneTransformer(d_model=128, N=2, heads=4, d_ff=64)
            self.after_transformer_linear = nn.Linear(128, c)
        else:
            self.before_transformer_linear = self.transformer = self.after_transformer_linear = None
        if len(nPlanes) > 1:
            key = f"spconv{indice_key_id}"
            self.conv = spconv.SparseSequential(
                norm_fn(c), nn.ReLU(),
                spconv.SparseConv3d(c, nPlanes[1], kernel_size=2, stride=2, bias=False, indice_key=key))
            self.u = UBlock(nPlanes[1:], norm_fn, block_reps, block, use_backbone_transformer,
                            indice_key_id=indice_key_id + 1)
            self.deconv = spconv.SparseSequential(
                norm_fn(nPlanes[1]), nn.ReLU(),
                spconv.SparseInverseConv3d(nPlanes[1], c, kernel_size=2, bias=False, indice_key=key))
            self.blocks_tail = spconv.SparseSequential(OrderedDict(
                (f"block{i}", block(c * (2 - i), c, norm_fn, indice_key=f"subm{indice_key_id}"))
                for i in range(block_reps)))

    def _transformer_fused(self, t):
        """Inference: before-linear, the per-scene transformer and after-linear as one HIP launch
        (csrc/backbone_attn.hip) instead of ~75 small ones."""
        from .. import pointops

        key = (self.before_transformer_linear.weight.data_ptr(), self.after_transformer_linear.bias.data_ptr())
        hit = getattr(self, "_gf_tr_params", None)
        if hit is None or hit[0] != key:
            table, nl = pointops.backbone_transformer_params(self.before_transformer_linear, self.transformer,
                                                             self.after_transformer_linear)
            hit = (key, table, nl)
            self._gf_tr_params = hit
        coords = t._coords()
        M = coords.shape[0]
        if t.batch_size == 1:
            offs = getattr(self, "_gf_offs1", None)
            if offs is None or offs[0] != M or offs[1].device != coords.device:
                offs = (M, torch.tensor([0, M], dtype=torch.int32, device=coords.device))
                self._gf_offs1 = offs
            offs = offs[1]
        else:
            counts = scene_counts(coords[:, 0], t.batch_size)
            offs = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).int()
        return pointops.backbone_transformer(t.features.contiguous(), coords, offs, t.batch_size, hit[1], hit[2])

    def forward(self, input):
        output = self.blocks(input)
        identity = spconv.SparseConvTensor(output.features, output.indices, output.spatial_shape, output.batch_size)
        if len(self.nPlanes) > 1 and _fusable(output, self.conv[0], self.deconv[0]):
            from .. import sparse

            down, up = self.conv[2], self.deconv[2]
            r = down.get_rules(output)
            s, t = bn_affine(self.conv[0])
            coarse = down.output_tensor(output, r)
            coarse.features = sparse.conv_fwd(output.features.contiguous(), down.weight, r.child, r.gmask_down, 8,
                                              r.M_out, r.ld, in_scale=s, in_shift=t)
            coarse = self.u(coarse)
            s, t = bn_affine(self.deconv[0])
            dec_feats = sparse.conv_fwd(coarse.features.contiguous(), up.weight, r.up, r.gmask_up, 8, r.M_in, r.ld_up,
                                        in_scale=s, in_shift=t)
            output.features = torch.cat((identity.features, dec_feats), dim=1)
            output = self.blocks_tail(output)
        elif len(self.nPlanes) > 1:
            dec = self.deconv(self.u(self.conv(output)))
            output.features = torch.cat((identity.features, dec.features), dim=1)
            output = self.blocks_tail(output)
        if self.before_transformer_linear is not None and _fusable(output) and output.features.shape[1] % 16 == 0:
            output.features = self._transformer_fused(output)
        elif self.before_transformer_linear is not None and torch.is_grad_enabled() and output.features.is_cuda and \
                _train_transformer_ok(output, self.transformer):
            # training on the GPU: forward and backward of the stack as a handful of native launches
            from .. import pointops

            output.features = pointops.backbone_transformer_train(
                output.features.contiguous(), output._coords(), output.batch_size, self.before_transformer_linear,
                self.transformer, self.after_transformer_linear)
        elif self.before_transformer_linear is not None:
            feats = self.before_transformer_linear(output.features)
            feats = self.transformer(xyz=output.indices[:, 1:].float(), features=feats, batch_ids=output.indices[:, 0],
                                     batch_size=output.batch_size)
            output.features = self.after_transformer_linear(feats)
        return output


def conv1d_bn_relu(in_channels, out_channels):
    """conv_with_kaiming_uniform("BN", activation=True) (geoformer_modules.py:132-161):
    Conv1d(k=1, no bias, kaiming_uniform a=1) + BatchNorm1d + ReLU."""
    conv = PointwiseConv1d(in_channels, out_channels, kernel_size=1, bias=False)
    nn.init.kaiming_uniform_(conv.weight, a=1)
    return nn.Sequential(conv, BatchNorm1d(out_channels), nn.ReLU(inplace=True))


def random_downsample(batch_offsets, batch_size, n_subsample=30000, host_offsets=None):
    """Host-RNG subsampling of the mask-head points (geoformer_modules.py:165-186); consumes
    np.random exactly like the reference (one np.random.choice per over-full scene).
    host_offsets: the offsets as a host list when the caller has them (the forward read the foreground counts for its
    sampling draw already): without it every scene's size is a read-back that blocks the host until the device has
    caught up -- 8 ms of the batch-4 training step, whose host side is what bounds it.  On the GPU the draw goes
    through the native restatement of numpy's legacy generator into pinned memory (a third of numpy's time, and the
    upload is an asynchronous copy)."""
    from .. import pointops

    dev = batch_offsets.device
    offs = host_offsets if host_offsets is not None else batch_offsets.tolist()
    idxs, raw = [], []
    for b in range(batch_size):
        start, n_b = int(offs[b]), int(offs[b + 1]) - int(offs[b])
        if n_subsample == -1 or n_subsample >= n_b:
            new = torch.arange(n_b, dtype=torch.long, device=dev)
        elif dev.type == "cuda":
            pin = torch.empty(n_subsample, dtype=torch.long, pin_memory=True)
            pointops.legacy_choice(n_b, n_subsample, out=pin.numpy())
            new = pin.to(dev, non_blocking=True)
        else:
            new = torch.tensor(np.random.choice(n_b, n_subsample, replace=False), dtype=torch.long, device=dev)
        raw.append(new)
        idxs.append(new + start)
    return torch.cat(idxs), raw
