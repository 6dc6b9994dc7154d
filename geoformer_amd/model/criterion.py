"""Set criterion of the training step (criterion.py:116-245, model/matcher.py:79-126).

Semantic cross-entropy, Hungarian matching over a (class + dice) cost, then dice + focal + classification
losses per decoder layer.  Same loss definitions and weights as the reference; the matching cost is computed
with one matmul instead of the reference's [nq * n_inst, N] repeat (matcher.py:102-105), which is the same number.

Two routes to the same numbers (tests/test_criterion_golden.py pins both to the reference's own criterion):
  * host route (CPU tensors, or GF_DEVICE_CRITERION=0): per scene the ground-truth instances are listed with
    torch.unique / nonzero and the cost matrix goes to scipy.optimize.linear_sum_assignment like the reference does
    -- about ten device synchronisations per scene and step;
  * device route (CUDA tensors): instances are addressed by their id inside the scene's id
    range (one read-back per STEP gives the ranges), the assignment is solved by gf_lsap (csrc/lsap.hip: scipy's
    algorithm, float64, same tie-breaking), losses run over the fixed-size instance axis with the unmatched rows
    masked out, and every reported number leaves the device in ONE copy at the end.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from . import config as _config


def raise_if_knn_truncated(value):
    """The forward's kNN truncation flag (outputs["knn_truncated"]), read back with the step's losses."""
    if value:
        from .._lib import GeoFormerHipError

        raise GeoFormerHipError("kNN graph: a point has more in-radius neighbours than the kernel's candidate list "
                                "holds (rows truncated, geodesic distances would be wrong)")


def _device_route(t):
    import os

    return t.is_cuda and os.environ.get("GF_DEVICE_CRITERION", "1") != "0"


def _fused_pair_loss():
    import os

    return os.environ.get("GF_FUSED_PAIR_LOSS", "1") != "0"  # dev knob: 0 = the operator-by-operator formulation


class DeviceMatch:
    """Ground truth and assignment of one scene, all on the device.  The instance axis has K entries = the scene's id
    range [lo, lo + K); `present` marks the ids that occur among the scene's (sub-sampled) foreground points."""

    __slots__ = ("inst_masks", "sem_labels", "present", "match_q", "match_of_q", "n_match", "status", "lo", "cls_target")

    def to_reference(self):
        """(rows, inst_masks[cols], sem_labels[cols]) as HungarianMatcher.forward_seg_single returns them
        (model/matcher.py:124-126): rows ascending, on the host.  Synchronises; for tests and debugging."""
        if int(self.status) != 0:
            raise RuntimeError(f"gf_lsap status {int(self.status)}")
        mq = self.match_of_q.cpu().numpy()
        rows = (mq >= 0).nonzero()[0]
        cols = torch.from_numpy(mq[rows]).long().to(self.inst_masks.device)
        return rows, self.inst_masks[cols], self.sem_labels[cols]


@torch.no_grad()
def device_match(mask_logit, sem_logit, inst, sem, lo, K, n_queries, fewshot=False):
    """Matching of one scene without leaving the device.  inst / sem: instance ids (-100 = none) and semantic labels of
    the scene's points; [lo, lo + K) the scene's id range (host integers)."""
    from .. import _lib
    from .._lib import check, ptr, stream_ptr

    dev = mask_logit.device
    m = DeviceMatch()
    m.lo = lo
    n_mask = inst.shape[0]
    local = inst - lo
    ids = torch.arange(K, device=dev)
    m.inst_masks = (local[None, :] == ids[:, None]).float()  # [K, n_mask]; ids outside the range (-100) match nothing
    counts = m.inst_masks.sum(-1)
    m.present = (counts > 0).int()
    # semantic label of the instance's first point (matcher.py:96)
    first = torch.where(m.inst_masks > 0, torch.arange(n_mask, device=dev)[None, :], n_mask).amin(1).clamp(max=n_mask - 1)
    m.sem_labels = sem[first].float()
    prob = mask_logit.sigmoid()
    numerator = 2 * prob @ m.inst_masks.t()
    denominator = prob.sum(-1)[:, None] + counts[None, :]
    cost = 1 - (numerator + 1) / (denominator + 1)
    if not fewshot:
        cls = torch.softmax(sem_logit, dim=-1)
        cost = cost - torch.gather(cls, 1, m.sem_labels[None].expand(n_queries, K).long().clamp(min=0))
    cost = cost.contiguous()
    m.match_q = torch.empty(K, dtype=torch.int32, device=dev)
    m.match_of_q = torch.empty(n_queries, dtype=torch.int32, device=dev)
    m.n_match = torch.empty(1, dtype=torch.int32, device=dev)
    m.status = torch.empty(1, dtype=torch.int32, device=dev)
    check(_lib.load().gf_lsap(ptr(cost), n_queries, K, ptr(m.present), ptr(m.match_q), ptr(m.match_of_q), ptr(m.n_match),
                              ptr(m.status), stream_ptr()), "gf_lsap")
    return m


def scene_id_ranges(instance_masked, counts):
    """Per scene (lo, K) of the instance ids among its points -- the one read-back of the device route.  counts: points
    per scene (host integers, scenes contiguous)."""
    big = torch.iinfo(instance_masked.dtype).max
    stats, s = [], 0
    for n in counts:
        seg = instance_masked[s:s + n]
        s += n
        if n == 0:
            stats.append(instance_masked.new_tensor([0, -1]))
            continue
        valid = seg >= 0
        stats.append(torch.stack([torch.where(valid, seg, big).amin(), torch.where(valid, seg, -1).amax()]))
    out = []
    for lo, hi in torch.stack(stats).tolist():
        out.append((int(lo), int(hi - lo + 1)) if hi >= 0 else (0, 1))  # no instance at all: one absent slot
    return out


class _PairLossFn(torch.autograd.Function):
    """(dice, focal) of one scene's matched pairs through gf_pair_losses_fwd / _bwd (csrc/pair_losses.hip): three
    launches instead of the ~75 of the operator-by-operator formulation below."""

    @staticmethod
    def forward(ctx, mask_logit_b, m):
        from .. import _lib
        from .._lib import check, ptr, stream_ptr

        x = mask_logit_b.contiguous()
        nq, n = x.shape
        K = m.inst_masks.shape[0]
        lib = _lib.load()
        sums = torch.empty(max(int(lib.gf_pair_losses_sums_floats(K)), 4), dtype=torch.float32, device=x.device)
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        check(_lib.load().gf_pair_losses_fwd(ptr(x), ptr(m.inst_masks), ptr(m.match_q), nq, K, n, ptr(m.n_match), ptr(sums),
                                             ptr(out), stream_ptr()), "gf_pair_losses_fwd")
        ctx.save_for_backward(x, sums)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from .. import _lib
        from .._lib import check, ptr, stream_ptr

        x, sums = ctx.saved_tensors
        m = ctx.m
        nq, n = x.shape
        g = grad_out.contiguous().float()
        dx = torch.empty_like(x)
        check(_lib.load().gf_pair_losses_bwd(ptr(x), ptr(m.inst_masks), ptr(m.match_of_q), ptr(sums), nq,
                                             m.inst_masks.shape[0], n, ptr(m.n_match), ptr(g), ptr(dx), stream_ptr()),
              "gf_pair_losses_bwd")
        return dx, None


def masked_pair_losses(mask_logit_b, m, n):
    """dice and focal loss sums over the matched (query, instance) pairs of one scene, unmatched instance rows masked
    out; n = number of pairs as a device scalar (compute_dice_loss / compute_sigmoid_focal_loss on the matched rows)."""
    if (mask_logit_b.is_cuda and mask_logit_b.dtype == torch.float32 and m.inst_masks.is_contiguous()
            and _fused_pair_loss()):
        out = _PairLossFn.apply(mask_logit_b, m)
        return out[0], out[1]
    valid = (m.match_q >= 0).float()
    pred = mask_logit_b[m.match_q.clamp(min=0).long()]  # [K, n_mask]
    tgt = m.inst_masks
    p = pred.sigmoid()
    dice = 1 - (2 * (p * tgt).sum(1) + 1) / (p.sum(-1) + tgt.sum(-1) + 1)
    ce = F.binary_cross_entropy_with_logits(pred, tgt, reduction="none")
    p_t = p * tgt + (1 - p) * (1 - tgt)
    focal = ((0.25 * tgt + 0.75 * (1 - tgt)) * ce * ((1 - p_t) ** 2)).mean(1)
    return (dice * valid).sum() / (n + 1e-6), (focal * valid).sum() / (n + 1e-6)


def compute_dice_loss(inputs, targets, num_boxes):
    inputs = inputs.sigmoid()
    numerator = 2 * (inputs * targets).sum(1)
    denominator = inputs.sum(-1) + targets.sum(-1)
    return (1 - (numerator + 1) / (denominator + 1)).sum() / (num_boxes + 1e-6)


def compute_sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2):
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / (num_boxes + 1e-6)


class HungarianMatcher(nn.Module):
    def __init__(self, batch_size, n_queries):
        super().__init__()
        self.batch_size, self.n_queries = batch_size, n_queries

    @torch.no_grad()
    def forward_seg_single(self, mask_logit, sem_logit, instance_masked, semantic_masked, fewshot=False):
        n_mask = instance_masked.shape[-1]
        if n_mask == 0:
            return None, None, None
        ids = [i for i in torch.unique(instance_masked).tolist() if i != -100]
        inst_masks = torch.stack([(instance_masked == i) for i in ids]).float() if ids else \
            torch.zeros((0, n_mask), device=mask_logit.device)
        first = [int(torch.nonzero(instance_masked == i)[0]) for i in ids]
        sem_labels = semantic_masked[first].float() if ids else torch.zeros(0, device=mask_logit.device)
        prob = mask_logit.sigmoid()
        numerator = 2 * prob @ inst_masks.t()
        denominator = prob.sum(-1)[:, None] + inst_masks.sum(-1)[None, :]
        cost = 1 - (numerator + 1) / (denominator + 1)  # dice cost, [nq, n_inst]
        if not fewshot:
            cls = torch.softmax(sem_logit, dim=-1)
            cost = cost - torch.gather(cls, 1, sem_labels[None].expand(self.n_queries, len(ids)).long())
        rows, cols = linear_sum_assignment(cost.cpu().numpy())
        return rows, inst_masks[cols], sem_labels[cols]


def semantic_cross_entropy(scores, labels, ignore_label, module):
    """nn.CrossEntropyLoss(ignore_index) over all points of the batch (criterion.py:120).  On the GPU the mean is taken
    outside the library's loss: its 'mean' reduction runs on ONE workgroup (0.57 ms forward + 0.52 ms backward over the
    550k points of a training batch); per-point losses (an element-wise kernel, zeros at ignored points) summed by a
    parallel reduction and divided by the number of counted points are the same number."""
    if not scores.is_cuda:
        return module(scores, labels)
    per_point = F.cross_entropy(scores, labels, ignore_index=ignore_label, reduction="none")
    return per_point.sum() / (labels != ignore_label).sum()


class InstSetCriterion(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg if cfg is not None else _config.cfg
        self.cfg = cfg
        self.semantic_criterion = nn.CrossEntropyLoss(ignore_index=cfg.ignore_label)
        self.batch_size, self.n_queries = cfg.batch_size, cfg.n_query_points
        self.matcher = HungarianMatcher(self.batch_size, self.n_queries)
        self.loss_weight = {"dice_loss": 1, "focal_loss": 1, "cls_loss": 1}
        self.cached = []

    def single_layer_loss(self, mask_prediction, instance_masked, semantic_masked, batch_ids, cal_match=False):
        dev = instance_masked.device
        mask_logits_list, cls_logits = mask_prediction["mask_logits"], mask_prediction["cls_logits"]
        loss_dict = {k: torch.zeros((), device=dev) for k in self.loss_weight}
        num_gt = 0
        for b in range(self.batch_size):  # loops cfg.batch_size like the reference (criterion.py:148)
            mask_logit_b = mask_logits_list[b]
            if mask_logit_b is None:
                if cal_match:
                    self.cached.append((None, None, None))
                continue
            cls_logit_b = cls_logits[b]
            sel = batch_ids == b
            if cal_match:
                m = self.matcher.forward_seg_single(mask_logit_b.detach(), cls_logit_b.detach(), instance_masked[sel],
                                                    semantic_masked[sel])
                self.cached.append(m)
            pred_inds, inst_mask_gt, sem_cls_gt = self.cached[b]
            if pred_inds is None:
                continue
            pred = mask_logit_b[pred_inds]
            n = len(pred_inds)
            num_gt += n
            loss_dict["dice_loss"] = loss_dict["dice_loss"] + compute_dice_loss(pred, inst_mask_gt, n)
            loss_dict["focal_loss"] = loss_dict["focal_loss"] + compute_sigmoid_focal_loss(pred, inst_mask_gt, n)
            cls_label = torch.zeros(self.n_queries, device=dev)
            cls_label[pred_inds] = sem_cls_gt
            loss_dict["cls_loss"] = loss_dict["cls_loss"] + F.cross_entropy(cls_logit_b, cls_label.long())
        loss = torch.zeros((), device=dev)
        for k, w in self.loss_weight.items():
            loss_dict[k] = loss_dict[k] * w / self.batch_size
            loss = loss + loss_dict[k]
        return loss, loss_dict, num_gt

    def _layer_loss_device(self, mask_prediction, matches):
        """single_layer_loss over the device matches (None entries: scenes the reference skips)."""
        dev = mask_prediction["cls_logits"].device
        mask_logits_list, cls_logits = mask_prediction["mask_logits"], mask_prediction["cls_logits"]
        loss_dict = {k: torch.zeros((), device=dev) for k in self.loss_weight}
        for b in range(self.batch_size):
            m = matches[b]
            if m is None:
                continue
            n = m.n_match[0].float()
            dice, focal = masked_pair_losses(mask_logits_list[b], m, n)
            loss_dict["dice_loss"] = loss_dict["dice_loss"] + dice
            loss_dict["focal_loss"] = loss_dict["focal_loss"] + focal
            # class targets: 0 for unmatched queries, the instance's class for matched ones (criterion.py:170-173);
            # the matching is the last layer's for every layer, so they are built once per scene
            tgt = getattr(m, "cls_target", None)
            if tgt is None:
                valid = m.match_q >= 0
                cls_label = torch.zeros(self.n_queries + 1, device=dev)
                cls_label.scatter_(0, torch.where(valid, m.match_q, self.n_queries).long(), m.sem_labels)
                tgt = m.cls_target = cls_label[:-1].long()
            loss_dict["cls_loss"] = loss_dict["cls_loss"] + F.cross_entropy(cls_logits[b], tgt)
        loss = torch.zeros((), device=dev)
        for k, w in self.loss_weight.items():
            loss_dict[k] = loss_dict[k] * w / self.batch_size
            loss = loss + loss_dict[k]
        return loss, loss_dict

    def _all_layer_losses_device(self, preds, matches):
        """The losses of ALL decoder layers over the device matches at once: the same terms as _layer_loss_device per
        layer (criterion.py:137-196: dice + focal over the matched pairs, class cross-entropy over the queries, each
        summed over the scenes and divided by the batch size), collected with a handful of launches -- per (layer, scene)
        only the fused pair-loss call remains; the scalar chains of the per-layer formulation (~45 small launches per
        layer forward and as many backward) bounded this part of the step on the host.  Returns ([L] losses, the last
        layer's terms)."""
        L, B = len(preds), self.batch_size
        live = [b for b in range(B) if matches[b] is not None]
        if not live or any(w != 1 for w in self.loss_weight.values()):
            return None
        if any(preds[l]["mask_logits"][b].dtype != torch.float32 for l in range(L) for b in live):
            return None
        dev = preds[-1]["cls_logits"].device
        pair = torch.stack([_PairLossFn.apply(preds[l]["mask_logits"][b], matches[b]) for l in range(L) for b in live])
        pair = pair.view(L, len(live), 2).sum(1)  # [L, 2]: dice, focal summed over the scenes
        tgts = []
        for b in range(B):
            m = matches[b]
            if m is None:
                tgts.append(torch.zeros(self.n_queries, dtype=torch.long, device=dev))
                continue
            tgt = getattr(m, "cls_target", None)
            if tgt is None:
                valid = m.match_q >= 0
                cls_label = torch.zeros(self.n_queries + 1, device=dev)
                cls_label.scatter_(0, torch.where(valid, m.match_q, self.n_queries).long(), m.sem_labels)
                tgt = m.cls_target = cls_label[:-1].long()
            tgts.append(tgt)
        tgt = torch.stack(tgts)  # [B, nq]
        logits = torch.stack([p["cls_logits"] for p in preds])  # [L, B, nq, C]
        C = logits.shape[-1]
        ce = F.cross_entropy(logits.reshape(-1, C), tgt.unsqueeze(0).expand(L, B, self.n_queries).reshape(-1),
                             reduction="none").view(L, B, self.n_queries).mean(2)  # [L, B]
        if len(live) < B:
            keep = torch.zeros(B, device=dev)
            keep[live] = 1
            ce = ce * keep
        terms = torch.cat([pair, ce.sum(1, keepdim=True)], 1) / B  # [L, 3]: dice, focal, class
        ld = {"dice_loss": terms[-1, 0], "focal_loss": terms[-1, 1], "cls_loss": terms[-1, 2]}
        return terms.sum(1), ld

    def _forward_device(self, model_outputs, semantic_loss, semantic_labels, instance_labels):
        cfg = self.cfg
        preds, fg_idxs = model_outputs["mask_predictions"], model_outputs["fg_idxs"]
        instance_masked, semantic_masked = instance_labels[fg_idxs], semantic_labels[fg_idxs]
        last = preds[-1]["mask_logits"]
        counts = [0 if last[b] is None else int(last[b].shape[1]) for b in range(self.batch_size)]
        assert sum(counts) == instance_masked.shape[0], "scenes must be contiguous in fg_idxs / batch_idxs"
        ranges = scene_id_ranges(instance_masked, counts)  # read-back 1 of 2
        if max(K for _, K in ranges) > 1024:
            return None  # instance ids spread over a wide range (not the collate's consecutive ids): host route
        matches, s = [], 0
        for b in range(self.batch_size):
            n_b = counts[b]
            if last[b] is None or n_b == 0:
                matches.append(None)
            else:
                lo, K = ranges[b]
                matches.append(device_match(last[b].detach(), preds[-1]["cls_logits"][b].detach(),
                                            instance_masked[s:s + n_b], semantic_masked[s:s + n_b], lo, K, self.n_queries))
            s += n_b
        self.device_matches = matches
        batched = self._all_layer_losses_device(preds, matches) if _fused_pair_loss() else None
        if batched is not None:
            per_layer, ld = batched
            loss = semantic_loss + per_layer.sum()
        else:
            main, ld = self._layer_loss_device(preds[-1], matches)
            loss = semantic_loss + main
            for l in range(cfg.dec_nlayers - 1):  # auxiliary losses reuse the matching of the last layer
                loss = loss + self._layer_loss_device(preds[l], matches)[0]
        live = [m for m in matches if m is not None]
        zero = torch.zeros(1, dtype=torch.int32, device=loss.device)
        num_gt = torch.cat([m.n_match for m in live]).sum() if live else zero.sum()
        status = torch.cat([m.status for m in live]).amax() if live else zero.sum()
        # the per-scene slices above rely on the model's layout: points of a scene contiguous, scenes in order
        bids = model_outputs["batch_idxs"]
        unsorted = (bids[1:] < bids[:-1]).any().float() if bids.numel() > 1 else zero.sum().float()
        trunc = model_outputs.get("knn_truncated")  # the forward's kNN truncation flag rides in this read-back
        trunc = trunc.reshape(()).float() if trunc is not None else zero.sum().float()
        vals = torch.stack([ld["focal_loss"].detach(), ld["dice_loss"].detach(), ld["cls_loss"].detach(),
                            semantic_loss.detach(), loss.detach(), num_gt.float(), status.float(), unsorted,
                            trunc]).tolist()  # 2 of 2
        raise_if_knn_truncated(vals[8])
        if vals[6] != 0:
            raise RuntimeError(f"gf_lsap status {int(vals[6])} (1: more than 512 x 1024 queries x instances, "
                               "2: non-finite costs)")
        if vals[7] != 0:
            raise RuntimeError("device criterion: batch_idxs is not sorted by scene (set GF_DEVICE_CRITERION=0 for "
                               "the host route, which selects every scene's points with a mask)")
        n, num_gt = semantic_labels.shape[0], int(vals[5])
        out = {"focal_loss": (vals[0], num_gt), "dice_loss": (vals[1], num_gt), "cls_loss": (vals[2], self.n_queries),
               "sem_loss": (vals[3], n), "loss": (vals[4], n)}
        return loss, out

    @property
    def matches_reference_format(self):
        """The last device matching in the host route's `cached` format (synchronises)."""
        return [(None, None, None) if m is None else m.to_reference() for m in self.device_matches]

    def forward(self, model_outputs, batch_inputs, epoch):
        cfg = self.cfg
        semantic_scores = model_outputs["semantic_scores"]
        semantic_labels, instance_labels = batch_inputs["labels"], batch_inputs["instance_labels"]
        out = {}
        if "semantic" not in cfg.fix_module:
            semantic_loss = semantic_cross_entropy(semantic_scores, semantic_labels, cfg.ignore_label,
                                                   self.semantic_criterion)
        else:
            semantic_loss = torch.zeros((), device=semantic_scores.device, requires_grad=True)
        loss = semantic_loss
        n = semantic_labels.shape[0]
        if epoch <= cfg.prepare_epochs or model_outputs.get("mask_predictions") is None:
            out["sem_loss"] = (semantic_loss.item(), n)
            out["loss"] = (loss.item(), n)
            return loss, out
        if _device_route(semantic_scores):
            res = self._forward_device(model_outputs, semantic_loss, semantic_labels, instance_labels)
            if res is not None:
                return res
            self.device_matches = None
        if model_outputs.get("knn_truncated") is not None:  # (the host route synchronises per scene anyway)
            raise_if_knn_truncated(int(model_outputs["knn_truncated"].item()))
        preds, fg_idxs = model_outputs["mask_predictions"], model_outputs["fg_idxs"]
        instance_masked, semantic_masked = instance_labels[fg_idxs], semantic_labels[fg_idxs]
        batch_ids = model_outputs["batch_idxs"]
        self.cached = []
        main, ld, num_gt = self.single_layer_loss(preds[-1], instance_masked, semantic_masked, batch_ids, cal_match=True)
        loss = loss + main
        for l in range(cfg.dec_nlayers - 1):  # auxiliary losses reuse the matching of the last layer
            loss = loss + self.single_layer_loss(preds[l], instance_masked, semantic_masked, batch_ids)[0]
        out["focal_loss"] = (ld["focal_loss"].item(), num_gt)
        out["dice_loss"] = (ld["dice_loss"].item(), num_gt)
        out["cls_loss"] = (ld["cls_loss"].item(), self.n_queries)
        out["sem_loss"] = (semantic_loss.item(), n)
        out["loss"] = (loss.item(), n)
        return loss, out
