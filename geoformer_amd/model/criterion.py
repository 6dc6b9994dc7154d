"""Set criterion of the training step (criterion.py:116-245, model/matcher.py:79-126), stock PyTorch + scipy.

Stays on the framework side of the boundary: semantic cross-entropy, Hungarian
matching on the host over a (class + dice) cost, then dice + focal + classification losses per decoder
layer.  Same loss definitions and weights as the reference; the matching cost is computed with one
matmul instead of the reference's [nq * n_inst, N] repeat (matcher.py:102-105), which is the same number.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from . import config as _config


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

    def forward(self, model_outputs, batch_inputs, epoch):
        cfg = self.cfg
        semantic_scores = model_outputs["semantic_scores"]
        semantic_labels, instance_labels = batch_inputs["labels"], batch_inputs["instance_labels"]
        out = {}
        if "semantic" not in cfg.fix_module:
            semantic_loss = self.semantic_criterion(semantic_scores, semantic_labels)
        else:
            semantic_loss = torch.zeros((), device=semantic_scores.device, requires_grad=True)
        loss = semantic_loss
        n = semantic_labels.shape[0]
        if epoch <= cfg.prepare_epochs or model_outputs.get("mask_predictions") is None:
            out["sem_loss"] = (semantic_loss.item(), n)
            out["loss"] = (loss.item(), n)
            return loss, out
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
