"""Few-shot set criterion (criterion_fs.py:93-280): dice + focal losses on the Hungarian-matched mask logits of
every decoder layer (matching by the dice cost alone, matcher.py:114-115), plus the hard-negative-mined BCE loss of
the similarity head (``outputs["simnet"]``, geoformer_fs.py:572).

Two routes like criterion.py: the host route (per-query Python loop with torch.mode and scipy's assignment, the
reference's arrangement: ~130 synchronisations per scene and step) and, for CUDA tensors, a device route (label counts
under every query's mask by one matmul, gf_lsap for the assignment, masked losses; two read-backs per step).  Behaviour is
pinned by tests/golden/criterion_fs.npz, produced by the reference's own FSInstSetCriterion; that includes one
quirk kept on purpose: ``loss_neg[train_label.long()] = 0`` (criterion_fs.py:176) indexes the BATCH dimension with
the 0/1 label tensor, i.e. it zeroes the whole rows 0 and 1 of the negative-loss matrix (and needs batch_size >= 2
as soon as any query is positive) instead of masking the positive entries.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import config as _config
from .criterion import (HungarianMatcher, raise_if_knn_truncated, _device_route, compute_dice_loss, compute_sigmoid_focal_loss, device_match,
                        masked_pair_losses, scene_id_ranges)


class FSInstSetCriterion(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg if cfg is not None else _config.cfg
        self.cfg = cfg
        self.similarity_criterion = nn.BCEWithLogitsLoss(reduction="none")
        self.batch_size, self.n_queries = cfg.batch_size, cfg.n_query_points
        self.matcher = HungarianMatcher(self.batch_size, self.n_queries)
        self.loss_weight = {"dice_loss": 1, "focal_loss": 1}
        self.cal_simloss = "similarity_net" not in cfg.fix_module
        self.cached = []

    def sim_loss(self, similarity_score, instance_masked, mask_logits, batch_ids):
        """Queries whose thresholded mask overlaps its dominant instance with IoU >= 0.5 are positives, IoU <= 0.3 (or
        no instance) negatives; all positives + the `negative_ratio` x hardest negatives (criterion_fs.py:117-190)."""
        cfg, dev = self.cfg, similarity_score.device
        B, nq = self.batch_size, self.n_queries
        train_label = torch.zeros((B, nq), device=dev)
        n_hard = torch.zeros(B, device=dev)
        for b in range(B):
            inst_b = instance_masked[batch_ids == b]
            pred = (mask_logits[b].detach().sigmoid() > 0.5)
            npos = nneg = 0
            pos = []
            # per-query dominant label (torch.mode: the smallest of the most frequent values), IoU with that instance
            sizes = pred.sum(1).tolist()
            for q in range(nq):
                if sizes[q] == 0:
                    nneg += 1
                    continue
                lab = int(torch.mode(inst_b[pred[q]])[0])
                if lab == -100:
                    nneg += 1
                    continue
                gt = inst_b == lab
                inter = int((pred[q] & gt).sum())
                union = int((pred[q] | gt).sum())
                iou = inter / union
                if iou >= 0.5:
                    npos += 1
                    pos.append(q)
                elif iou <= 0.3:
                    nneg += 1
            n_hard[b] = min(nneg, cfg.negative_ratio * npos)
            train_label[b, pos] = 1
        if train_label.sum() == 0:
            return torch.zeros((), device=dev, requires_grad=True)
        loss_all = self.similarity_criterion(similarity_score, train_label)
        loss_pos = loss_all * train_label
        loss_neg = loss_all.clone()
        loss_neg[train_label.long()] = 0  # the reference's indexing (see module docstring)
        loss_neg, _ = loss_neg.sort(dim=1, descending=True)
        ranks = torch.arange(nq, device=dev).unsqueeze(0).expand_as(loss_neg)
        hard = ranks < n_hard.unsqueeze(1)
        return (loss_neg[hard].sum() + loss_pos.sum()) / train_label.sum().float()

    # ---- device route (CUDA tensors): no per-query / per-instance host round trips ----------------------------
    def _sim_loss_device(self, similarity_score, instance_masked, mask_logits, counts, ranges):
        """sim_loss with the per-query loop as tensor algebra: counts of every instance id under every query's
        thresholded mask by one matmul, the dominant label by arg-max over the ids in ascending order (-100 first:
        torch.mode returns the smallest of the most frequent values), IoU with that instance in fp32 like
        torch.true_divide on the reference's integer tensors (criterion_fs.py:117-190)."""
        cfg, dev = self.cfg, similarity_score.device
        B, nq = self.batch_size, self.n_queries
        labels, hard = [], []
        s = 0
        for b in range(B):
            n_b = counts[b]
            seg = instance_masked[s:s + n_b]
            s += n_b
            lo, K = ranges[b]
            pred = mask_logits[b].detach().sigmoid() > 0.5  # [nq, n_b]
            local = torch.where(seg >= 0, seg - lo + 1, torch.zeros_like(seg))  # 0 = no instance, 1..K = ids ascending
            onehot = (local[None, :] == torch.arange(K + 1, device=dev)[:, None]).float()  # [K+1, n_b]
            C = pred.float() @ onehot.t()  # [nq, K+1], exact
            sizes = pred.sum(1).float()
            lab = C.argmax(1)  # first maximum = smallest value among the most frequent
            inter = C.gather(1, lab[:, None]).squeeze(1)
            union = sizes + onehot.sum(1)[lab] - inter
            iou = inter / union.clamp(min=1)
            negative0 = (sizes == 0) | (lab == 0)
            pos = ~negative0 & (iou >= 0.5)
            neg = negative0 | (~pos & (iou <= 0.3))
            npos, nneg = pos.sum().float(), neg.sum().float()
            hard.append(torch.minimum(nneg, cfg.negative_ratio * npos))
            labels.append(pos.float())
        train_label, n_hard = torch.stack(labels), torch.stack(hard)
        loss_all = self.similarity_criterion(similarity_score, train_label)
        loss_pos = loss_all * train_label
        loss_neg = loss_all.clone()
        loss_neg[train_label.long()] = 0  # the reference's indexing (see module docstring)
        loss_neg, _ = loss_neg.sort(dim=1, descending=True)
        ranks = torch.arange(nq, device=dev).unsqueeze(0).expand_as(loss_neg)
        hard_mask = (ranks < n_hard.unsqueeze(1)).float()
        tot = train_label.sum()
        # no positive query anywhere: both sums are 0 (n_hard = min(., ratio * 0)); the reference then returns a constant
        # 0 without a graph -- the caller drops this term after its read-back in that case (second return value)
        return ((loss_neg * hard_mask).sum() + loss_pos.sum()) / tot.clamp(min=1), tot

    def _layer_loss_device(self, mask_prediction, matches):
        dev = matches[0].match_q.device if any(m is not None for m in matches) else None
        loss_dict = None
        for b in range(self.batch_size):
            m = matches[b]
            if m is None:
                continue
            if loss_dict is None:
                loss_dict = {k: torch.zeros((), device=m.match_q.device) for k in self.loss_weight}
            dice, focal = masked_pair_losses(mask_prediction["mask_logits"][b], m, m.n_match[0].float())
            loss_dict["dice_loss"] = loss_dict["dice_loss"] + dice
            loss_dict["focal_loss"] = loss_dict["focal_loss"] + focal
        if loss_dict is None:
            z = torch.zeros(())
            return z, {k: z for k in self.loss_weight}
        loss = torch.zeros((), device=loss_dict["dice_loss"].device)
        for k, w in self.loss_weight.items():
            loss_dict[k] = loss_dict[k] * w / self.batch_size
            loss = loss + loss_dict[k]
        return loss, loss_dict

    def _all_layer_losses_device(self, preds, matches):
        """The dice + focal terms of ALL decoder layers at once (the same sums as _layer_loss_device per layer, in a
        different order): per (layer, scene) only the fused pair-loss call remains, the scalar chains around it are one
        stack and one reduction (InstSetCriterion._all_layer_losses_device)."""
        from .criterion import _PairLossFn, _fused_pair_loss

        L = len(preds)
        live = [b for b in range(self.batch_size) if matches[b] is not None]
        if not live or not _fused_pair_loss() or any(w != 1 for w in self.loss_weight.values()):
            return None
        if set(self.loss_weight) != {"dice_loss", "focal_loss"}:
            return None
        for l in range(L):
            for b in live:
                ml = preds[l]["mask_logits"][b]
                if ml is None or not ml.is_cuda or ml.dtype != torch.float32:
                    return None
        pair = torch.stack([_PairLossFn.apply(preds[l]["mask_logits"][b], matches[b]) for l in range(L) for b in live])
        terms = pair.view(L, len(live), 2).sum(1) / self.batch_size  # [L, 2]: dice, focal
        return terms.sum(1), {"dice_loss": terms[-1, 0], "focal_loss": terms[-1, 1]}

    def _forward_device(self, model_outputs, batch_inputs, epoch):
        cfg = self.cfg
        preds, fg_idxs = model_outputs["mask_predictions"], model_outputs["fg_idxs"]
        instance_labels, semantic_labels = batch_inputs["instance_labels"], batch_inputs["labels"]
        instance_masked, semantic_masked = instance_labels[fg_idxs], semantic_labels[fg_idxs]
        similarity_score = model_outputs["simnet"]
        last = preds[-1]["mask_logits"]
        if any(ml is None for ml in last):
            return None  # a scene without foreground: the host route handles the reference's skips
        counts = [int(ml.shape[1]) for ml in last]
        assert sum(counts) == instance_masked.shape[0]
        ranges = scene_id_ranges(instance_masked, counts)  # read-back 1 of 2
        if max(K for _, K in ranges) > 1024:
            return None
        dev = similarity_score.device
        loss = torch.zeros((), device=dev)
        sim, sim_tot = None, None
        if epoch > cfg.prepare_epochs and self.cal_simloss:
            sim, sim_tot = self._sim_loss_device(similarity_score, instance_masked, last, counts, ranges)
        matches, s = [], 0
        for b in range(self.batch_size):
            n_b = counts[b]
            lo, K = ranges[b]
            matches.append(None if n_b == 0 else device_match(last[b].detach(), None, instance_masked[s:s + n_b],
                                                              semantic_masked[s:s + n_b], lo, K, self.n_queries,
                                                              fewshot=True))
            s += n_b
        self.device_matches = matches
        batched = self._all_layer_losses_device(preds, matches)
        if batched is not None:
            per_layer, ld = batched
            loss = loss + per_layer.sum()
        else:
            main, ld = self._layer_loss_device(preds[-1], matches)
            loss = loss + main
            for l in range(cfg.dec_nlayers - 1):
                loss = loss + self._layer_loss_device(preds[l], matches)[0]
        live = [m for m in matches if m is not None]
        num_gt = torch.cat([m.n_match for m in live]).sum().float()
        status = torch.cat([m.status for m in live]).amax().float()
        bids = model_outputs["batch_idxs"]
        unsorted = (bids[1:] < bids[:-1]).any().float()
        zero = torch.zeros((), device=dev)
        trunc = model_outputs.get("knn_truncated")  # the forward's kNN truncation flag rides in this read-back
        vals = torch.stack([ld["focal_loss"].detach(), ld["dice_loss"].detach(), loss.detach(), num_gt, status, unsorted,
                            (sim.detach() if sim is not None else zero),
                            (sim_tot.detach() if sim is not None else zero),
                            (trunc.reshape(()).float() if trunc is not None else zero)]).tolist()  # 2 of 2
        raise_if_knn_truncated(vals[8])
        if vals[4] != 0 or vals[5] != 0:
            raise RuntimeError(f"device criterion: gf_lsap status {int(vals[4])}, batch_idxs unsorted {int(vals[5])}")
        if sim is not None and vals[7] > 0:
            loss = loss + sim  # (with no positive query the term is a constant 0: no gradient into the similarity net)
            vals[2] += vals[6]
        out = {}
        if sim is not None:
            out["sim_loss"] = (vals[6], self.n_queries)
        num_gt = int(vals[3])
        out["focal_loss"] = (vals[0], num_gt)
        out["dice_loss"] = (vals[1], num_gt)
        out["loss"] = (vals[2], semantic_labels.shape[0])
        return loss, out

    @property
    def matches_reference_format(self):
        return [(None, None, None) if m is None else m.to_reference() for m in self.device_matches]

    def single_layer_loss(self, mask_prediction, similarity_score, instance_masked, semantic_masked, batch_ids,
                          cal_match=False):
        dev = instance_masked.device
        mask_logits_list = mask_prediction["mask_logits"]
        loss_dict = {k: torch.zeros((), device=dev) for k in self.loss_weight}
        num_gt = 0
        for b in range(self.batch_size):
            mask_logit_b = mask_logits_list[b]
            if mask_logit_b is None:
                continue
            sel = batch_ids == b
            if cal_match:
                self.cached.append(self.matcher.forward_seg_single(mask_logit_b.detach(), similarity_score[b],
                                                                    instance_masked[sel], semantic_masked[sel],
                                                                    fewshot=True))
            pred_inds, inst_mask_gt, _ = self.cached[b]
            if pred_inds is None:
                continue
            n = len(pred_inds)
            num_gt += n
            if n == 0:
                continue
            pred = mask_logit_b[pred_inds]
            loss_dict["dice_loss"] = loss_dict["dice_loss"] + compute_dice_loss(pred, inst_mask_gt, n)
            loss_dict["focal_loss"] = loss_dict["focal_loss"] + compute_sigmoid_focal_loss(pred, inst_mask_gt, n)
        loss = torch.zeros((), device=dev)
        for k, w in self.loss_weight.items():
            loss_dict[k] = loss_dict[k] * w / self.batch_size
            loss = loss + loss_dict[k]
        return loss, loss_dict, num_gt

    def forward(self, model_outputs, batch_inputs, epoch):
        cfg = self.cfg
        self.device_matches = None
        if _device_route(model_outputs["simnet"]):
            res = self._forward_device(model_outputs, batch_inputs, epoch)
            if res is not None:
                return res
            self.device_matches = None
        if model_outputs.get("knn_truncated") is not None:  # (the host route synchronises per scene anyway)
            raise_if_knn_truncated(int(model_outputs["knn_truncated"].item()))
        preds, fg_idxs = model_outputs["mask_predictions"], model_outputs["fg_idxs"]
        instance_labels, semantic_labels = batch_inputs["instance_labels"], batch_inputs["labels"]
        instance_masked, semantic_masked = instance_labels[fg_idxs], semantic_labels[fg_idxs]
        batch_ids = model_outputs["batch_idxs"]
        similarity_score = model_outputs["simnet"]
        out = {}
        loss = torch.zeros((), device=similarity_score.device)
        if epoch > cfg.prepare_epochs and self.cal_simloss:
            sim = self.sim_loss(similarity_score, instance_masked, preds[-1]["mask_logits"], batch_ids)
            loss = loss + sim
            out["sim_loss"] = (sim.item(), self.n_queries)
        self.cached = []
        main, ld, num_gt = self.single_layer_loss(preds[-1], similarity_score, instance_masked, semantic_masked,
                                                  batch_ids, cal_match=True)
        loss = loss + main
        for l in range(cfg.dec_nlayers - 1):
            loss = loss + self.single_layer_loss(preds[l], similarity_score, instance_masked, semantic_masked,
                                                 batch_ids)[0]
        out["focal_loss"] = (ld["focal_loss"].item(), num_gt)
        out["dice_loss"] = (ld["dice_loss"].item(), num_gt)
        out["loss"] = (loss.item(), semantic_labels.shape[0])
        return loss, out
