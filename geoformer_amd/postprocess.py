"""Post-processing of the eval forward: matrix NMS over the proposals (util/utils_3d.py:95-141, called by
test.py:88-93 right after the forward)."""
from __future__ import annotations

import torch


def matrix_non_max_suppression(proposals_pred, scores, categories, kernel="gaussian", sigma=2.0,
                               final_score_thresh=0.05):
    """Same signature and result as the reference: indices (into the inputs) of the proposals whose decayed score
    stays >= final_score_thresh, in descending score order.  proposals_pred [n,N] 0/1 (int or float), scores [n],
    categories [n].  On the GPU the [n,n] intersection matrix comes from the bit-packed popcount kernel
    (csrc/proposal.hip) instead of a dense float einsum over N points; the rest is the reference's [n,n] algebra."""
    ixs = torch.argsort(scores, descending=True)
    n = len(ixs)
    categories_sorted = categories[ixs]
    scores_sorted = scores[ixs]
    if proposals_pred.is_cuda:
        from . import pointops

        masks = proposals_pred if proposals_pred.dtype == torch.int32 else (proposals_pred != 0).int()
        inter = pointops.mask_intersections(masks.contiguous())
        intersection = inter[ixs][:, ixs].to(scores.dtype)
    else:
        p = proposals_pred[ixs].type(scores.dtype)
        intersection = torch.einsum("nc,mc->nm", p, p)
    pointnum = torch.diagonal(intersection)
    ious = intersection / (pointnum[:, None] + pointnum[None, :] - intersection)
    cat_x = categories_sorted[None, :].expand(n, n)
    label_matrix = (cat_x == cat_x.transpose(1, 0)).float().triu(diagonal=1)
    compensate_iou, _ = (ious * label_matrix).max(0)
    compensate_iou = compensate_iou.expand(n, n).transpose(1, 0)
    decay_iou = ious * label_matrix
    if kernel == "gaussian":
        decay_matrix = torch.exp(-1 * sigma * (decay_iou ** 2))
        compensate_matrix = torch.exp(-1 * sigma * (compensate_iou ** 2))
        decay_coefficient, _ = (decay_matrix / compensate_matrix).min(0)
    elif kernel == "linear":
        decay_coefficient, _ = ((1 - decay_iou) / (1 - compensate_iou)).min(0)
    else:
        raise NotImplementedError
    return ixs[(scores_sorted * decay_coefficient) >= final_score_thresh]
