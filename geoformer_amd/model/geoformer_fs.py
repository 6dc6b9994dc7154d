"""GeoFormerFS (few-shot) on MI355X: counterpart of ``model/geoformer/geoformer_fs.py:21-793``.

Same pipeline as GeoFormer plus the support branch: ``process_support`` (backbone under no_grad ->
masked points -> FPS to 32 -> SA-MLP with average pooling -> mean embedding, :377-422), the
support (x) query fusion ``cat(ctx*s, ctx-s, ctx)`` (:532-538), ``similarity_net`` (:151-159,572) and the
result cache ``remember`` (:439-455,508-523).  Differences from GeoFormer kept on purpose: ``pc_dims`` is
[mins, maxs] (:434-437), the aggregator does not subsample (:619-660), there is no per-query class head,
proposals are scored by mask score * sqrt(similarity) (:191-239).  The reference evaluates the decoder
and mask head under ``autocast`` (fp16 on a GPU); this build evaluates them in fp32 (its fused kernels are
fp32), i.e. with strictly more precision.
"""
from __future__ import annotations

import functools

import os

import torch
import torch.nn as nn

from . import config as _config
from .backbone import random_downsample
from .criterion import raise_if_knn_truncated
from .. import pointops
from .geoformer import GeoFormer, _offsets_list, cal_geodesic, get_batch_offsets, knn_graphs, knn_truncated
from .layers import BatchNorm1d, GenericMLP


class GeoFormerFS(GeoFormer):
    def __init__(self, cfg=None):
        cfg = cfg if cfg is not None else _config.cfg
        super().__init__(cfg)
        m = cfg.m
        agg = 2 * m
        norm_fn = functools.partial(BatchNorm1d, eps=1e-4, momentum=0.1)
        del self.detr_sem_head
        self.encoder_to_decoder_projection = GenericMLP(
            input_dim=agg * 3, hidden_dims=[agg * 3], output_dim=cfg.dec_dim, norm_fn_name="bn1d", activation="relu",
            use_conv=True, output_use_activation=True, output_use_norm=True, output_use_bias=False)
        self.similarity_net = nn.Sequential(
            nn.Linear(3 * agg, 3 * agg, bias=True), norm_fn(3 * agg), nn.ReLU(),
            nn.Linear(3 * agg, 3 * agg, bias=True), norm_fn(3 * agg), nn.ReLU(),
            nn.Linear(3 * agg, 1, bias=True))
        self.apply(self.set_bn_init)
        for name in self.fix_module:
            for p in getattr(self, name).parameters():
                p.requires_grad = False
        self.cache_data = None

    # -- support branch ---------------------------------------------------------------------------
    def process_support(self, batch_input, training=True):
        batch_idxs = batch_input["locs"][:, 0].int()
        locs_float = batch_input["locs_float"]
        batch_size = len(batch_input["batch_offsets"]) - 1
        assert batch_size > 0
        with torch.no_grad():
            x = self.unet_features(self.preprocess_input(batch_input, batch_size), batch_size)
            output_feats = x.features[batch_input["p2v_map"].long()].contiguous()
            mask_indices = torch.nonzero(batch_input["support_masks"] == 1).view(-1)
            feats_, locs_ = output_feats[mask_indices], locs_float[mask_indices]
            offs = get_batch_offsets(batch_idxs[mask_indices], batch_size).tolist()
            embs = []
            for b in range(batch_size):
                xyz_b = locs_[offs[b]:offs[b + 1]].unsqueeze(0)
                f_b = feats_[offs[b]:offs[b + 1]].unsqueeze(0)
                _, gfeat, gxyz, _ = self.set_aggregator.group_points(xyz_b.contiguous(),
                                                                      f_b.transpose(1, 2).contiguous(), npoint_new=32)
                ctx = self.set_aggregator.mlp(gfeat, gxyz, pooling="avg").transpose(1, 2)  # 1 x 32 x C
                embs.append(torch.mean(ctx, dim=1))
            return torch.cat(embs)  # batch x channel

    # -- overrides ----------------------------------------------------------------------------------
    def forward_backbone(self, batch_input, batch_size):
        ctx = torch.enable_grad if self.training and "unet" not in self.fix_module else torch.no_grad
        with ctx():
            x = self.unet_features(self.preprocess_input(batch_input, batch_size), batch_size)
            output_feats = x.features[batch_input["p2v_map"].long()].contiguous()
            chain = self._pointwise_chain("semantic", [self.semantic, self.semantic_linear], output_feats)
            if chain is not None:
                semantic_scores = pointops.pointwise_mlp(output_feats, chain)
            else:
                semantic_scores = self.semantic_linear(self.semantic(output_feats))
            return output_feats, semantic_scores, semantic_scores.max(1)[1]

    def forward_aggregator(self, locs_float_, output_feats_, batch_offsets_, batch_size):
        ctx = torch.enable_grad if self.training and "set_aggregator" not in self.fix_module else torch.no_grad
        offs = _offsets_list(batch_offsets_)
        with ctx():
            locs, gfeat, gxyz, inds = [], [], [], []
            for b in range(batch_size):
                if offs[b + 1] - offs[b] == 0:
                    return None
                xyz_b = locs_float_[offs[b]:offs[b + 1]].unsqueeze(0)
                f_b = output_feats_[offs[b]:offs[b + 1]].unsqueeze(0)
                l, gf, gx, idx = self.set_aggregator.group_points(xyz_b.contiguous(), f_b.transpose(1, 2).contiguous())
                locs.append(l); gfeat.append(gf); gxyz.append(gx); inds.append(idx)
            context_feats = self.set_aggregator.mlp(torch.cat(gfeat), torch.cat(gxyz)).transpose(1, 2)
            return torch.cat(locs), context_feats, torch.cat(inds)

    def get_mask_prediction(self, geo_dists, param_kernels, mask_features, locs_float_, fps_sampling_locs,
                            batch_offsets_):
        num_layers, n_queries, batch = param_kernels.shape[:3]
        offs = _offsets_list(batch_offsets_)
        outputs = []
        # training on the GPU: the fused mask head forward AND backward (csrc/mask_head.hip) like GeoFormer's training
        # route, instead of autograd over [nq, 16, N] batched GEMMs (16 x (236 + 3 x 168) us per episode step)
        fused_train = (mask_features.is_cuda and torch.is_grad_enabled() and self.output_dim == 16 and self.use_coords
                       and os.environ.get("GF_FUSED_BWD", "1") != "0")
        per_scene = {}
        if fused_train:
            for b in range(batch):
                s, e = offs[b], offs[b + 1]
                if e - s == 0:
                    continue
                g = geo_dists[b].contiguous()
                mx = torch.max(g, dim=1)[0]
                mx = torch.sqrt(torch.where(mx < 0, torch.max(mx), mx)).contiguous()
                per_scene[b] = (mask_features[s:e].reshape(e - s, self.output_dim).contiguous(),
                                locs_float_[s:e].contiguous(), g, fps_sampling_locs[b].reshape(-1, 3).contiguous(), mx)
        for l in range(num_layers):
            pk2 = param_kernels[l].transpose(0, 1).flatten(0, 1)
            controllers = self.controller(self.before_embedding_tower(pk2.unsqueeze(2))).squeeze(2)
            controllers = controllers.reshape(batch, n_queries, -1)
            mask_logits_list = []
            for b in range(batch):
                s, e = offs[b], offs[b + 1]
                if e - s == 0:
                    mask_logits_list.append(None)
                    continue
                if fused_train:
                    mf_b, locs_b, g, fps_b, mx = per_scene[b]
                    mask_logits_list.append(pointops.mask_head_train(mf_b, controllers[b].contiguous(), locs_b, g, fps_b, mx))
                    continue
                weights, biases = self.parse_dynamic_params(controllers[b], self.output_dim)
                ml = self.mask_heads_forward(geo_dists[b], mask_features[s:e], weights, biases, n_queries,
                                             locs_float_[s:e], fps_sampling_locs[b], use_geo=True)
                mask_logits_list.append(ml.float().squeeze(0))
            outputs.append({"mask_logits": mask_logits_list})
        return outputs

    def generate_proposal(self, mask_logits, similarity_score, fg_idxs, batch_offsets, logit_thresh=0.5,
                          score_thresh=0.5, npoint_thresh=100, sim_score_thresh=0.5):
        b = 0
        num_points = int(batch_offsets[b + 1] - batch_offsets[b])
        if mask_logits[b].is_cuda:
            # one sweep over the logit rows (gf_proposal_stats_fs), ONE read-back (acceptance flags + the scene's kNN
            # truncation flag), then the membership rows written by the native scatter straight from the logits
            ml = mask_logits[b].float().contiguous()
            _, scores, final = pointops.proposal_stats_fs(ml, similarity_score[b].float().contiguous(), logit_thresh,
                                                          score_thresh, npoint_thresh, sim_score_thresh)
            flag = getattr(self, "_knn_flag", None)
            host = (torch.cat([final, flag.reshape(1).int()]) if flag is not None else final).cpu()
            if flag is not None:
                raise_if_knn_truncated(int(host[-1]))
            return self._cut_proposals(ml, scores, host[:final.shape[0]], fg_idxs, logit_thresh, num_points)
        prob = mask_logits[b].sigmoid()
        sim = similarity_score[b]
        mask_bool = prob >= logit_thresh
        npts = torch.sum(mask_bool, dim=1)
        mask_scores = torch.sum(prob * mask_bool.int(), dim=1) / (npts + 1e-6)
        scores = mask_scores * torch.pow(sim, 0.5)
        final = (sim >= sim_score_thresh) & (npts >= npoint_thresh) & (mask_scores >= score_thresh)
        flag = getattr(self, "_knn_flag", None)
        if flag is not None:  # the scene's kNN truncation flag rides in this read-back
            n_final, trunc = torch.stack([torch.count_nonzero(final), flag.reshape(()).long()]).tolist()
            raise_if_knn_truncated(trunc)
        else:
            n_final = int(torch.count_nonzero(final))
        if n_final == 0:
            return [], []
        masks_final = mask_bool[final]
        proposals = torch.zeros((masks_final.shape[0], num_points), dtype=torch.int, device=prob.device)
        inst, pts = torch.nonzero(masks_final, as_tuple=True)
        proposals[inst, fg_idxs[pts]] = 1
        return scores[final], proposals

    @staticmethod
    def _cut_proposals(logits, scores, final_host, fg_idxs, logit_thresh, num_points):
        """(scores, 0/1 membership rows) of the accepted queries: `final_host` is the acceptance vector ON THE HOST (the
        caller's one read-back), the rows come from gf_proposal_scatter (sigmoid(logit) >= logit_thresh over the scene's
        points, geoformer_fs.py:229-238)."""
        idx = torch.nonzero(final_host).view(-1)
        if idx.numel() == 0:
            return [], []
        sel = idx.to(dtype=torch.int32).to(logits.device, non_blocking=True)
        proposals = pointops.proposal_scatter(logits, sel, fg_idxs.contiguous(), logit_thresh, num_points)
        return scores[sel.long()], proposals

    # -- batched re-query of a cached scene ---------------------------------
    REQUERY_CHUNK = 16  # episodes per decoder pass (the pair products of a pass are E x 2048 x 64 floats per layer)

    @torch.no_grad()
    def requery_many(self, scene_dict, support_embeddings):
        """The few-shot test loop re-queries ONE scene with many support embeddings (every label of the scene x
        ``run_num`` support draws, test_fs.py:157-174): ``forward(..., remember=True, support_embeddings=e)`` once per
        embedding, each ending in the host read-backs of generate_proposal (count_nonzero, boolean indexing).
        Here the E re-queries are ONE decoder pass with the episode as the batch index: the fused context
        [E, nc, 3C], the projections, the token stages and the cross-attention kernels run once over E "scenes" that
        share the geodesic distances, context / query positions and mask features of the cached scene.  The mask head
        has an episode dimension as well (gf_mask_head_episodes: the chunk's [E, nq, N] logits in one launch,
        gf_proposal_stats_fs over all E * nq rows in another).
        The host synchronises ONCE at the end to cut the accepted proposals.
        ``support_embeddings``: [E, C] (or a list of [1, C]); the scene must have gone through
        ``forward(..., remember=False)`` before (``cache_data``).  Returns a list of E ``(scores, proposals)`` pairs
        as the sequential calls return them (``([], [])`` where nothing is accepted)."""
        cfg = self.cfg
        assert self.cache_data is not None, "requery_many: run forward(..., remember=False) on the scene first"
        (context_locs, context_feats, pre_enc_inds, fg_idxs, batch_offsets, output_feats_, batch_idxs_, locs_float_,
         batch_offsets_, semantic_preds_, semantic_scores, query_locs, mask_features_, geo_dists) = self.cache_data
        if torch.is_tensor(support_embeddings):
            embs = support_embeddings
        else:
            embs = torch.cat(list(support_embeddings))
        n_emb = embs.shape[0]
        if len(fg_idxs) == 0:
            return [None] * n_emb
        assert context_locs.shape[0] == 1, "requery_many: one cached scene"
        nq, nc = cfg.n_query_points, cfg.n_decode_point
        num_points = int(batch_offsets[1] - batch_offsets[0])
        pending = []
        mx = None  # sqrt of the queries' largest geodesic distance (mask head), once for all episodes
        for c0 in range(0, n_emb, self.REQUERY_CHUNK):
            e = embs[c0:c0 + self.REQUERY_CHUNK]
            E = e.shape[0]
            s = e.unsqueeze(1)  # [E, 1, C]
            ctx = context_feats.expand(E, -1, -1)
            aggregation = torch.cat([ctx * s, ctx - s, ctx], dim=2)  # [E, nc, 3C]
            pc_dims = [scene_dict["pc_mins"].expand(E, -1), scene_dict["pc_maxs"].expand(E, -1)]
            dec_outputs = self.forward_decoder(context_locs.expand(E, -1, -1), aggregation, query_locs.expand(E, -1, -1),
                                               pc_dims, [geo_dists[0]] * E, pre_enc_inds.expand(E, -1))[-1]  # [nq, E, d]
            pk2 = dec_outputs.transpose(0, 1).flatten(0, 1)  # [E * nq, d] token rows
            controllers = self.controller(self.before_embedding_tower(pk2.unsqueeze(2))).squeeze(2).reshape(E, nq, -1)
            sim_all = self.similarity_net(aggregation[:, :nq, :].flatten(0, 1)).squeeze(-1).reshape(E, nq).sigmoid()
            if controllers.is_cuda and self.output_dim == 16:
                # the mask head with an episode dimension: ONE launch writes the [E, nq, N] logits of the chunk (the
                # generated parameters read in place from the controller's output), ONE launch their proposal statistics
                if mx is None:
                    mx = torch.max(geo_dists[0], dim=1)[0]
                    mx = torch.sqrt(torch.where(mx < 0, torch.max(mx), mx)).contiguous()
                ml_all = pointops.mask_head_episodes(
                    mask_features_.reshape(-1, self.output_dim).contiguous(), locs_float_.contiguous(),
                    geo_dists[0].contiguous(), query_locs[0].reshape(-1, 3).contiguous(), mx, controllers.float().contiguous())
                _, scores_all, final_all = pointops.proposal_stats_fs(
                    ml_all.view(E * nq, -1), sim_all.float().reshape(-1).contiguous(), 0.2, cfg.TEST_SCORE_THRESH,
                    cfg.TEST_NPOINT_THRESH, cfg.similarity_thresh)
                for i in range(E):
                    pending.append((scores_all[i * nq:(i + 1) * nq], final_all[i * nq:(i + 1) * nq], ml_all[i]))
                continue
            for i in range(E):
                weights, biases = self.parse_dynamic_params(controllers[i], self.output_dim)
                ml = self.mask_heads_forward(geo_dists[0], mask_features_, weights, biases, nq, locs_float_,
                                             query_locs[0], use_geo=True).float().squeeze(0)
                # generate_proposal (geoformer_fs.py:191-239) up to the point where it needs the host
                prob = ml.sigmoid()
                mask_bool = prob >= 0.2
                npts = torch.sum(mask_bool, dim=1)
                mask_scores = torch.sum(prob * mask_bool.int(), dim=1) / (npts + 1e-6)
                sim = sim_all[i]
                scores = mask_scores * torch.pow(sim, 0.5)
                final = (sim >= cfg.similarity_thresh) & (npts >= cfg.TEST_NPOINT_THRESH) & \
                    (mask_scores >= cfg.TEST_SCORE_THRESH)
                pending.append((scores, final, mask_bool))
        flag = getattr(self, "_knn_flag", None)
        if flag is not None:  # the scene's kNN truncation flag rides in the one synchronisation
            both = torch.cat([torch.stack([p[1] for p in pending]).reshape(-1).int(), flag.reshape(1).int()]).cpu()
            raise_if_knn_truncated(int(both[-1]))
            finals = both[:-1].reshape(len(pending), -1).bool()
        else:
            finals = torch.stack([p[1] for p in pending]).cpu()  # the one synchronisation
        out = []
        for (scores, final, mask_bool), f in zip(pending, finals):
            if scores.is_cuda:  # (mask_bool holds the logits here)
                out.append(self._cut_proposals(mask_bool, scores, f, fg_idxs, 0.2, num_points))
                continue
            if not bool(f.any()):
                out.append(([], []))
                continue
            masks_final = mask_bool[final]
            proposals = torch.zeros((masks_final.shape[0], num_points), dtype=torch.int, device=scores.device)
            inst, pts = torch.nonzero(masks_final, as_tuple=True)
            proposals[inst, fg_idxs[pts]] = 1
            out.append((scores[final], proposals))
        return out

    def forward(self, support_dict, scene_dict, training=True, remember=False, support_embeddings=None):
        cfg = self.cfg
        outputs = {}
        batch_idxs = scene_dict["locs"][:, 0].int()
        locs_float = scene_dict["locs_float"]
        batch_offsets = scene_dict["batch_offsets"]
        batch_size = len(batch_offsets) - 1
        assert batch_size > 0
        pc_dims = [scene_dict["pc_mins"], scene_dict["pc_maxs"]]

        if remember:
            (context_locs, context_feats, pre_enc_inds, fg_idxs, batch_offsets, output_feats_, batch_idxs_, locs_float_,
             batch_offsets_, semantic_preds_, semantic_scores, query_locs, mask_features_, geo_dists) = self.cache_data
            outputs["semantic_scores"] = semantic_scores
        else:
            output_feats, semantic_scores, semantic_preds = self.forward_backbone(scene_dict, batch_size)
            outputs["semantic_scores"] = semantic_scores
            fg = semantic_preds >= 4 if cfg.train_fold == cfg.cvfold else semantic_preds == 3
            fg_idxs = torch.nonzero(fg).view(-1)
            batch_idxs_ = batch_idxs[fg_idxs]
            batch_offsets_ = get_batch_offsets(batch_idxs_, batch_size)
            locs_float_, output_feats_ = locs_float[fg_idxs], output_feats[fg_idxs]
            semantic_preds_ = semantic_preds[fg_idxs]
            ctx = torch.enable_grad if self.training and "mask_tower" not in self.fix_module else torch.no_grad
            with ctx():
                mask_features_ = self.mask_tower(output_feats_.unsqueeze(2).permute(2, 1, 0)).permute(2, 1, 0)
            max_step = 128 if self.training else 256
            geo_dists = None
            offs_ = _offsets_list(batch_offsets_)
            graphs = None
            if locs_float_.is_cuda and min(offs_[b + 1] - offs_[b] for b in range(batch_size)) > 0:
                graphs = knn_graphs(locs_float_, batch_offsets_, batch_size, neighbor=64, radius=0.05)
            if graphs is not None and not torch.is_grad_enabled() and os.environ.get("GF_OVERLAP", "1") != "0":
                # inference on the GPU: sampling cut after the query picks, BFS beside the rest of it (GeoFormer)
                contexts, geo_dists = self._aggregate_geodesic_overlapped(
                    locs_float_, output_feats_, batch_offsets_, batch_size, graphs, max_step, sample=False,
                    epilogue=False)
                self._join_side_stream()
            else:
                contexts = self.forward_aggregator(locs_float_, output_feats_, batch_offsets_, batch_size)
            if contexts is None:
                outputs["mask_predictions"] = None
                return outputs
            context_locs, context_feats, pre_enc_inds = contexts
            query_locs = context_locs[:, :cfg.n_query_points, :]
            if geo_dists is None:
                geo_dists = cal_geodesic(pre_enc_inds, locs_float_, batch_offsets_, max_step=max_step, neighbor=64,
                                         radius=0.05, n_queries=cfg.n_query_points, graphs=graphs)
            # the kNN truncation flag: read with the criterion's values (training) / the proposals' count (inference)
            self._knn_flag = knn_truncated(graphs)
            self.cache_data = (context_locs, context_feats, pre_enc_inds, fg_idxs, batch_offsets, output_feats_,
                               batch_idxs_, locs_float_, batch_offsets_, semantic_preds_, semantic_scores, query_locs,
                               mask_features_, geo_dists)

        if len(fg_idxs) == 0:
            outputs["proposal_scores"] = None
            return outputs
        if support_embeddings is None:
            support_embeddings = self.process_support(support_dict, training)  # batch x channel

        s = support_embeddings.unsqueeze(1).repeat(1, cfg.n_decode_point, 1)
        aggregation = torch.cat([context_feats * s, context_feats - s, context_feats], dim=2)  # B x nc x 3C
        dec_outputs = self.forward_decoder(context_locs, aggregation, query_locs, pc_dims, geo_dists, pre_enc_inds)
        if not training:
            dec_outputs = dec_outputs[-1:, ...]
        else:
            idxs_sub, idxs_sub_raw = random_downsample(batch_offsets_, batch_size, n_subsample=30000,
                                                         host_offsets=_offsets_list(batch_offsets_))
            geo_dists = [geo_dists[b][:, idxs_sub_raw[b]] for b in range(batch_size)]
            fg_idxs = fg_idxs[idxs_sub]
            mask_features_, locs_float_, batch_idxs_ = mask_features_[idxs_sub], locs_float_[idxs_sub], batch_idxs_[idxs_sub]
            batch_offsets_ = get_batch_offsets(batch_idxs_, batch_size)
        mask_predictions = self.get_mask_prediction(geo_dists, dec_outputs, mask_features_, locs_float_, query_locs,
                                                    batch_offsets_)
        similarity = self.similarity_net(aggregation[:, :cfg.n_query_points, :].flatten(0, 1)).squeeze(-1)
        similarity = similarity.reshape(batch_size, cfg.n_query_points)
        if training:
            outputs.update(fg_idxs=fg_idxs, num_insts=cfg.n_query_points * batch_size, batch_idxs=batch_idxs_,
                           simnet=similarity, mask_predictions=mask_predictions)
            if getattr(self, "_knn_flag", None) is not None:
                outputs["knn_truncated"] = self._knn_flag
            return outputs
        outputs["proposal_scores"] = self.generate_proposal(
            mask_predictions[-1]["mask_logits"], similarity.detach().sigmoid(), fg_idxs, batch_offsets,
            logit_thresh=0.2, score_thresh=cfg.TEST_SCORE_THRESH, npoint_thresh=cfg.TEST_NPOINT_THRESH,
            sim_score_thresh=cfg.similarity_thresh)
        return outputs
