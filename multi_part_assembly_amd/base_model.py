"""`BaseModel` — the training-step contract of the reference's LightningModule, without Lightning
(reference: multi_part_assembly/models/modules/base_model.py:17-464).

Kept: the subclass contract (`forward(data_dict) -> {'rot': Rotation3D, 'trans': Tensor}` plus
`_loss_function`), `training_step / validation_step / forward_pass / loss_function (MoN) /
_calc_loss / _match_parts / _linear_sum_assignment / configure_optimizers`, the loss-term names and
the weighting by `cfg.loss.<term>_w`.  Dropped: everything that needs a pl.Trainer (per-step
`.item()` logging through `self.trainer.profiler`, base_model.py:137-146 — which also removes one
device sync per loss term per step) and wandb visualisation.  The eval-time metrics (`_calc_metrics`,
SURVEY.md §8 row N2) are in eval_utils.py.
"""
from __future__ import annotations

import numpy as np
import os

import torch
import torch.nn as nn

from .chamfer import chamfer_distance
from .eval_utils import calc_connectivity_acc, calc_part_acc, rot_metrics, trans_metrics
from .matching import SUBSAMPLE, match_parts
from .loss import (LossTerms, geometric_assembly_loss, part_order, search_mode, rot_cosine_loss, rot_points_cd_loss,
                   rot_points_l2_loss, shape_cd_loss, trans_l2_loss)
from .rotation import Rotation3D
from .transforms import transform_pc


class BaseModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.rot_type = cfg.model.rot_type
        if self.rot_type != "quat":
            raise NotImplementedError(f"rotation {self.rot_type} is not supported")
        self.pose_dim = 7
        self.semantic = cfg.data.dataset != "geometry"
        self.max_num_part = cfg.data.max_num_part
        self.pc_feat_dim = cfg.model.pc_feat_dim
        self.use_part_label = "part_label" in cfg.data.data_keys
        self.sample_iter = cfg.loss.get("sample_iter", 1)
        # fused HIP loss path for geometric data (csrc/assembly_loss.hip); the per-function path is
        # kept for the semantic datasets and as a cross-check.  keep_pts: also return the transformed
        # clouds (only visualisation needs them).
        self.fused_loss = True
        self.keep_pts = False

    # ---- hooks a trainer calls ---------------------------------------------------------------
    def training_step(self, data_dict, batch_idx=0, optimizer_idx=-1):
        return self.forward_pass(data_dict, mode="train", optimizer_idx=optimizer_idx)["loss"]

    def validation_step(self, data_dict, batch_idx=0):
        return self.forward_pass(data_dict, mode="val", optimizer_idx=-1)

    test_step = validation_step

    def forward_pass(self, data_dict, mode="train", optimizer_idx=-1):
        """data_dict as produced by the reference datasets (geometry_data.py:173-207); `part_quat` is
        wrapped into `part_rot` here, like base_model.py:128-132 (non-destructively)."""
        data_dict = dict(data_dict)
        if "part_rot" not in data_dict:
            data_dict["part_rot"] = Rotation3D(data_dict.pop("part_quat"), rot_type="quat")
        return self.loss_function(data_dict, optimizer_idx=optimizer_idx)

    # ---- GT <-> prediction matching (semantic datasets only) --------------------------------------
    @torch.no_grad()
    def _linear_sum_assignment(self, pts, trans1, rot1, trans2, rot2):
        """Hungarian match between two pose sets of one equivalence group (base_model.py:150-179):
        100 sub-sampled points, p x p Chamfer cost matrix on the GPU, scipy on the host.  Kept with the
        reference's signature as the per-group cross-check of the batched device path `_match_parts` uses."""
        from scipy.optimize import linear_sum_assignment

        p, N, _ = pts.shape
        n = min(SUBSAMPLE, N)
        sample_idx = torch.randperm(N)[:n].to(pts.device).long()
        pts = pts[:, sample_idx]
        pts1 = transform_pc(trans1, rot1, pts, self.rot_type)
        pts2 = transform_pc(trans2, rot2, pts, self.rot_type)
        pts1 = pts1[:, None].expand(p, p, n, 3).reshape(-1, n, 3)
        pts2 = pts2[None].expand(p, p, n, 3).reshape(-1, n, 3)
        dist1, dist2 = chamfer_distance(pts1, pts2)
        cost = (dist1.mean(1) + dist2.mean(1)).view(p, p)
        rind, cind = linear_sum_assignment(cost.cpu().numpy())
        return (torch.from_numpy(rind).type_as(sample_idx), torch.from_numpy(cind).type_as(sample_idx))

    @torch.no_grad()
    def _match_parts(self, part_pcs, pred_trans, pred_rot, gt_trans, gt_rot, match_ids, ids_host=None):
        """Permute the GT poses inside every group of geometrically equivalent parts so that they line up with
        the predictions at minimum Chamfer cost (base_model.py:181-238).  All groups of the batch are matched on
        the device in one call (`matching.match_parts`: cost matrices, scipy's assignment algorithm, permutation);
        the host only draws the point sub-samples — `torch.randperm(N)[:100]` per existing group, in the
        reference's order, so a seeded run consumes the CPU generator exactly as the reference does.  `ids_host`
        (numpy [B,P]) spares the one device-to-host copy of `match_ids` when the loader still has it."""
        B, P, N, _ = part_pcs.shape
        if ids_host is None:
            ids_host = match_ids.long().cpu().numpy()
        n = min(SUBSAMPLE, N)
        G = max(1, int(ids_host.max()))
        idx = torch.zeros((B, G, n), dtype=torch.int32)
        for b in range(B):
            for group in range(1, int(ids_host[b].max()) + 1):
                if (ids_host[b] == group).any():
                    idx[b, group - 1] = torch.randperm(N)[:n].to(torch.int32)
        idx = idx.pin_memory().to(part_pcs.device, non_blocking=True)
        new_trans, new_q = match_parts(part_pcs, pred_trans, pred_rot.rot, gt_trans, gt_rot.rot, match_ids, idx)
        return new_trans, Rotation3D(new_q, rot_type=self.rot_type)

    # ---- loss assembly ------------------------------------------------------------------------------
    def _calc_loss(self, out_dict, data_dict):
        """Loss terms of one prediction, each [B] (base_model.py:240-314)."""
        pred_trans, pred_rot = out_dict["trans"], out_dict["rot"]
        part_pcs, valids = data_dict["part_pcs"], data_dict["part_valids"]
        gt_trans, gt_rot = data_dict["part_trans"], data_dict["part_rot"]
        if self.semantic:
            if "_match_ids_host" not in data_dict:  # one copy per batch, shared by the min-of-N samples
                data_dict["_match_ids_host"] = data_dict["match_ids"].long().cpu().numpy()
            new_trans, new_rot = self._match_parts(part_pcs, pred_trans, pred_rot, gt_trans, gt_rot,
                                                   data_dict["match_ids"], data_dict["_match_ids_host"])
        else:
            new_trans, new_rot = gt_trans.detach(), gt_rot.detach()
        if self.fused_loss and (not self.semantic or os.environ.get("MPA_FUSED_SEMANTIC", "1") != "0"):
            # one fused forward/backward pair instead of the per-function composition below (semantic data: the matched
            # poses are plain inputs of the same five terms; its whole-shape term always normalises as in training,
            # base_model.py:281 of the reference)
            terms, pts = geometric_assembly_loss(part_pcs, pred_trans, pred_rot, new_trans, new_rot,
                                                 valids, training=self.semantic or self.training, ret_pts=self.keep_pts,
                                                 order=self._join_part_order(data_dict), search=self._shape_search())
            loss_dict = LossTerms((k, terms[k]) for k in ("trans_loss", "rot_pt_cd_loss", "transform_pt_cd_loss"))
            if self.cfg.loss.use_rot_loss:
                loss_dict["rot_loss"] = terms["rot_loss"]
            if self.cfg.loss.use_rot_pt_l2_loss:
                loss_dict["rot_pt_l2_loss"] = terms["rot_pt_l2_loss"]
            if self.training and tuple(loss_dict) == terms.stacked[0]:
                loss_dict.stacked = terms.stacked  # every row is used, in order: no select / re-stack
            if not self.training:
                loss_dict.update(self._calc_metrics(data_dict, out_dict, new_trans, new_rot))
            out_dict = {"pred_trans": pred_trans, "pred_rot": pred_rot,
                        "pred_trans_pts": pts[0] if pts else None, "gt_trans_pts": pts[1] if pts else None}
            return loss_dict, out_dict

        loss_dict = {
            "trans_loss": trans_l2_loss(pred_trans, new_trans, valids),
            "rot_pt_cd_loss": rot_points_cd_loss(part_pcs, pred_rot, new_rot, valids),
        }
        loss_dict["transform_pt_cd_loss"], pred_pts, gt_pts = shape_cd_loss(
            part_pcs, pred_trans, new_trans, pred_rot, new_rot, valids, ret_pts=True,
            training=self.semantic or self.training)
        if self.cfg.loss.use_rot_loss:
            loss_dict["rot_loss"] = rot_cosine_loss(pred_rot, new_rot, valids)
        if self.cfg.loss.use_rot_pt_l2_loss:
            loss_dict["rot_pt_l2_loss"] = rot_points_l2_loss(part_pcs, pred_rot, new_rot, valids)
        if not self.training:
            loss_dict.update(self._calc_metrics(data_dict, out_dict, new_trans, new_rot))
        out_dict = {"pred_trans": pred_trans, "pred_rot": pred_rot, "gt_trans_pts": gt_pts,
                    "pred_trans_pts": pred_pts}
        return loss_dict, out_dict

    @torch.no_grad()
    def _calc_metrics(self, data_dict, out_dict, gt_trans, gt_rot):
        """Evaluation-time metrics (base_model.py:316-339): part accuracy always; connectivity accuracy for the
        semantic datasets that annotate contacts; translation / rotation MSE, RMSE, MAE for geometric data."""
        part_pcs, valids = data_dict["part_pcs"], data_dict["part_valids"]
        pred_trans, pred_rot = out_dict["trans"], out_dict["rot"]
        metrics = {"part_acc": calc_part_acc(part_pcs, pred_trans, gt_trans, pred_rot, gt_rot, valids)}
        if self.semantic and "contact_points" in data_dict:
            metrics["connectivity_acc"] = calc_connectivity_acc(pred_trans, pred_rot, data_dict["contact_points"])
        if not self.semantic:
            for m in ("mse", "rmse", "mae"):
                metrics[f"trans_{m}"] = trans_metrics(pred_trans, gt_trans, valids, metric=m)
                metrics[f"rot_{m}"] = rot_metrics(pred_rot, gt_rot, valids, metric=m)
        return metrics

    @staticmethod
    def aggregate_eval(outputs, prefix="val"):
        """Batch-size-weighted average of the per-batch dictionaries `validation_step` returns — what the
        reference's `validation_epoch_end` logs (base_model.py:69-84)."""
        sizes = torch.tensor([float(o["batch_size"]) for o in outputs], device=outputs[0]["loss"].device)
        keys = [k for k in outputs[0] if k != "batch_size"]
        return {f"{prefix}/{k}": (torch.stack([o[k] for o in outputs]) * sizes).sum() / sizes.sum() for k in keys}

    def _loss_function(self, data_dict, out_dict={}, optimizer_idx=-1):
        raise NotImplementedError

    def _shape_search(self):
        """`cfg.loss.shape_search` ("brute" | "grid" | "leaf" | "auto"; None: MPA_SHAPE_SEARCH, else "grid"): the exact
        searches behind the fused loss's Chamfer terms.  Configurations for data with many small parts ask for "auto"."""
        return self.cfg.loss.get("shape_search", None)

    # ---- the batch's k-d order (csrc/leaf_nn.hip): once per batch, beside the encoder --------------------------------------
    def prepare_streams(self, dev):
        """Create the side stream of `_start_part_order` now (Trainer.__init__ calls this): a capture then forks to a
        stream that already exists instead of creating one mid-capture."""
        dev = torch.device(dev)
        side = getattr(self, "_order_stream", None)
        if dev.type == "cuda" and (side is None or side.device != dev):
            self._order_stream = torch.cuda.Stream(device=dev)

    def _start_part_order(self, data_dict):
        """Both Chamfer searches of the fused loss run on a k-d order of each part's points that depends on the batch
        only (`loss.part_order`).  It is computed ONCE per `loss_function` call — every GNN iteration and every min-of-N
        sample of the step reuses it — and on a side stream, so that it runs beside the encoder instead of in front of
        the first loss evaluation (works under graph capture: the side stream forks from and joins the capturing one).
        Returns a shallow copy of `data_dict` carrying the pending order; the caller's dict is never touched (a cached
        order would go stale when a loader refills its batch tensors in place)."""
        pcs = data_dict["part_pcs"]
        if not (self.fused_loss and not self.semantic and pcs.is_cuda) or search_mode(self._shape_search()) < 2:
            return data_dict  # (the grid / brute-force searches need no order)
        dev = pcs.device
        side = getattr(self, "_order_stream", None)
        if side is None or side.device != dev:
            side = self._order_stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)  # (the batch may have been produced on the current stream)
        with torch.cuda.stream(side):
            order = part_order(pcs, data_dict["part_valids"])
        if order is None:  # N > 2048: the loss keeps its grid search
            return data_dict
        out = dict(data_dict)
        out["_part_order"] = [order, side]
        return out

    @staticmethod
    def _join_part_order(data_dict):
        """The order started by `_start_part_order` (None if there is none); the first caller makes the current stream wait."""
        pending = data_dict.get("_part_order")
        if pending is None:
            return None
        order, side = pending
        if side is not None:
            cur = torch.cuda.current_stream(order.device)
            cur.wait_stream(side)
            order.record_stream(cur)
            pending[1] = None
        return order

    def loss_function(self, data_dict, optimizer_idx=-1):
        """Min-of-N over `sample_iter` stochastic predictions, per sample (base_model.py:348-387)."""
        data_dict = self._start_part_order(data_dict)
        try:
            return self._loss_function_impl(data_dict, optimizer_idx)
        finally:
            self._join_part_order(data_dict)  # (a path that never evaluated the fused loss must still join the side stream)

    def _loss_function_impl(self, data_dict, optimizer_idx=-1):
        if self.sample_iter == 1:
            # one prediction: stack the terms once and weight / average them with a handful of launches instead of
            # five per term (the step is ~200 launches, these would be ~40 of them)
            sample_loss, _ = self._loss_function(data_dict, {}, optimizer_idx=optimizer_idx)
            keys = list(sample_loss)
            stacked = getattr(sample_loss, "stacked", None)
            if stacked is not None and list(stacked[0]) == keys:
                terms = stacked[1]                                                            # [K, B] as computed
            else:
                terms = torch.stack([sample_loss[k] for k in keys], dim=0)                   # [K, B]
            cache = getattr(self, "_loss_weight_cache", None)
            if cache is None or cache[0] != keys or cache[1].device != terms.device:
                w = [float(self.cfg.loss[f"{k}_w"]) if k.endswith("_loss") else 0.0 for k in keys]
                cache = (keys, torch.tensor(w, dtype=terms.dtype, device=terms.device))
                self._loss_weight_cache = cache
            # mean over the batch of sum_k w_k t_kb = sum_k w_k mean_b t_kb: one launch each way on the device the fused
            # loss runs on (mean + dot and their two backward kernels were 23 us of the 2.07 ms step)
            if terms.is_cuda and terms.dtype == torch.float32:
                from .loss import weighted_term_means
                means, total = weighted_term_means(terms, cache[1])
            else:
                means = terms.mean(dim=1)                                                    # [K]
                total = torch.dot(means, cache[1])
            result = {k: means[i] for i, k in enumerate(keys)}
            result["loss"] = total
            if not self.training:
                result["batch_size"] = terms.shape[1]
            return result
        samples, out_dict = None, {}
        for _ in range(self.sample_iter):
            sample_loss, out_dict = self._loss_function(data_dict, out_dict, optimizer_idx=optimizer_idx)
            if samples is None:
                samples = {k: [] for k in sample_loss}
            for k, v in sample_loss.items():
                samples[k].append(v)
        stacked = {k: torch.stack(v, dim=0) for k, v in samples.items()}  # [sample_iter, B]
        total = 0.0
        for k, v in stacked.items():
            if k.endswith("_loss"):
                total = total + v * self.cfg.loss[f"{k}_w"]
        stacked["loss"] = total
        if self.sample_iter == 1:
            result = {k: v[0].mean() for k, v in stacked.items()}
        else:
            best = total.argmin(0)
            cols = torch.arange(best.shape[0], device=best.device)
            result = {k: v[best, cols].mean() for k, v in stacked.items()}
        if not self.training:
            result["batch_size"] = total.shape[1]
        return result

    # ---- optimiser ------------------------------------------------------------------------------------
    def configure_optimizers(self, steps_per_epoch=None):
        """Adam(lr, wd=0) [AdamW when wd > 0], cosine schedule with warm-up stepped per epoch
        (base_model.py:389-425).  Returns (optimizer, lr_lambda_or_None)."""
        from .optim import FusedAdam, cosine_warmup_lr

        opt_cfg = self.cfg.optimizer
        optimizer = FusedAdam(self.parameters(), lr=opt_cfg.lr, weight_decay=opt_cfg.weight_decay,
                              clip_grad=opt_cfg.get("clip_grad", None))
        if opt_cfg.weight_decay > 0:  # AdamW parameter groups: biases and normalisation weights are not decayed
            from .optim import decay_mask_for
            optimizer.decay_mask = decay_mask_for(self, optimizer.flat)
        schedule = None
        if opt_cfg.lr_scheduler:
            assert opt_cfg.lr_scheduler == "cosine"
            total = self.cfg.exp.num_epochs
            schedule = cosine_warmup_lr(total, int(total * opt_cfg.warmup_ratio), opt_cfg.lr,
                                        opt_cfg.lr / opt_cfg.lr_decay_factor)
        return optimizer, schedule

    def _wrap_rotation(self, rot_tensor):
        return Rotation3D(rot_tensor, rot_type=self.rot_type)
