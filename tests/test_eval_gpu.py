"""Evaluation path: the metric functions and the evaluation-mode forward_pass against
fixtures produced by the reference's utils/eval_utils.py and `BaseModel.forward_pass` (tests/golden/make_golden.py)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import param_fill  # noqa: E402

from multi_part_assembly_amd import config, eval_utils  # noqa: E402
from multi_part_assembly_amd.pn_transformer import build_model  # noqa: E402
from multi_part_assembly_amd.rotation import Rotation3D  # noqa: E402

pytestmark = pytest.mark.gpu


def test_metric_functions_match_reference(golden, cuda_device):
    z = golden("eval_metrics")
    t = lambda k: torch.from_numpy(z[k].copy()).to(cuda_device)
    pcs, valids = t("pcs"), t("valids")
    gt_t, pr_t = t("gt_t"), t("pr_t")
    r_gt, r_pr = Rotation3D(t("gt_q")), Rotation3D(t("pr_q"))
    np.testing.assert_array_equal(eval_utils.calc_part_acc(pcs, pr_t, gt_t, r_pr, r_gt, valids).cpu().numpy(), z["part_acc"])
    for m in ("mse", "rmse", "mae"):
        np.testing.assert_allclose(eval_utils.trans_metrics(pr_t, gt_t, valids, m).cpu().numpy(), z[f"trans_{m}"], rtol=1e-5)
        np.testing.assert_allclose(eval_utils.rot_metrics(r_pr, r_gt, valids, m).cpu().numpy(), z[f"rot_{m}"], rtol=2e-4)
    np.testing.assert_allclose(r_pr.to_euler().cpu().numpy(), z["euler_pr"], rtol=1e-4, atol=1e-3)
    contact = t("contact_points")
    np.testing.assert_allclose(eval_utils.calc_connectivity_acc(pr_t, r_pr, contact).cpu().numpy(),
                               z["connectivity_acc_pred"], rtol=1e-6)
    ident = Rotation3D(torch.tensor([1.0, 0, 0, 0], device=cuda_device).repeat(*pr_t.shape[:2], 1))
    np.testing.assert_allclose(eval_utils.calc_connectivity_acc(torch.zeros_like(pr_t), ident, contact).cpu().numpy(),
                               z["connectivity_acc_zero"], rtol=1e-6)


def test_evaluation_forward_pass_matches_reference(golden, cuda_device):
    z = golden("pn_transformer_eval")
    d, heads, ffn, layers = (int(v) for v in z["cfg"])
    cfg = config.pn_transformer_everyday()
    cfg.model.pc_feat_dim, cfg.model.transformer_heads = d, heads
    cfg.model.transformer_feat_dim, cfg.model.transformer_layers = ffn, layers
    cfg.data.max_num_part = 5
    seed = int(z["seed"][0])
    torch.manual_seed(seed)
    model = build_model(cfg)
    param_fill.fill_parameters(model, seed)
    model.to(cuda_device).eval()
    data = {k[5:]: torch.from_numpy(z[k].copy()).to(cuda_device) for k in z if k.startswith("data.")}
    with torch.no_grad():
        res = model.validation_step(data, 0)
    for k in z:
        if k.startswith("res."):
            got = res[k[4:]]
            np.testing.assert_allclose(float(got), float(z[k]), rtol=3e-4, atol=1e-6, err_msg=k)
    agg = model.aggregate_eval([res, res])
    np.testing.assert_allclose(float(agg["val/loss"]), float(z["res.loss"]), rtol=3e-4)
