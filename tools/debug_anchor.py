"""Per-entry look at the gradient tensors of a caller fixture that sit further from the float64 anchor than the float32
reference does (tests/test_callers_gpu.py): how many entries carry the deviation?   python tools/debug_anchor.py NAME"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import param_fill  # noqa: E402
from multi_part_assembly_amd import config  # noqa: E402
from multi_part_assembly_amd.pn_transformer import build_model  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "rgl_net_dgcnn_artifact_step"
CASES = {"dgl_step": config.dgl_everyday, "rgl_net_step": config.rgl_net_everyday,
         "dgl_dgcnn_step": config.dgl_dgcnn_everyday, "rgl_net_dgcnn_artifact_step": config.rgl_net_dgcnn_artifact}
z = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", name + ".npz")))
cfg = CASES[name]()
cfg.model.pc_feat_dim = int(z["cfg"][0])
cfg.data.max_num_part = 5
seed = int(z["seed"][0])
dev = torch.device("cuda:0")
torch.manual_seed(seed)
model = build_model(cfg)
param_fill.fill_parameters(model, seed)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
model.to(dev).train()
data = {k[5:]: torch.from_numpy(z[k].copy()).to(dev) for k in z if k.startswith("data.")}
torch.manual_seed(seed + 1)
res = model.forward_pass(data, mode="train")
res["loss"].backward()
for k, p in model.named_parameters():
    if ("grad64." + k) in z:
        t = z["grad64." + k].astype(np.float64)
        r = z["grad." + k].astype(np.float64)
        a = p.grad.cpu().numpy().reshape(-1).astype(np.float64)
    elif ("grad64." + k + "#sample") in z:
        idx = np.linspace(0, p.numel() - 1, param_fill.SAMPLE).astype(np.int64)
        t = z["grad64." + k + "#sample"].astype(np.float64)
        r = z["grad." + k + "#sample"].astype(np.float64)
        a = p.grad.cpu().numpy().reshape(-1).astype(np.float64)[idx]
    else:
        continue
    scale = max(np.abs(t).max(), 1e-4)
    em, er = np.abs(a - t) / scale, np.abs(r - t) / scale
    if em.max() > 2 * er.max() + 1e-4:
        print(f"{k}: n={a.size} scale={scale:.3e} mine max {em.max():.2e} (entries > 1e-3: {(em > 1e-3).sum()}, > 1e-4: "
              f"{(em > 1e-4).sum()}), ref32 max {er.max():.2e} (entries > 1e-3: {(er > 1e-3).sum()}, > 1e-4: {(er > 1e-4).sum()})")
