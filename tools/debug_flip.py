"""Where does the HIP gradient of edge_mlps.0.* differ from the full float64 / float32 reference gradients?
(tools/_debug_full_edge0.npz: dumped from the reference here, not committed.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import param_fill  # noqa: E402
from multi_part_assembly_amd import config  # noqa: E402
from multi_part_assembly_amd.pn_transformer import build_model  # noqa: E402

name = "rgl_net_dgcnn_artifact_step"
here = os.path.dirname(os.path.abspath(__file__))
z = dict(np.load(os.path.join(here, "..", "tests", "golden", name + ".npz")))
full = dict(np.load(os.path.join(here, "_debug_full_edge0.npz")))
cfg = config.rgl_net_dgcnn_artifact()
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
    if not k.startswith("edge_mlps.0."):
        continue
    a = p.grad.cpu().numpy().astype(np.float64).reshape(full["g64." + k].shape)
    t, r = full["g64." + k], full["g32." + k]
    scale = max(np.abs(t).max(), 1e-12)
    em, er = np.abs(a - t) / scale, np.abs(r - t) / scale
    print(f"{k}: scale {scale:.3e} hip max {em.max():.2e} ref32 max {er.max():.2e}; hip entries > 1e-2: {(em > 1e-2).sum()}, "
          f"> 1e-3: {(em > 1e-3).sum()}; ref32 > 1e-3: {(er > 1e-3).sum()}")
    if em.max() > 1e-2:
        idx = np.argwhere(em > 1e-3)
        rows = sorted({int(i[0]) for i in idx})
        print("   rows with entries > 1e-3:", rows[:20], "count per row:", {r_: int((em[r_] > 1e-3).sum()) for r_ in rows[:8]})
        i = np.unravel_index(np.argmax(em), em.shape)
        print("   worst entry", i, "hip", a[i], "f64", t[i], "f32", r[i])
# the float64 / float32 reference pre-activations of the suspicious channels
for bn in ("bn1", "bn2", "bn3"):
    a64, a32 = full[f"g64.act.edge_mlps.0.{bn}"][0], full[f"g32.act.edge_mlps.0.{bn}"][0]  # [15, C, 5]
    flips = np.argwhere((a64 > 0) != (a32 > 0))
    print(bn, "sign flips between the float32 and float64 REFERENCE passes:", len(flips), flips[:5].tolist())
    lo = np.abs(a64).transpose(1, 0, 2).reshape(a64.shape[1], -1).min(1)
    print("   channels with |z64| min < 1e-5:", np.argwhere(lo < 1e-5).ravel().tolist()[:10])
