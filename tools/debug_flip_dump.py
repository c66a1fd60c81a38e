"""Run in tests/golden/ of the BUILD container (imports /root/reference through the shim, like make_golden.py): dumps the full
float32 / float64 reference gradients and BatchNorm outputs of edge_mlps.0 for tools/debug_flip.py -> tools/_debug_full_edge0.npz
(git-ignored)."""
import sys, os, importlib
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import _reference_shim as shim
shim.import_reference()
import make_golden as mg, param_fill
from multi_part_assembly.models import build_model
cfg = mg._load_cfg("configs/rgl_net", "rgl_net-32x1-cosine_200e-everyday")
cfg.model.encoder = "dgcnn"; cfg.model.pc_feat_dim = 64; cfg.data.max_num_part = 5
g = torch.Generator().manual_seed(1015)
data = mg.synthetic_batch(g, 3, 5, 64, [4, 5, 5]); data["part_pcs"] = data["part_pcs"] * 0.33
seed = 1015
out = {}
for dt, tag in ((torch.float32, "g32"), (torch.float64, "g64")):
    torch.manual_seed(seed); model = build_model(cfg); param_fill.fill_parameters(model, seed); mg.zero_dropout(model)
    model.to(dt).train()
    acts = {}
    def mk(name):
        def hook(m, i, o): acts.setdefault(name, []).append(o.detach().clone())
        return hook
    for n, m in model.named_modules():
        if n in ("edge_mlps.0.bn2", "edge_mlps.0.bn1", "edge_mlps.0.bn3"): m.register_forward_hook(mk(n))
    torch.manual_seed(seed + 1)
    ld = model.forward_pass({k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in data.items()}, mode="val", optimizer_idx=-1)
    ld["loss"].backward()
    for k, p in model.named_parameters():
        if k.startswith("edge_mlps.0."): out[f"{tag}.{k}"] = p.grad.detach().double().numpy()
    for n, v in acts.items(): out[f"{tag}.act.{n}"] = torch.stack(v).double().numpy()
np.savez_compressed("/root/repo/tools/_debug_full_edge0.npz", **out)
print({k: v.shape for k, v in out.items()})
