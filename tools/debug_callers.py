import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import param_fill
from multi_part_assembly_amd import config
from multi_part_assembly_amd.pn_transformer import build_model
name = sys.argv[1]
cfgfn = {"dgl_step": config.dgl_everyday, "rgl_net_step": config.rgl_net_everyday, "global_semantic_step": config.global_partnet_chair}[name]
z = dict(np.load(os.path.join(R, "tests", "golden", name + ".npz"), allow_pickle=False))
dev = torch.device("cuda:0")
cfg = cfgfn(); cfg.model.pc_feat_dim = int(z["cfg"][0]); cfg.data.max_num_part = 5
seed = int(z["seed"][0]); torch.manual_seed(seed)
model = build_model(cfg); param_fill.fill_parameters(model, seed); model.to(dev).train()
data = {k[5:]: torch.from_numpy(z[k].copy()).to(dev) for k in z if k.startswith("data.")}
torch.manual_seed(seed + 1)
res = model.forward_pass(data, mode="train"); res["loss"].backward()
for k in z:
    if k.startswith("loss."):
        print(k, float(res[k[5:]]), float(z[k]))
rows = []
for k, p in model.named_parameters():
    a = p.grad.cpu().numpy().reshape(-1)
    if "grad." + k in z:
        ref = z["grad." + k]
    elif "grad." + k + "#sample" in z:
        idx = np.linspace(0, a.size - 1, param_fill.SAMPLE).astype(np.int64); a = a[idx]; ref = z["grad." + k + "#sample"]
    else:
        continue
    rows.append((np.abs(a - ref).max() / (np.abs(ref).max() + 1e-12), np.abs(ref).max(), k))
for r in sorted(rows, reverse=True)[:25]:
    print("%.4e  refmax %.3e  %s" % r)

# sensitivity: the same step with the input perturbed by 1e-6 (relative) — how far do the gradients move?
g0 = {k: p.grad.clone() for k, p in model.named_parameters()}
model.zero_grad()
param_fill.fill_parameters(model, seed)  # (running stats back to the start)
data2 = dict(data); data2["part_pcs"] = data["part_pcs"] * (1 + 1e-6 * torch.randn_like(data["part_pcs"]))
torch.manual_seed(seed + 1)
res2 = model.forward_pass(data2, mode="train"); res2["loss"].backward()
print("loss", float(res["loss"]), float(res2["loss"]))
rows = []
for k, p in model.named_parameters():
    a, b = g0[k], p.grad
    rows.append((float((a - b).abs().max() / (a.abs().max() + 1e-12)), float(a.abs().max()), k))
print("gradient change under a 1e-6 input perturbation:")
for r in sorted([r for r in rows if r[1] > 1e-5], reverse=True)[:12]:
    print("%.4e  max %.3e  %s" % r)
