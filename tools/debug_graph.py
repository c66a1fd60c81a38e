import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import config
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.trainer import Trainer
dev = torch.device("cuda:0")
z = dict(np.load("tests/golden/pn_transformer_step.npz"))
def fresh(**kw):
    d, heads, ffn, layers = (int(v) for v in z["cfg"])
    cfg = config.pn_transformer_everyday()
    cfg.model.pc_feat_dim, cfg.model.transformer_heads = d, heads
    cfg.model.transformer_feat_dim, cfg.model.transformer_layers = ffn, layers
    cfg.data.max_num_part = 5; cfg.optimizer.lr_scheduler = ""
    m = build_model(cfg)
    m.load_state_dict({k[4:]: torch.from_numpy(v.copy()) for k, v in z.items() if k.startswith("sd0.")})
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention): mod.dropout = 0.0
    m.to(dev)
    return Trainer(m, cfg, **kw)
batch = {k[5:]: torch.from_numpy(v).to(dev) for k, v in z.items() if k.startswith("data.")}
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
a = fresh()
b = fresh(use_graph=True, graph_warmup=1) if mode == "graph" else fresh()
for step in range(5):
    la = a.train_step(batch); lb = b.train_step(batch)
    ga, gb = a.flat.flat_grad, b.flat.flat_grad
    print(step, float(la), float(lb), "grad diff max", float((ga-gb).abs().max()), "mean", float((ga-gb).abs().mean()),
          "param diff mean", float((a.flat.flat_param-b.flat.flat_param).abs().mean()))
names = [k for k, _ in a.model.named_parameters()]
pa = dict(a.model.named_parameters()); pb = dict(b.model.named_parameters())
rows = sorted(((float((pa[k]-pb[k]).abs().mean()), k) for k in names), reverse=True)[:8]
for r in rows: print(r)
