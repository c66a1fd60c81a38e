import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import config, synthetic
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.trainer import Trainer
dev = torch.device("cuda:0")
cfg = config.pn_transformer_everyday()
torch.manual_seed(0)
model = build_model(cfg).to(dev)
tr = Trainer(model, cfg, use_graph=True)
batch = synthetic.make_batch(32, 20, 1000, seed=1234, device=dev); batch.pop("num_parts")
for i in range(int(os.environ.get("STEPS", "8"))):
    l = tr.train_step(batch); torch.cuda.synchronize()
    g = tr.flat.flat_grad; p = tr.flat.flat_param
    print(i, "graph" if tr._graph is not None else "eager", float(l), "grad norm", float(g.norm()), "finite", bool(torch.isfinite(g).all()),
          "param norm", float(p.norm()), flush=True)
rows = sorted(((float(p.grad.abs().max()), k) for k, p in model.named_parameters()), reverse=True)[:10]
for r in rows: print(r)
print("eager pass"); 
for i in range(2):
    tr._fwd_bwd(batch); tr.optimizer.prepare_hyper(); tr.optimizer.step_dev(); torch.cuda.synchronize(); print("ok", i, flush=True)
