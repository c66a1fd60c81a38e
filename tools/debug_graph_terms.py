"""Dev probe (GPU): which tensors of the captured step hold garbage after a replay?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import config, synthetic
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.trainer import Trainer
dev = torch.device("cuda:0")
cfg = config.pn_transformer_everyday()
torch.manual_seed(0)
model = build_model(cfg).to(dev).train()
tr = Trainer(model, cfg, use_graph=False)
batch = synthetic.make_batch(32, 20, 1000, seed=1234, device=dev); batch.pop("num_parts")
for i in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
tr.optimizer.prepare_hyper()
with_bwd = os.environ.get("BWD", "1") == "1"
keep = {}
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    tr.optimizer.zero_grad()
    with tr.sink:
        res = model.forward_pass(batch, mode="train")
        keep = {k: v.detach().clone() for k, v in res.items()}
        keep["loss_alias"] = res["loss"].detach()
        if with_bwd:
            res["loss"].backward()
    if os.environ.get("ADAM", "0") == "1":
        tr.optimizer.step_dev()
for r in range(4):
    tr.optimizer.prepare_hyper()
    g.replay(); torch.cuda.synchronize()
    print(r, {k: float(v) for k, v in keep.items()}, "grad norm", float(tr.flat.flat_grad.norm()), flush=True)
