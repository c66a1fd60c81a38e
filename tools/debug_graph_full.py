import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import config, synthetic
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.trainer import Trainer
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
def fresh(**kw):
    cfg = config.pn_transformer_everyday()
    torch.manual_seed(0)
    m = build_model(cfg)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention): mod.dropout = 0.0
    m.to(dev)
    return Trainer(m, cfg, **kw)
batch = synthetic.make_batch(B, 20, N, seed=1234, device=dev); batch.pop("num_parts")
a = fresh(); b = fresh(use_graph=True, graph_warmup=1)
for step in range(6):
    la = a.train_step(batch); lb = b.train_step(batch); torch.cuda.synchronize()
    ga, gb = a.flat.flat_grad, b.flat.flat_grad
    d = (ga - gb).abs()
    print(step, "graph" if b._graph is not None else "eager", float(la), float(lb), "grad diff max", float(d.max()), "at", int(d.argmax()), "of", ga.numel(), flush=True)
    if float(d.max()) > 0:
        i = int(d.argmax())
        off = b.flat.offsets
        import bisect
        k = bisect.bisect_right(off, i) - 1
        names = {id(p): n for n, p in b.model.named_parameters()}
        print("   param:", names[id(b.flat.params[k])], "eager", float(ga[i]), "graph", float(gb[i]))
        break
