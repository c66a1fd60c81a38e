import sys, torch
sys.path.insert(0, "/root/repo")
from multi_part_assembly_amd import config, synthetic
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.trainer import Trainer
dev = torch.device("cuda:0")
def fresh(use_graph):
    cfg = config.dgl_dgcnn_everyday()
    torch.manual_seed(3)
    model = build_model(cfg).to(dev)
    return Trainer(model, cfg, use_graph=use_graph, graph_warmup=1)
a, b, g = fresh(False), fresh(False), fresh(True)
for step in range(4):
    batch = synthetic.make_batch(3, 20, 256, preset="everyday", seed=50 + step, device=dev); batch.pop("num_parts")
    la, lb, lg = a.train_step(dict(batch)), b.train_step(dict(batch)), g.train_step(dict(batch))
    print(step, float(la), float(lb), float(lg), "eager-eager param diff", float((a.flat.flat_param - b.flat.flat_param).abs().max()),
          "graph-eager", float((a.flat.flat_param - g.flat.flat_param).abs().max()))
    ga, gb, gg = a.flat.flat_grad, b.flat.flat_grad, g.flat.flat_grad
    print("   grad diff eager-eager", float((ga - gb).abs().max()), "graph-eager", float((ga - gg).abs().max()))
    if step == 1:
        names = []
        off = 0
        for k, p in a.model.named_parameters():
            n = p.numel()
        d = (ga - gg).abs()
        i = int(d.argmax()); print("   worst index", i, "of", d.numel())
        # find the parameter
        for (k, p), (k2, p2) in zip(a.model.named_parameters(), g.model.named_parameters()):
            dd = float((p.grad - p2.grad).abs().max())
            if dd > 0: print("     ", k, dd, float(p.grad.abs().max()))
