"""Dev probe (GPU): the 640-row node MLP of DGL / RGL-NET forward + backward: csrc/mlp.hip vs library ops."""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import config
from multi_part_assembly_amd.pn_transformer import build_model
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(config.dgl_dgcnn_everyday()).to(dev).train()
m = model.node_mlps[0]
ref = copy.deepcopy(m)
cin = m.conv1.in_channels
for rows in (640, 1280, 2560):
    x = torch.randn(rows // 20, 20, cin, device=dev, requires_grad=True)
    w = torch.randn(rows // 20, 20, m.conv3.out_channels, device=dev)
    def run(mod, hip):
        mod.MIN_ROWS = 1 if hip else 10 ** 9
        for _ in range(3):
            mod.zero_grad(); (mod(x) * w).sum().backward()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            mod.zero_grad(); (mod(x) * w).sum().backward()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20
    print("node MLP (cin %d) fwd+bwd at %d rows: hip %.3f ms, library %.3f ms" % (cin, rows, run(m, True), run(ref, False)))
