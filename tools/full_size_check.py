"""Dev tool: one DGL + DGCNN training-mode forward_pass + backward at the benchmark's part size (BASELINE.json
configs[2]: P = 20, N = 1000; B given) on the HIP path, on the float32 CPU oracle (oracle/callers.py) and on the SAME
oracle in float64 — loss and every parameter gradient, each float32 result measured against the float64 one.
    python tools/full_size_check.py [B=4] [c3|c2]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import config, synthetic
from multi_part_assembly_amd.pn_transformer import build_model
from oracle import callers as oc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
which = sys.argv[2] if len(sys.argv) > 2 else "c3"
dev = torch.device("cuda:0")
cfg = config.dgl_dgcnn_everyday() if which == "c3" else config.pn_transformer_everyday()
torch.manual_seed(0)
model = build_model(cfg)
if which != "c3":  # dropout off: the three evaluations see the same network
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
batch = synthetic.make_batch(B, 20, 1000, preset="everyday", seed=1234, device=dev)
batch.pop("num_parts", None)
model.to(dev).train()
loss = model.training_step(batch, 0)
loss.backward()
torch.cuda.synchronize()
hip = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.grad is not None}
torch.set_num_threads(16)


def oracle(dt):
    sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    params = {k: sd[k].requires_grad_() for k, _ in model.named_parameters()}
    cb = {k: (v.cpu().to(dt) if v.is_floating_point() else v.cpu()) for k, v in batch.items() if hasattr(v, "cpu")}
    t0 = time.time()
    if which == "c3":
        losses = oc.dgl_loss(sd, cb, cfg.model.gnn_iter, cfg.model.encoder, True, {})
    else:
        from oracle import nets as on
        losses = on.pn_transformer_loss(sd, cb, cfg.model.transformer_layers, cfg.model.transformer_heads, training=True,
                                        stats_out={})[0]
    losses["loss"].backward()
    print(f"oracle {dt}: {time.time() - t0:.1f} s, loss {float(losses['loss'].detach()):.8f}")
    return float(losses["loss"].detach()), {k: p.grad.double() for k, p in params.items() if p.grad is not None}


l32, g32 = oracle(torch.float32)
l64, g64 = oracle(torch.float64)
print(f"loss: hip {float(loss.detach()):.8f}  |hip - f64| / f64 = {abs(float(loss.detach()) - l64) / l64:.2e}   "
      f"|oracle32 - f64| / f64 = {abs(l32 - l64) / l64:.2e}")
rows = []
for k, b in g64.items():
    if k not in hip:
        continue
    s = float(b.abs().max()) + 1e-30
    rows.append((float((hip[k] - b).abs().max()) / s, float((g32[k] - b).abs().max()) / s, s, k))
rows.sort(reverse=True)
print("  |hip-f64|   |o32-f64|   max|g64|   tensor")
for eh, eo, s, k in rows:
    print(f"  {eh:.3e}  {eo:.3e}  {s:.2e}  {k}")
