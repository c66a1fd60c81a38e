#!/usr/bin/env python3
"""Hunt for run-to-run non-determinism in the c2 training step: two trainers with identical initial weights walk the same
batch sequence side by side; after every step their flat parameter buffers must be bit-equal.  At the first divergence the
step's gradients are compared per parameter to name the module.   python tools/exp_race_hunt.py [steps=3000] [config=c2] [bf16]
(c5: RGL-NET draws its GRU initial states on the CPU generator — two trainers taking turns would see different draws, so the
draw is pinned to one fixed tensor per batch size for the hunt; c1 goes through the generic Chamfer backward, whose float
atomics are unordered BY CONTRACT, as the reference's: its trajectory is not expected to repeat.)"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from multi_part_assembly_amd.pn_transformer import build_model  # noqa: E402
from multi_part_assembly_amd.trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
name = sys.argv[2] if len(sys.argv) > 2 else "c2"
bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"
dev = torch.device("cuda", 0)
cfg, batch0, _, B, P = bench.workload(name, 0, dev)
batches = [batch0] + [bench.workload(name, 0, dev, k)[1] for k in range(1, 4)]
for bt in batches:
    bt.pop("num_parts", None)


if name == "c5":
    from multi_part_assembly_amd import gnn
    _fixed = {}

    def _fixed_hidden(self, B, device=None):
        if B not in _fixed:
            _fixed[B] = torch.randn((2, B, 2 * self.pc_feat_dim), generator=torch.Generator().manual_seed(11)).to(device)
        return _fixed[B]

    gnn.RGLNet._init_gru_hidden = _fixed_hidden


def make():
    torch.manual_seed(0)
    model = build_model(cfg).to(dev)
    if bf16:
        from multi_part_assembly_amd.encoder import PointNet
        for m in model.modules():
            if isinstance(m, PointNet):
                m.precision = "bf16"
    return Trainer(model, cfg)


a, b = make(), make()
assert torch.equal(a.flat.flat_param, b.flat.flat_param)
names = [n for n, p in a.model.named_parameters() if p.requires_grad]
by_id = {id(p): n for n, p in a.model.named_parameters()}
events = 0
for i in range(steps):
    bt = batches[i % 4]
    la = a.train_step(bt, i)
    lb = b.train_step(bt, i)
    if not torch.equal(a.flat.flat_param, b.flat.flat_param):
        events += 1
        ga, gb = a.flat.flat_grad, b.flat.flat_grad
        print(f"step {i}: parameters diverge; loss {float(la)!r} vs {float(lb)!r}; gradients differ in:")
        for p, off in zip(a.flat.params, a.flat.offsets):
            d = (ga[off:off + p.numel()] - gb[off:off + p.numel()]).abs()
            if float(d.max()) > 0:
                print(f"    {by_id.get(id(p), '?'):45s} {int((d > 0).sum()):7d} of {p.numel():7d} entries, max |diff| {float(d.max()):.3e} "
                      f"(|grad| max {float(ga[off:off + p.numel()].abs().max()):.3e})")
        # re-synchronise b to a and go on hunting
        b.flat.flat_param.copy_(a.flat.flat_param)
        for (ka, va), (kb, vb) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
            if not va.is_floating_point() or "running" in ka or "num_batches" in ka:
                vb.copy_(va)
        if hasattr(b.optimizer, "state_buffers"):
            pass
        if events >= 3:
            break
print(f"{steps if events < 3 else i + 1} steps, {events} divergence events")
