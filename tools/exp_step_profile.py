"""GPU time of every step from process start (events on torch's stream, no host sync inside the loop): how long does a fresh
process take to reach its steady step time?   python tools/exp_step_profile.py [config=c2] [steps=80]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda", 0)
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.trainer import Trainer

cfg, batch, desc, B, P = bench.workload(cfgname, 0, dev)
batches = [batch] + [bench.workload(cfgname, 0, dev, k)[1] for k in range(1, 4)]
for b in batches:
    b.pop("num_parts")
torch.manual_seed(0)
trainer = Trainer(build_model(cfg).to(dev), cfg, use_graph=False)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for i in range(steps):
    trainer.train_step(batches[i % 4], i)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
print("steps 0-4  ", " ".join(f"{v:.2f}" for v in ms[:5]))
for lo in range(5, steps, 10):
    seg = ms[lo:lo + 10]
    print(f"steps {lo}-{lo + len(seg) - 1}: mean {sum(seg) / len(seg):.3f}  min {min(seg):.3f} max {max(seg):.3f}")
