"""Where do the step's small torch launches (fill / add / mul / copy ...) come from?  torch.profiler with stacks over
two eager training steps; prints, per aten op that launches a kernel, the call count and the innermost repo frame."""
import collections
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from torch.profiler import ProfilerActivity, profile

from multi_part_assembly_amd import config, synthetic
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.trainer import Trainer

dev = torch.device("cuda:0")
cfg = config.pn_transformer_everyday()
torch.manual_seed(0)
model = build_model(cfg).to(dev)
trainer = Trainer(model, cfg, use_graph=False)
batch = synthetic.make_batch(32, 20, 1000, preset="everyday", seed=1234, device=dev)
batch.pop("num_parts")
for i in range(3):
    trainer.train_step(batch, i)
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(STEPS):
        trainer.train_step(batch, 3 + i)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or any(c.name.startswith("aten::") for c in ev.cpu_children):
        continue
    if not any("Launch" in c.name or "launch" in c.name for c in ev.cpu_children):
        continue  # leaf aten ops that launch a kernel
    frame = next((f for f in ev.stack if "multi_part_assembly_amd" in f or "bench" in f), ev.stack[0] if ev.stack else "?")
    agg[(ev.name, frame.split("/root/repo/")[-1][:110])] += 1
for (name, frame), c in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{c / STEPS:6.1f}/step  {name:28s} {frame}")
