"""Host-side cost of an eager training step: cProfile over 20 steps (GPU work is asynchronous, so this is launch
+ Python overhead only).  Usage: python tools/probe_cpu.py [top]"""
import cProfile
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

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
for i in range(5):
    trainer.train_step(batch, i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    trainer.train_step(batch, 5 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
