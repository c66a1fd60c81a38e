"""Work statistics of the whole-shape grid search in the benchmark's regime (needs the instrumented library build:
hipcc -DMPA_GRID_STATS grid_nn.hip, linked as build_variants/grid_stats.so and copied over libmpa_hip.so on the GPU box):
work items, active lanes per item, candidate records scanned per item."""
import ctypes
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from multi_part_assembly_amd import _lib, config, synthetic
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.trainer import Trainer

dev = torch.device("cuda:0")
cfg = config.pn_transformer_everyday()
torch.manual_seed(0)
model = build_model(cfg).to(dev)
trainer = Trainer(model, cfg, use_graph=False)
batch = synthetic.make_batch(32, 20, 1000, preset="everyday", seed=1234, device=dev)
num_parts = batch.pop("num_parts")
L = _lib.lib()
fn = L.mpa_debug_grid_stats
fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 24)()
for i in range(5):
    trainer.train_step(batch, i)
torch.cuda.synchronize()
fn(buf, 1)
trainer.train_step(batch, 5)
torch.cuda.synchronize()
fn(buf, 1)
items, lanes, calls, cand, lng = max(buf[0], 1), buf[1], buf[2], buf[3], buf[4]
queries = 2 * 1000 * sum(num_parts)
print(f"queries {queries}  work items {items}  active lanes/item {lanes / items:.1f} of 64  scan batches/item {calls / items:.1f}")
print(f"candidate records per item {cand / items:.0f} ({lng / max(1, cand):.0%} in long contiguous ranges)  "
      f"pair evaluations: useful {cand / items * lanes / items * items:.3e}, issued {cand * 64:.3e}")
chunks, hit, pinned = buf[16], buf[17], buf[18]
if chunks:
    print(f"scan chunks per item {chunks / items:.0f}; with a lane past the gate {hit / chunks:.1%}; pinned evaluations (wave-level "
          f"candidate slots) per chunk {pinned / chunks:.2f}")
t = list(buf)[8:16]
if t[7]:
    names = ["prologue", "item header", "seed", "rings 0-1", "outer rings", "pads + store"]
    print("shader-clock ticks per wave: " + ", ".join(f"{n} {t[i] / t[7]:.0f}" for i, n in enumerate(names))
          + f"; whole wave {t[6] / t[7]:.0f}; waves {t[7]}")
