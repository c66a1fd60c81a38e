"""Which Python call sites create zero/fill/copy launches in one eager training step (monkeypatched torch entry points;
what autograd does internally is the difference to the profiler's count, tools/probe_glue.py)."""
import collections
import sys
import traceback
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
for i in range(3):
    trainer.train_step(batch, i)
torch.cuda.synchronize()
hits = collections.Counter()


def wrap(owner, name):
    orig = getattr(owner, name)

    def fn(*a, **k):
        fr = [f for f in traceback.extract_stack()[:-1] if "multi_part_assembly_amd" in f.filename]
        hits[(name, f"{Path(fr[-1].filename).name}:{fr[-1].lineno}" if fr else "?")] += 1
        return orig(*a, **k)

    setattr(owner, name, fn)


for owner, names in ((torch, ("zeros", "zeros_like", "ones_like", "full", "ones", "empty_like", "clone", "cat", "stack", "where")),
                     (torch.Tensor, ("zero_", "fill_", "new_zeros", "clone", "copy_", "contiguous", "float", "to", "detach"))):
    for n in names:
        wrap(owner, n)
trainer.train_step(batch, 3)
torch.cuda.synchronize()
for (name, site), c in sorted(hits.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {name:12s} {site}")
