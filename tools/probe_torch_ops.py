"""Dev probe (GPU): census of the torch library ops (aten::*) launched by one eager training step of a config — what is
left of the step outside libmpa_hip.so.    python tools/probe_torch_ops.py [c3|c5|c2]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from multi_part_assembly_amd import config, synthetic
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.trainer import Trainer

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda:0")
cfg = {"c3": config.dgl_dgcnn_everyday, "c5": config.rgl_net_dgcnn_artifact, "c2": config.pn_transformer_everyday}[which]()
torch.manual_seed(0)
model = build_model(cfg).to(dev)
tr = Trainer(model, cfg, use_graph=False)
batch = synthetic.make_batch(32, 20, 1000, preset="artifact" if which == "c5" else "everyday", seed=1234, device=dev)
batch.pop("num_parts")
for i in range(3):
    tr.train_step(batch, i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.train_step(batch, 3)
torch.cuda.synchronize()
skip = ("aten::view", "aten::reshape", "aten::expand", "aten::select", "aten::slice", "aten::as_strided", "aten::t",
        "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::empty", "aten::to",
        "aten::_unsafe_view", "aten::empty_like", "aten::empty_strided", "aten::contiguous", "aten::permute", "aten::flatten",
        "aten::unflatten", "aten::result_type", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh",
        "aten::view_as", "aten::expand_as", "aten::narrow", "aten::unbind", "aten::split", "aten::chunk", "aten::size")
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name not in skip:
        cnt[(e.name, str(e.input_shapes)[:70])] += 1
tot = sum(cnt.values())
print(f"{which}: {tot} aten ops (views excluded; nested ops are counted with their parents)")
for (k, v) in sorted(cnt.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{v:4d} {k[0]:28s} {k[1]}")
