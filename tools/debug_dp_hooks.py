import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
dist.init_process_group("gloo", rank=0, world_size=1)
import test_dp_gpu as T
from multi_part_assembly_amd import dp
from multi_part_assembly_amd.trainer import Trainer
dev = torch.device("cuda", 0)
model, cfg = T._small_model(); model.to(dev)
orig_init = dp.BucketedGradReducer.__init__
tr = Trainer(model, cfg)
# pretend world = 2: hooks active
tr.reducer.world = 2
for p in tr.flat.params: p.register_post_accumulate_grad_hook(tr.reducer._on_grad)
from multi_part_assembly_amd.gradsink import GradSink
tr.sink = GradSink(on_ready=tr.reducer._on_grad)
calls = []
real = dist.all_reduce
def fake(t, *a, **k):
    calls.append(t.numel()); return real(t, *a, **k)
dist.all_reduce = fake; dp.dist.all_reduce = fake
fin = tr.reducer.finish
def finish():
    print("at finish:", [(b["ready"], b["count"]) for b in tr.reducer.buckets], "calls so far", calls)
    return fin()
tr.reducer.finish = finish
tr.train_step(T._shard(0, dev), 0)
print("all_reduce calls:", calls)
