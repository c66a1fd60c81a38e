import sys, time, torch
sys.path.insert(0, "/root/repo")
from multi_part_assembly_amd import config, synthetic
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.trainer import Trainer
dev = torch.device("cuda:0")
cfg = config.pn_transformer_everyday()
torch.manual_seed(0)
model = build_model(cfg).to(dev)
trainer = Trainer(model, cfg, use_graph=False)
batch = synthetic.make_batch(32, 20, 1000, preset="everyday", seed=1234, device=dev); batch.pop("num_parts")
for i in range(5): trainer.train_step(batch, i)
torch.cuda.synchronize()
# host-only time: few steps so that the launch queue never fills
for n in (1, 2, 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): trainer.train_step(batch, i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{n} steps: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, until done {1e3 * (t2 - t0) / n:.3f} ms/step")
