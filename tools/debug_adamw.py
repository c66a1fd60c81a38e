"""Debug helper: FusedAdam (mask / clip variants) against torch.optim.AdamW step by step."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from multi_part_assembly_amd.optim import FlatBuffers, FusedAdam, decay_mask_for

dev = torch.device("cuda:0")


def net():
    torch.manual_seed(4)
    return torch.nn.Sequential(torch.nn.Linear(7, 9), torch.nn.LayerNorm(9), torch.nn.Linear(9, 5),
                               torch.nn.BatchNorm1d(5)).to(dev)


for use_mask, clip in ((False, None), (True, None), (False, 0.05), (True, 0.05)):
    mine, theirs = net(), net()
    flat = FlatBuffers(list(mine.parameters()))
    mask = decay_mask_for(mine, flat) if use_mask else None
    opt = FusedAdam(flat, lr=1e-2, weight_decay=0.1, decay_mask=mask, clip_grad=clip)
    nd = [theirs[0].bias, theirs[1].weight, theirs[1].bias, theirs[2].bias, theirs[3].weight, theirs[3].bias]
    dc = [theirs[0].weight, theirs[2].weight]
    topt = torch.optim.AdamW([{"params": nd, "weight_decay": 0.0 if use_mask else 0.1}, {"params": dc, "weight_decay": 0.1}], lr=1e-2)
    torch.manual_seed(3)
    for step in range(4):
        x = torch.randn(16, 7, device=dev) * (3.0 if step % 2 else 0.01)
        opt.zero_grad(); topt.zero_grad()
        mine(x).square().sum().backward(); theirs(x).square().sum().backward()
        gd = max(float((a.grad - b.grad).abs().max()) for a, b in zip(mine.parameters(), theirs.parameters()))
        if clip:
            tn = torch.nn.utils.clip_grad_norm_(theirs.parameters(), clip)
        opt.step(); topt.step()
        torch.cuda.synchronize()
        errs = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(mine.parameters(), theirs.parameters())]
        print(use_mask, clip, step, "graddiff", gd, "hyper", opt._hyper_dev.tolist()[:7], "tn", float(tn) if clip else None,
              ["%.1e" % e for e in errs])
