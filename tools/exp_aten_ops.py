"""Which library (aten) operators does one training step still run, and from where?  A TorchDispatchMode over a few
eager steps of a bench workload records every aten call that touches a device tensor with the innermost frame of this
repository on the python stack (backward nodes run without one: they are listed by operator only).
   python tools/exp_aten_ops.py [config=c2] [steps=2]"""
import os
import sys
import traceback
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten

import bench

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIEWS = ("view", "reshape", "expand", "select", "slice", "unsqueeze", "squeeze", "transpose", "permute", "detach", "alias",
         "as_strided", "aten.t.", "unbind", "split", "_unsafe_view", "empty", "size", "stride", "is_", "_local_scalar")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name for v in VIEWS):
            return out
        flat, _ = tree_flatten((args, kwargs, out))
        if not any(isinstance(t, torch.Tensor) and t.is_cuda for t in flat):
            return out
        shapes = [tuple(t.shape) for t in tree_flatten((args, kwargs))[0] if isinstance(t, torch.Tensor)]
        frame = f"(no repo frame: autograd) {shapes[:2]}"
        for f in reversed(traceback.extract_stack()):
            if f.filename.startswith(HERE) and "/tools/" not in f.filename:
                frame = f"{f.filename[len(HERE) + 1:]}:{f.lineno} {f.line[:70]}"
                break
        self.rows[(name, frame)] += 1
        return out


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    from multi_part_assembly_amd.pn_transformer import build_model
    from multi_part_assembly_amd.trainer import Trainer
    cfg, batch, desc, B, P = bench.workload(cfgname, 0, dev)
    batch.pop("num_parts")
    torch.manual_seed(0)
    trainer = Trainer(build_model(cfg).to(dev), cfg, use_graph=False)
    for i in range(4):
        trainer.train_step(batch, i)
    torch.cuda.synchronize()
    with Log() as log:
        for i in range(steps):
            trainer.train_step(batch, i)
    torch.cuda.synchronize()
    total = 0
    for (name, frame), n in sorted(log.rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f"{n / steps:5.1f}  {name:34s} {frame}")
        total += n
    print(f"{total / steps:5.1f}  aten calls on device tensors per step (views excluded)")


if __name__ == "__main__":
    main()
