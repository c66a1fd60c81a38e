#!/usr/bin/env python3
"""The stand-alone timing of the drop-in Chamfer operator (bench.py's `chamfer_standalone` object) on its own, so that
`rocprofv3 --kernel-trace --stats -- python tools/chamfer_standalone.py` profiles nothing but mpa_chamfer_forward:
    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $REPO/tools/chamfer_standalone.py
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    print(json.dumps(bench.chamfer_standalone(torch.device("cuda", 0), reps=reps)))
