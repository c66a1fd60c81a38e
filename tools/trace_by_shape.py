"""Per-launch-shape durations of selected kernels in a rocprofv3 kernel trace (csv): one line per (kernel, grid, block).
   python tools/trace_by_shape.py <kernel_trace.csv> <name substring> [<name substring> ...] [--skip-frac 0.5]
The first `skip-frac` of every shape's launches (warm-up steps) is left out."""
import csv
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = 0.5
    for i, a in enumerate(sys.argv):
        if a == "--skip-frac":
            skip = float(sys.argv[i + 1])
            args.remove(sys.argv[i + 1])
    path, subs = args[0], args[1:]
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if subs and not any(s in name for s in subs):
                continue
            grid = tuple(int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            wg = tuple(int(r[k]) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
            blocks = tuple(g // w for g, w in zip(grid, wg))
            rows[(name[:70], blocks, wg[0])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out = []
    for key, v in rows.items():
        v.sort()
        v = v[int(len(v) * skip):]
        d = [(e - s) / 1e3 for s, e in v]
        out.append((sum(d), key, len(d), sum(d) / len(d), min(d), max(d)))
    out.sort(reverse=True)
    print(f"{'total us':>10} {'calls':>6} {'avg us':>8} {'min':>7} {'max':>7}  blocks x threads  kernel")
    for tot, (name, blocks, wg), n, avg, lo, hi in out:
        print(f"{tot:10.1f} {n:6d} {avg:8.1f} {lo:7.1f} {hi:7.1f}  {blocks} x {wg}  {name}")


if __name__ == "__main__":
    main()
