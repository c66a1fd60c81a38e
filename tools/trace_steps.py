"""Per-step kernel breakdown from a rocprofv3 kernel_trace.csv: steps are delimited by the dispatches
of a marker kernel (default: the fused Adam step, once per training step); reports the last K steps.
Usage: python tools/trace_steps.py trace.csv [--marker adam_kernel] [--last 3] [--top 40]"""
import argparse, csv, collections

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", default="adam_kernel")
ap.add_argument("--last", type=int, default=3)
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--gaps", type=float, default=0.0, help="list idle gaps longer than this many us in the last step")
ap.add_argument("--seq", default=None, help="also list, in launch order, the last step's kernels whose name contains this")
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
assert len(marks) > a.last, f"only {len(marks)} marker dispatches"
lo, hi = marks[-a.last - 1] + 1, marks[-1] + 1
win = rows[lo:hi]
t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in win:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"]][0] += d
    agg[r["Kernel_Name"]][1] += 1
    busy += d
n = a.last
print(f"steps={n} wall/step={(t1 - t0) / n / 1e6:.3f} ms  kernel-busy/step={busy / n / 1e6:.3f} ms  "
      f"dispatches/step={len(win) / n:.0f}")
print(f"{'ms/step':>9} {'calls/step':>10} {'avg us':>9}  kernel")
for name, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[: a.top]:
    print(f"{d / n / 1e6:9.3f} {c / n:10.1f} {d / c / 1e3:9.1f}  {name[:120]}")
if a.seq is not None:
    last = rows[marks[-2] + 1: marks[-1] + 1]
    print(f"-- last step, launch order, kernels matching {a.seq!r} (start offset us, duration us, grid, block)")
    for r in last:
        if a.seq in r["Kernel_Name"]:
            print(f"{(int(r['Start_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3:10.1f} "
                  f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  "
                  f"{r.get('Grid_Size_X', r.get('Grid_Size', '?'))} {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}  {r['Kernel_Name'][:60]}")
if a.gaps > 0:
    last = rows[marks[-2] + 1: marks[-1] + 1]
    print(f"-- last step: idle gaps > {a.gaps} us (gap us, previous kernel -> next kernel)")
    for x, y in zip(last, last[1:]):
        gap = (int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3
        if gap > a.gaps:
            print(f"{gap:9.1f}  {x['Kernel_Name'][:70]}  ->  {y['Kernel_Name'][:70]}")
