"""Dev helper: per-kernel durations of the LAST iteration in a rocprofv3 results database (rocpd sqlite).
    python tools/prof_table.py gpurun_out/<dir>/prof/<name>_results.db [marker-substring] [min_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "dg_prepare"
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = [dict(zip(cols, r)) for r in cur.execute("select * from kernels order by start")]
starts = [i for i, d in enumerate(rows) if marker in d["name"]]
seg = rows[starts[-1]:] if starts else rows
total = 0.0
agg = {}
for d in seg:
    us = (d["end"] - d["start"]) / 1e3
    total += us
    key = d["name"].split("(")[0][-48:]
    agg[key] = agg.get(key, 0.0) + us
    if us >= min_us:
        print(f"{d['name'][:64]:64s} {us:8.0f} us  grid {d['grid_x']}x{d['grid_y']}x{d['grid_z']}")
print(f"--- last iteration: {len(seg)} launches, {total / 1e3:.2f} ms of kernel time")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:14]:
    print(f"   {k:48s} {v:8.0f} us")
