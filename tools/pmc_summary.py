"""Summarise a rocprofv3 --pmc counter_collection.csv: average counter value per launch and kernel.
Usage: python tools/pmc_summary.py <counter_collection.csv> [--top 40]"""
import argparse, collections, csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(a.csv)):
    key = (r["Counter_Name"], r["Kernel_Name"])
    agg[key][0] += float(r["Counter_Value"])
    agg[key][1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])[: a.top]
for (counter, kernel), (total, n) in rows:
    print(f"{counter:<12} avg/launch {total / n:14.1f}  launches {n:4d}  {kernel[:110]}")
