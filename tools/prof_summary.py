"""Summarise a rocprofv3 kernel_stats.csv: top kernels by total time (name shortened)."""
import csv
import glob
import sys

for d in sys.argv[1:]:
    fs = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)
    if not fs:
        print(d, "no stats")
        continue
    rows = list(csv.DictReader(open(fs[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"== {d}: {tot / 1e6:.2f} ms GPU kernel time, {sum(int(r['Calls']) for r in rows)} launches")
    for r in rows[:22]:
        print(f"  {r['Name'][:86]:86s} {r['Calls']:>5s} x {float(r['AverageNs']) / 1e3:8.1f} us = {float(r['Percentage']):5.1f}%")
