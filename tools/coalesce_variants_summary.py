"""Table of tools/coalesce_variants.sh: L2 hits / misses of the gather-sort kernel per launch under every variant."""
import csv
import os
import sys
from collections import defaultdict

out = sys.argv[1]
times = {}
if os.path.exists(os.path.join(out, "times.txt")):
    for line in open(os.path.join(out, "times.txt")):
        p = line.split()
        if len(p) > 4 and p[1] == "whole":
            times[p[0]] = float(p[4])
print("| variant | whole Connect call, µs | gather-sort launches seen | TCC_HIT per launch | TCC_MISS per launch | hit rate |")
print("|---|---|---|---|---|---|")
for v in ("full", "col32", "full_diag", "col32_diag", "no_table", "dummy4", "dummy2", "dummy1", "nt_table", "no_edges"):
    path = os.path.join(out, f"tcc_{v}.csv")
    if not os.path.exists(path):
        print(f"| {v} | {times.get(v, float('nan')):.1f} | (no counter file) | | | |")
        continue
    per = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(path)):
        if "cr_gather_sort_kernel" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    ds = sorted(per, key=int)[-5:]  # the last five launches (warm)
    if not ds:
        print(f"| {v} | {times.get(v, float('nan')):.1f} | 0 | | | |")
        continue
    h = sum(per[d]["TCC_HIT_sum"] for d in ds) / len(ds)
    m = sum(per[d]["TCC_MISS_sum"] for d in ds) / len(ds)
    print(f"| {v} | {times.get(v, float('nan')):.1f} | {len(per)} | {h:,.0f} | {m:,.0f} | {h / max(h + m, 1):.3f} |")
