"""L2 hit / miss per kernel of one measured call from two rocprofv3 --pmc passes (TCC_HIT_sum, TCC_MISS_sum) over
tools/run_kernel.py: only the dispatches behind the tool's marker kernel count.  usage: tcc_summary.py hit.csv miss.csv"""
import csv
import sys
from collections import defaultdict


def load(path, ctr):
    rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == ctr]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    marks = [i for i, r in enumerate(rows) if "entropy_partial_kernel" in r["Kernel_Name"]]
    if marks:
        rows = [r for r in rows[marks[-1] + 1:] if "final_sum_kernel" not in r["Kernel_Name"]]
    per = defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


hit, miss = load(sys.argv[1], "TCC_HIT_sum"), load(sys.argv[2], "TCC_MISS_sum")
print("| kernel | launches | TCC_HIT (avg per launch) | TCC_MISS (avg per launch) | hit rate |")
print("|---|---|---|---|---|")
for k in sorted(set(hit) | set(miss), key=lambda k: -(sum(miss.get(k, [0])) + sum(hit.get(k, [0])))):
    if "tgp::" not in k:
        continue
    h, m = hit.get(k, [0.0]), miss.get(k, [0.0])
    ha, ma = sum(h) / len(h), sum(m) / len(m)
    if ha + ma < 1000:
        continue
    print(f"| `{k[:80]}` | {max(len(h), len(m))} | {ha:,.0f} | {ma:,.0f} | {ha / (ha + ma):.3f} |")
