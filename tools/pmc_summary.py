"""Summarise the --pmc passes collected by tools/collect_profiles.sh: per kernel, average FETCH_SIZE / WRITE_SIZE per
launch, in bytes, with the gfx950 correction of MI355X_MICROARCH.md ("HBM": FETCH_SIZE reports half the bytes of wide
coalesced streaming reads: doubled here; WRITE_SIZE is exact for 16-byte-per-lane streaming stores; other widths are
uncalibrated, so totals of gather-heavy kernels are upper bounds)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
res = defaultdict(lambda: defaultdict(dict))
for path in sorted(glob.glob(os.path.join(root, "pmc_*_*.csv"))):
    base = os.path.basename(path)[4:-4]
    work, ctr = base.rsplit("_", 2)[0], "_".join(base.rsplit("_", 2)[1:])
    per = defaultdict(list)
    rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == ctr]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    # tools/run_kernel.py launches a marker (entropy_partial_kernel) right before the measured repetitions: only the
    # dispatches behind it belong to the measured call (r2 counted GraclusSelect's set-up kernels into the Connect)
    marks = [i for i, r in enumerate(rows) if "entropy_partial_kernel" in r["Kernel_Name"]]
    if marks:
        rows = [r for r in rows[marks[-1] + 1:] if "final_sum_kernel" not in r["Kernel_Name"]]
    for row in rows:
        per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    for k, v in per.items():
        res[work][k][ctr] = (sum(v) / len(v), len(v))
print("| workload | kernel | launches | FETCH_SIZE KB (raw) | read MB (x2 rule) | WRITE_SIZE KB | write MB | total MB |")
print("|---|---|---|---|---|---|---|---|")
for work, ks in res.items():
    tot = 0.0
    ks = {k: c for k, c in ks.items() if "tgp::" in k}  # the workload's own kernels (not the set-up's torch kernels)
    reps = max((max(c.get("FETCH_SIZE", (0, 0))[1], c.get("WRITE_SIZE", (0, 0))[1]) for c in ks.values()), default=1)
    for k, c in sorted(ks.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[0] + kv[1].get("WRITE_SIZE", (0, 0))[0])):
        f, nf = c.get("FETCH_SIZE", (0.0, 0))
        w, nw = c.get("WRITE_SIZE", (0.0, 0))
        rd, wr = f * 1024 * 2 / 1e6, w * 1024 / 1e6
        if rd + wr < 0.5:
            continue
        calls = max(nf, nw)
        if calls >= 4:  # launched by every repetition of the measured call (set-up kernels of the tool run once)
            tot += (rd + wr) * (calls // 4)
        print(f"| {work} | `{k[:70]}` | {max(nf, nw)} | {f:.1f} | {rd:.1f} | {w:.1f} | {wr:.1f} | {rd + wr:.1f} |")
    print(f"| {work} | **kernels of one measured call (launch counts / 4 repetitions)** | | | | | | **{tot:.1f}** |")
