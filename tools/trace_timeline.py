"""Print the tail of a rocprofv3 kernel trace as a timeline: start offset, duration, idle gap before each kernel."""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) - n - skip: len(rows) - skip]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:100]}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
