"""Timeline of the last KronConnect call in a rocprofv3 kernel trace (gpurun_out/kron_prof/run_kernel_trace.csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/kron_prof/run_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'kron_flags' in n][-1]
t0 = int(rows[idx]['Start_Timestamp'])
prev_end = None
for r in rows[idx:]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.2f}  gap {gap:6.2f}  q{r.get('Queue_Id','')} {r['Kernel_Name'][:48]}")
    prev_end = max(prev_end or 0, e)
