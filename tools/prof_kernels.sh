#!/bin/bash
# usage: tools/prof_kernels.sh <tag> <bench.py args...>: rocprofv3 kernel-trace stats of one bench workload -> gpurun_out/<tag>/
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o run -- python3 bench.py "$@" > $out/bench.json 2> $out/bench.err
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv 2>/dev/null
python3 - "$out/kernel_stats.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.2f} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} {r["Percentage"]}%')
P
