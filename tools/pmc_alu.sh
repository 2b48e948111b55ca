#!/bin/bash
# usage: tools/pmc_alu.sh <run_kernel.py case> <kernel regex>: instruction-mix / busy counters of one kernel -> gpurun_out/pmc_alu/
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_alu
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-include-regex "$2" --output-format csv -d $out/p$i -o run -- python3 tools/run_kernel.py $1 4 > /dev/null 2> $out/err$i.txt
done
python3 - $out <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:50], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:50s} {c:24s} per-launch {v / max(n,1):16.1f}  (rows {n})")
P
