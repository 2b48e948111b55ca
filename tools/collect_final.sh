#!/bin/bash
export TMPDIR=/tmp
out=$PWD/gpurun_out/r02f
mkdir -p $out
stats() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o run -- "$@" > $out/$name.out 2> $out/$name.err; cp $(find $out/$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv 2>/dev/null; }
pmc() { name=$1; ctr=$2; shift; shift; rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_${name}_$ctr -o run -- "$@" > /dev/null 2> $out/pmc_${name}_$ctr.err; cp $(find $out/pmc_${name}_$ctr -name "*counter_collection.csv" | head -1) $out/pmc_${name}_$ctr.csv 2>/dev/null; }
stats bench python3 bench.py --steps 20 --warmup 5
stats c4_graclus_sorted python3 bench.py --workload c4_graclus --secondary none --no-cpu-baseline --steps 50
stats c4_graclus_unsorted python3 bench.py --workload c4_graclus --unsorted-edges --secondary none --no-cpu-baseline --steps 50
stats topk_connect python3 bench.py --workload topk_connect --secondary none --no-cpu-baseline --steps 50
for k in coalesce_c4_sorted subgraph_topk c3 reduce_topk gemm_c2; do
  pmc $k FETCH_SIZE python3 tools/run_kernel.py $k 4
  pmc $k WRITE_SIZE python3 tools/run_kernel.py $k 4
done
python3 tools/pmc_summary.py $out > $out/pmc_summary.md
grep "kernels of one measured call" $out/pmc_summary.md
