#!/bin/bash
# TCC_HIT / TCC_MISS of cr_gather_sort_kernel under each variant of tools/coalesce_variants.py + the call's time.
# usage: bash tools/coalesce_variants.sh [outdir]   (needs the diagnostic build: make -C torch-geometric-pool_amd/csrc stamps)
out=${1:-gpurun_out/coalesce_variants}
mkdir -p $out
export TMPDIR=/tmp
for v in full col32 full_diag col32_diag no_table dummy4 dummy2 dummy1 nt_table no_edges; do
  python3 tools/coalesce_variants.py $v --time >> $out/times.txt 2>/dev/null
  rm -rf $out/pmc_$v
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/pmc_$v -o p -- python3 tools/coalesce_variants.py $v > $out/pmc_$v.log 2>&1
  f=$(find $out/pmc_$v -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $out/tcc_$v.csv
  rm -rf $out/pmc_$v
done
python3 tools/coalesce_variants_summary.py $out > $out/summary.md
cat $out/summary.md
