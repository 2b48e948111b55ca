#!/bin/bash
# rocprofv3 kernel statistics of tools/bench_kron.py --no-reference into gpurun_out/kron_prof/ (run on the GPU box)
out=$GRAFT_REPO_ROOT/gpurun_out/kron_prof
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_kron.py --no-reference > $out/run.out 2> $out/run.err
cd $GRAFT_REPO_ROOT
cat $out/run.out | tail -2
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-140
