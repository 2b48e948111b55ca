#!/bin/bash
# rocprofv3 kernel statistics of one tool run into gpurun_out/prof_tool/: tools/prof_tool.sh <script.py> [args...]
out=$GRAFT_REPO_ROOT/gpurun_out/prof_tool
rm -rf $out; mkdir -p $out
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 $script "$@" > $out/run.out 2> $out/run.err
cd $GRAFT_REPO_ROOT
tail -3 $out/run.out
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -${TOPN:-14} $f | cut -c1-150
