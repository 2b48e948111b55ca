#!/bin/bash
# NDPSelect chip-wide LOBPCG on one graph of 3000 / 20000 / 1M nodes: fused steps (default) against the seven-launch step.
for n in 3000 20000 1000000; do
  echo "== n=$n fused"; python tools/bench_ndp_large.py $n 2>&1 | tail -2
  echo "== n=$n classic"; TGP_NDP_LARGE_CLASSIC=1 python tools/bench_ndp_large.py $n 2>&1 | tail -2
done
