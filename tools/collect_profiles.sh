#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into profiles/).
# usage: tools/collect_profiles.sh <tag>
tag=${1:-r06}
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
stats() {  # name, command...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o run -- "$@" > $out/$name.out 2> $out/$name.err
  cp $(find $out/$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv 2>/dev/null
}
pmc() {  # name, counter, command...
  name=$1; ctr=$2; shift; shift
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_${name}_$ctr -o run -- "$@" > /dev/null 2> $out/pmc_${name}_$ctr.err
  cp $(find $out/pmc_${name}_$ctr -name "*counter_collection.csv" | head -1) $out/pmc_${name}_$ctr.csv 2>/dev/null
}
( time python3 bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default_time.txt
python3 bench.py --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err
TGP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29617 python3 bench.py --workload topk_batch --secondary none --no-cpu-baseline --steps 50 2> $out/bench_topk_batch_rccl.err | grep '^{' > $out/bench_topk_batch_rccl.json
TGP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29618 python3 bench.py --workload c2 --secondary none --no-cpu-baseline --steps 50 2> $out/bench_c2_rccl.err | grep '^{' > $out/bench_c2_rccl.json
python3 tools/bench_graclus_hubs.py 2>&1 | grep -v -i 'warn\|amdgpu.ids' > $out/graclus_hubs.txt
python3 tools/bench_ndp_hubs.py 2>&1 | grep -v -i 'warn\|amdgpu.ids' > $out/ndp_hubs_after.txt
stats bench python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
# the headline alone: the secondaries (training steps) launch the same GEMM instantiation on other shapes, so only this
# run's average for gemm_f32_mfma_kernel<false,true,128,64,0,2> is the headline kernel's duration
stats bench_headline python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary none
stats c4_graclus_sorted python3 bench.py --workload c4_graclus --secondary none --no-cpu-baseline --steps 50
stats c4_graclus_unsorted python3 bench.py --workload c4_graclus --unsorted-edges --secondary none --no-cpu-baseline --steps 50
stats topk_connect python3 bench.py --workload topk_connect --secondary none --no-cpu-baseline --steps 50
stats c2_f64 python3 bench.py --workload c2_f64 --secondary none --no-cpu-baseline --steps 20
python3 tools/bench_f64.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $out/fp64_dense.txt
python3 tools/fresh_profile.py topk_c3 graclus_c3 mincut_c3 2>&1 | grep -E "^==|us  |GPU-busy" > $out/fresh_profile_kernels.txt
stats kron python3 tools/bench_kron.py --no-reference
python3 tools/bench_kron.py > $out/kron.txt 2>&1
python3 tools/e2e_launches.py > $out/e2e_launches.txt 2>&1
python3 tools/train_step_sequence.py 2>&1 | grep -v -i 'warn\|amdgpu.ids' | cut -c1-170 > $out/train_step_sequence.txt
python3 tools/bench_mlp_select_bwd.py 2>&1 | grep -v -i 'warn\|amdgpu.ids' > $out/mlp_select_bwd.txt
python3 tools/bench_coalesce_f64.py 2>&1 | grep -v -i 'warn\|amdgpu.ids' > $out/coalesce_f64.txt
python3 tools/e2e_fresh_batch.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $out/e2e_fresh_batch.txt
python3 tools/bench_poolers_e2e.py > $out/e2e_poolers.txt 2>&1
python3 tools/bench_ndp_large.py > $out/ndp_large.txt 2>&1
python3 tools/e2e_train_step.py mincut_c3 diff_c3 mincut_c2 diff_c2 mincut_med diff_med topk_c3 graclus_c3 --top 8 2>&1 | grep -v -i "warn" > $out/e2e_train_steps.txt
# r6: the C2-sized training step, launch by launch, with bench.py's loss: the rows route (default for sparse inputs) and
# the densifying route (TGP_ROWS_ROUTE=0, functions._PoolLargeFn), and the r5 operator-by-operator graph (TGP_FOLD_TRAINING=0)
(echo "== default: un-padded rows route (functions._PoolUnbatchedFn)"; python3 tools/e2e_train_step.py mincut_c2 diff_c2 --sequence --bench-loss 2>&1 | grep -v -i "amdgpu.ids\|warn" | cut -c1-170; echo; echo "== TGP_ROWS_ROUTE=0: densifying route (functions._PoolLargeFn)"; TGP_ROWS_ROUTE=0 python3 tools/e2e_train_step.py mincut_c2 diff_c2 --sequence --bench-loss 2>&1 | grep -v -i "amdgpu.ids\|warn" | cut -c1-170; echo; echo "== TGP_ROWS_ROUTE=0 TGP_FOLD_TRAINING=0: operator by operator (r5)"; TGP_ROWS_ROUTE=0 TGP_FOLD_TRAINING=0 python3 tools/e2e_train_step.py mincut_c2 diff_c2 --bench-loss 2>&1 | grep -v -i "amdgpu.ids\|warn" | cut -c1-170) > $out/train_step_c2.txt
# r6: host time of the batched sparse poolers on NEW tensor objects (verdict r5 item 5), and the relabel pass a look-up-free
# coalesce Connect would need (item 4)
(for w in topk_batch_fresh graclus_batch_fresh e2e:topk_c3 e2e:graclus_c3 e2e:diff_c3; do python3 tools/host_profile_fresh.py $w 2>&1 | grep -v -i "amdgpu.ids\|warn" | head -16; echo; done; echo "== TGP_SPS_ARENA=0 (four allocations of their own + the compaction wrapper: r5 form)"; for w in topk_batch_fresh graclus_batch_fresh; do TGP_SPS_ARENA=0 python3 tools/host_profile_fresh.py $w 2>&1 | grep -v -i "amdgpu.ids\|warn" | head -1; done) > $out/host_time_fresh.txt
python3 tools/coalesce_relabel_cost.py 2>&1 | grep -v -i "amdgpu.ids\|warn" > $out/coalesce_relabel_cost.txt
python3 tools/rows_route_crossover.py 2>&1 | grep -v -i "amdgpu.ids\|warn" > $out/rows_route_crossover.txt
python3 tools/profile_unbatched.py 2>&1 | grep -v -i "amdgpu.ids\|warn" | cut -c1-150 > $out/unbatched_forward.txt
(echo "== TGP_FOLD_TRAINING=0 (operator-by-operator graph, staged Reduce + Connect: r4 form)"; TGP_FOLD_TRAINING=0 python3 tools/e2e_train_step.py topk_c3 graclus_c3 --sequence 2>&1 | grep -v -i "amdgpu.ids\|warn" | cut -c1-180; echo; echo "== default (r5, late)"; python3 tools/e2e_train_step.py topk_c3 graclus_c3 --sequence 2>&1 | grep -v -i "amdgpu.ids\|warn" | cut -c1-180) > $out/sparse_train_steps.txt
python3 tools/bench_select_fold.py 2>&1 | grep select > $out/select_fold.txt
python3 tools/ndp_small_ab.py 2>&1 | grep -v -i "warn" > $out/ndp_small.txt
python3 tools/ndp_launch_list.py 2>&1 | grep -E " us  |GPU-busy" | cut -c1-150 > $out/ndp_forward_kernels.txt
bash tools/ndp_large_ab.sh 2>&1 | grep -v -i "warn\|amdgpu.ids" > $out/ndp_mid_ab.txt
python3 tools/bench_reference_harness.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $out/reference_harness.txt
python3 tools/kron_timeline.py $(find $out/kron -name "*kernel_trace.csv" | head -1) > $out/kron_timeline.txt 2>&1
for k in coalesce_c4_sorted subgraph_topk c3 reduce_topk reduce_ndp reduce_graclus topk_batch graclus_batch gemm_c2; do
  pmc $k FETCH_SIZE python3 tools/run_kernel.py $k 4
  pmc $k WRITE_SIZE python3 tools/run_kernel.py $k 4
done
# r5 (verdict r4 item 5): L2 hit / miss of the coalesce Connect's kernels (the cluster-table look-ups of cr_gather_sort_kernel)
pmc coalesce_c4_sorted TCC_HIT_sum python3 tools/run_kernel.py coalesce_c4_sorted 4
pmc coalesce_c4_sorted TCC_MISS_sum python3 tools/run_kernel.py coalesce_c4_sorted 4
cp $out/pmc_coalesce_c4_sorted_TCC_HIT_sum.csv $out/tcc_hit_coalesce.csv 2>/dev/null
cp $out/pmc_coalesce_c4_sorted_TCC_MISS_sum.csv $out/tcc_miss_coalesce.csv 2>/dev/null
rm -f $out/pmc_coalesce_c4_sorted_TCC_*.csv  # (pmc_summary.py reads every pmc_*_*.csv as a FETCH / WRITE pass)
python3 tools/tcc_summary.py $out/tcc_hit_coalesce.csv $out/tcc_miss_coalesce.csv > $out/tcc_coalesce.md 2>&1
python3 tools/pmc_summary.py $out --json $out/roofline_traffic.json --source profiles/${tag}_pmc_summary.md > $out/pmc_summary.md
rm -rf $out/pmc_*_FETCH_SIZE $out/pmc_*_WRITE_SIZE $out/pmc_*_TCC_HIT_sum $out/pmc_*_TCC_MISS_sum $out/bench $out/bench_headline $out/c4_graclus_sorted $out/c4_graclus_unsorted $out/topk_connect $out/kron
ls $out | head -60
