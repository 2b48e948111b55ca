#!/bin/bash
# Copy the summaries of a tools/collect_profiles.sh run (gpurun_out/<tag>/) into profiles/ under the round's prefix.
# usage: tools/install_profiles.sh <tag> <prefix>      e.g.  tools/install_profiles.sh r06e r06
src=gpurun_out/$1; pre=$2
for f in bench_kernel_stats.csv bench_headline_kernel_stats.csv c2_f64_kernel_stats.csv c4_graclus_sorted_kernel_stats.csv c4_graclus_unsorted_kernel_stats.csv topk_connect_kernel_stats.csv kron_kernel_stats.csv pmc_summary.md e2e_fresh_batch.txt e2e_launches.txt e2e_poolers.txt e2e_train_steps.txt fp64_dense.txt fresh_profile_kernels.txt graclus_hubs.txt kron.txt kron_timeline.txt ndp_large.txt ndp_mid_ab.txt ndp_small.txt reference_harness.txt train_step_sequence.txt mlp_select_bwd.txt coalesce_f64.txt sparse_train_steps.txt select_fold.txt train_step_c2.txt unbatched_forward.txt host_time_fresh.txt coalesce_relabel_cost.txt bench_default_time.txt rows_route_crossover.txt ndp_forward_kernels.txt; do
  [ -f $src/$f ] && cp $src/$f profiles/${pre}_$f
done
cp $src/ndp_hubs_after.txt profiles/${pre}_ndp_hubs.txt
cp $src/tcc_coalesce.md profiles/${pre}_tcc_coalesce.md
grep '^{' $src/bench_line.json | tail -1 > profiles/${pre}_bench_line.json
grep '^{' $src/bench_line.json | head -n -1 > profiles/${pre}_bench_secondaries.jsonl
grep '^{' $src/bench_default.json | tail -1 > profiles/${pre}_bench_default_line.json
cp $src/roofline_traffic.json profiles/roofline_traffic.json
sed -i "s#profiles/$1_pmc_summary.md#profiles/${pre}_pmc_summary.md#" profiles/roofline_traffic.json
cp $src/bench_c2_rccl.json profiles/${pre}_bench_c2_rccl_one_rank.json
cp $src/bench_topk_batch_rccl.json profiles/${pre}_bench_topk_batch_rccl_one_rank.json
