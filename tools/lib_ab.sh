# same-box A/B of two builds of the library on bench.py workloads:  bash tools/lib_ab.sh "<workloads>" <libA.so> <libB.so> ...
export TMPDIR=/tmp
wls=$1; shift
for wl in $wls; do for r in 1 2; do for lib in "$@"; do
TGP_HIP_LIB=$lib python bench.py --workload $wl --secondary none --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$wl', '$lib'.split('/')[-1], 'step', d['windows']['ms_per_step_median'], 'kernel', r.get('avg_launch_ms'), 'frac', r['frac'])"
done; done; done
