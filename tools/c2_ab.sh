# same-box A/B of two builds of the library on the headline workload:  bash tools/c2_ab.sh <libA.so> <libB.so>
export TMPDIR=/tmp
for r in 1 2; do for lib in "$@"; do
TGP_HIP_LIB=$lib python bench.py --workload c2 --secondary none --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$lib', 'step', d['windows']['ms_per_step_median'], 'kernel', r['avg_launch_ms'], 'frac', r['frac'])"
done; done
