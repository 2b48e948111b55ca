# a build of the library with other GEMM schedule constants, for same-box A/B (tools/c2_ab.sh):
#   bash tools/build_gemm_variant.sh NAME -DTGP_LOAD_AT=2 -DTGP_STORE_AT=10   ->  torch-geometric-pool_amd/lib/libtgp_hip_NAME.so
set -e
cd "$(dirname "$0")/../torch-geometric-pool_amd/csrc"
name=$1; shift
mkdir -p ../lib/obj_$name
for f in dense.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function "$@" -x hip -c $f -o ../lib/obj_$name/$f.o &
done
wait
objs=""
for o in ../lib/obj/*.o; do b=$(basename $o); if [ -f ../lib/obj_$name/$b ]; then objs="$objs ../lib/obj_$name/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtgp_hip_$name.so $objs
echo built ../lib/libtgp_hip_$name.so
