#!/bin/bash
# tools/build_variant.sh <name> <unit|all> <extra hipcc flags...>: a tuning build of the DP library with ONE unit (force_lj, force_eam,
# neighbor ...) — or `all` of them — recompiled with extra -D flags, into variants/<name>/libmmd_hip_dp.so (travels with gpurun; git-ignored).
# Run with MMD_LIB_DIR=variants/<name>. PREC=sp builds the single-precision library instead (libmmd_hip_sp.so).
#   tools/build_variant.sh profile all -DMMD_PROFILE      the profiling library: mmd_set_option("ablate") switches phases of the hot kernels off
set -e
cd "$(dirname "$0")/../minimd_amd/csrc"
name=$1; unit=$2; shift 2
out=../../variants/$name
mkdir -p $out
P=${PREC:-dp}; PN=2; [ $P = sp ] && PN=1
make -j8 $P > /dev/null
units="util atom neighbor integrate comm api force_lj force_eam host sim launch"
objs=""
for u in $units; do
  if [ $u = $unit ] || [ $unit = all ]; then
    fl="-ffp-contract=off"; case $u in force_lj|force_eam) fl="-ffp-contract=fast";; esac
    src=$u.hip; lang=""; [ -f $u.cpp ] && { src=$u.cpp; lang="-x hip"; }
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-result -I/opt/rocm/include -mllvm -amdgpu-mfma-vgpr-form=1 $fl -DMMD_PRECISION=$PN "$@" $lang -c $src -o $out/$u.o &
    objs="$objs $out/$u.o"
  else objs="$objs ../build/$P/$u.o"; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $out/libmmd_hip_$P.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $out/libmmd_hip_$P.so ($unit: $*)"
