#!/bin/bash
# Experiment builds that never touch the product library:  tools/build_variant.sh <name> "<shape ...|all>" [extra hipcc flags...]
#   e.g. tools/build_variant.sh TI 16_3_1 -DM4Q_TWO_INDEX_COMPLEX=1   -> tools/bin/libTI.so
# Recompiles the named shapes' kernel objects with the extra flags into tools/bin/obj_<name>/ and links them with the product's
# other objects (mpc4quantum_amd/csrc/build/, which must be up to date).  Run a variant on the GPU box with
#   M4Q_LIB=tools/bin/lib<name>.so python ...     (mpc4quantum_amd/_lib.py honours M4Q_LIB)
set -e
name=$1; shapes=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/mpc4quantum_amd/csrc
out=$root/tools/bin/obj_$name
mkdir -p $out
all=$(sed -n 's/^M4Q_SHAPE(\([0-9]*\), *\([0-9]*\), *\([0-9]*\)).*/\1_\2_\3/p' $src/m4q_shapes.inc)
[ "$shapes" = all ] && shapes=$all
objs="$src/build/capi.o"
pids=""
for s in $all; do
  if echo " $shapes " | grep -q " $s "; then
    IFS=_ read nx nu ord <<< "$s"
    po=""; grep -q "^M4Q_SHAPE($nx, *$nu, *$ord).*plant-only" $src/m4q_shapes.inc && po="-DM4Q_PLANT_ONLY"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-command-line-argument \
        -DM4Q_DEV -DM4Q_NX=$nx -DM4Q_NU=$nu -DM4Q_ORDER=$ord $po "$@" -c $src/m4q_kernels.hip -o $out/kernels_$s.o &
    pids="$pids $!"
    objs="$objs $out/kernels_$s.o"
  else
    objs="$objs $src/build/kernels_$s.o"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/bin/lib$name.so $objs
echo $root/tools/bin/lib$name.so
