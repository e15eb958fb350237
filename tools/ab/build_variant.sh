#!/bin/bash
# Builds an A/B variant of the library:   tools/ab/build_variant.sh <name> "<extra hipcc flags>" [file.hip ...]
# The listed sources (default: so3x_diffusion.hip) are compiled with -DSO3X_AB_BUILD and the extra flags, everything else
# comes from the in-tree objects; result: build/libso3x_<name>.so (git-ignored; travels to the GPU box with gpurun).
set -e
name=$1; flags=$2; shift 2 || true
files=${@:-so3x_diffusion.hip}
root=$(cd "$(dirname "$0")/../.." && pwd)
csrc=$root/diffusion-extensions_amd/csrc
mkdir -p $root/build/obj_$name
make -s -C $csrc -j8 >/dev/null
objs=""
skip=""
for f in $files; do
  o=$root/build/obj_$name/${f%.hip}.o
  (cd $csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -fno-fast-math -fno-slp-vectorize \
     -DSO3X_AB_BUILD $flags -c $f -o $o)
  objs="$objs $o"; skip="$skip ${f%.hip}.o"
done
for o in $csrc/so3x_*.o; do
  b=$(basename $o)
  case " $skip " in *" $b "*) ;; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/libso3x_$name.so $objs
echo built build/libso3x_$name.so
