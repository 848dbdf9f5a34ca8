#!/bin/bash
# Same-box A/B builds of the library: gpurun_variants/lib_<name>.so = the library with ONE translation unit rebuilt with extra
# -D flags (the other objects are compiled once).   usage: tools/build_variants.sh <file.hip> name1:"-DA=1 -DB" name2:"..." ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/wheeledlab_amd/csrc
OBJ=/tmp/wl_variant_obj
mkdir -p $OBJ $R/gpurun_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=14"
UNIT=$1; shift
for f in wl_drift wl_elev wl_visual wl_depth wl_policy wl_ppo wl_ppo_wide wl_actor wl_startup; do
  if [ "$f.hip" != "$UNIT" ]; then
    if [ ! -f $OBJ/$f.o ] || [ -n "$(find $SRC $R/include -newer $OBJ/$f.o \( -name '*.h' -o -name "$f.hip" \) | head -1)" ]; then
      /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OBJ/$f.o &
    fi
  fi
done
wait
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $SRC/$UNIT -o $OBJ/var_$name.o
    objs=""
    for f in wl_drift wl_elev wl_visual wl_depth wl_policy wl_ppo wl_ppo_wide wl_actor wl_startup; do
      if [ "$f.hip" != "$UNIT" ]; then objs="$objs $OBJ/$f.o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $OBJ/var_$name.o -o $R/gpurun_variants/lib_$name.so
    echo "built lib_$name.so ($defs)" ) &
done
wait
