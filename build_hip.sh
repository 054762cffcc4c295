#!/bin/bash
# Build libabcnet_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")/abc-net_amd/csrc"
OUT=../libabcnet_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result"
[ -n "$ABC_RESOURCE_USAGE" ] && FLAGS="$FLAGS -Rpass-analysis=kernel-resource-usage"
# debug build: phase-skipping ablations (ABC_CONV_DBG / ABC_WGRAD_DBG) and in-kernel phase timestamps (scratch/prof_*.py)
FLAVOUR=production
[ -n "$ABC_KERNEL_DEBUG" ] && FLAGS="$FLAGS -DABC_KERNEL_DEBUG=1" && FLAVOUR=debug
# (objects of the other flavour must not be linked: rebuild the kernels that differ when the flavour changes)
[ "$(cat .build_flavour 2>/dev/null)" != "$FLAVOUR" ] && rm -f conv_fast.o conv_igemm.o wgrad.o heads_fused.o
echo $FLAVOUR > .build_flavour
OBJS=""
pids=""
for f in conv_igemm conv_fast conv_narrow stem heads heads_fused wgrad bn_act loss misc cbam metrics extract raster; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.hpp -nt $f.o ] || [ conv_fast.hpp -nt $f.o ] || [ loss_math.hpp -nt $f.o ] || [ capi_util.hpp -nt $f.o ] || [ ../../include/abcnet_hip.h -nt $f.o ]; then
    hipcc $FLAGS -c $f.hip -o $f.o &
    pids="$pids $!"
  fi
  OBJS="$OBJS $f.o"
done
for p in $pids; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $(readlink -f $OUT)"
