#!/bin/bash
# Build libabcnet_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")/abc-net_amd/csrc"
OUT=../libabcnet_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result"
[ -n "$ABC_RESOURCE_USAGE" ] && FLAGS="$FLAGS -Rpass-analysis=kernel-resource-usage"
# debug build (ABC_KERNEL_DEBUG=1 ./build_hip.sh): phase-skipping ablations, in-kernel phase timestamps and the
# environment-driven experiment switches (abc_knob in common.hpp); the production library reads no environment variable
FLAVOUR=production
[ -n "$ABC_KERNEL_DEBUG" ] && FLAGS="$FLAGS -DABC_KERNEL_DEBUG=1" && FLAVOUR=debug
# (objects of the other flavour must not be linked)
[ "$(cat .build_flavour 2>/dev/null)" != "$FLAVOUR" ] && rm -f *.o *.d
echo $FLAVOUR > .build_flavour

# an object is stale when its source or ANY header it included last time (the -MD dependency file) is newer
stale() {
  local f=$1
  [ -f $f.o ] && [ -f $f.d ] || return 0
  local dep
  for dep in $(sed -e 's/^[^:]*://' -e 's/\\$//' $f.d); do
    [ -e "$dep" ] || return 0
    [ "$dep" -nt $f.o ] && return 0
  done
  return 1
}

OBJS=""
pids=""
for f in conv_igemm conv_fast conv_narrow stem heads heads_fused wgrad bn_act loss misc cbam metrics extract raster; do
  if stale $f; then
    hipcc $FLAGS -MD -MF $f.d -c $f.hip -o $f.o &
    pids="$pids $!"
  fi
  OBJS="$OBJS $f.o"
done
for p in $pids; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $(readlink -f $OUT)"
