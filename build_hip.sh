#!/bin/bash
# Build libabcnet_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")/abc-net_amd/csrc"
OUT=../libabcnet_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $ABC_EXTRA_FLAGS"   # (ABC_EXTRA_FLAGS: -D switches of measured A/B builds, e.g. -DABC_DEEP_TM=4)
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
for f in conv_igemm conv_fast conv_fast8 conv_fast_lp convt_fused conv_narrow stem heads heads_fused wgrad wgrad_narrow bn_act loss misc cbam metrics extract raster; do
  if stale $f; then
    hipcc $FLAGS -MD -MF $f.d -c $f.hip -o $f.o &
    pids="$pids $!"
  fi
  OBJS="$OBJS $f.o"
done
for p in $pids; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
# The gfx950 store-data hazard (a >8-byte buffer store WITH a scalar offset register, then a VALU write to its
# data registers within two instructions -- outside LLVM's hazard recogniser; the kernels keep the data registers live past such
# stores by hand): disassemble every code object of the library just built and fail the build on a hit.
if [ -z "$ABC_SKIP_HAZARD_SCAN" ]; then
  # llvm-objdump of the toolchain that built the library: beside hipcc's clang (hipconfig --rocmpath), else /opt/rocm
  ROCM=$(hipconfig --rocmpath 2>/dev/null || true)
  OBJDUMP=""
  for c in "$ROCM/lib/llvm/bin/llvm-objdump" "$(dirname "$(readlink -f "$(command -v hipcc)")")/../lib/llvm/bin/llvm-objdump" /opt/rocm/lib/llvm/bin/llvm-objdump; do
    [ -x "$c" ] && OBJDUMP="$c" && break
  done
  SCANNER="$(cd ../.. && pwd)/profiles/tools/store_hazard_scan.py"
  if [ -z "$OBJDUMP" ] || [ ! -f "$SCANNER" ]; then
    echo "WARNING: store-hazard scan skipped (llvm-objdump or profiles/tools/store_hazard_scan.py not found); the library is built"
  else
    SCAN=$(mktemp -d)
    cp $OUT $SCAN/lib.so
    "$OBJDUMP" --offloading $SCAN/lib.so > /dev/null
    n=0
    for co in $SCAN/lib.so.*gfx950*; do
      [ -f "$co" ] || continue
      "$OBJDUMP" -d "$co" > "$co.s"; n=$((n + 1))
    done
    if [ $n -eq 0 ]; then echo "store-hazard scan: no gfx950 code object found in the library"; rm -rf $SCAN; exit 1; fi
    python3 "$SCANNER" --uncovered $SCAN/*.s || { echo "store-data hazard in the built library (see above)"; rm -rf $SCAN; rm -f $OUT; exit 1; }
    rm -rf $SCAN
  fi
fi
echo "built $(readlink -f $OUT)"
