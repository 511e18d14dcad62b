#!/bin/bash
# A/B build of the library with extra -D flags on chosen translation units (development aid):
#   tools/probes/build_variant.sh <name> "<flags>" [unit ...]      (default unit: wmd_conv_wino32)
# -> tools/probes/_build/libwmd_<name>.so = the ordinary objects of wavelet_monodepth_amd/csrc with the named units recompiled.
# Load it through WMD_LIB_PATH.  The ordinary library must be up to date (python wavelet_monodepth_amd/build.py).
set -e
cd "$(dirname "$0")/../.."
NAME=$1; FLAGS=$2; shift 2
UNITS=${*:-wmd_conv_wino32}
mkdir -p tools/probes/_build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-bitwise-instead-of-logical -Wno-pass-failed -Iinclude $FLAGS"
EXCL=""
VOBJ=""
for u in $UNITS; do
  /opt/rocm/bin/hipcc $F -c wavelet_monodepth_amd/csrc/$u.hip -o tools/probes/_build/${u}_$NAME.o &
  EXCL="$EXCL -e $u.o"
  VOBJ="$VOBJ tools/probes/_build/${u}_$NAME.o"
done
wait
OBJS=$(ls wavelet_monodepth_amd/csrc/*.o | grep -v $EXCL)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/_build/libwmd_$NAME.so $OBJS $VOBJ -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
ls -la tools/probes/_build/libwmd_$NAME.so
