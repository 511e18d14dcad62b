#!/bin/bash
# -DWMD_STAMPS build of the two translation units that carry conv_wino32_kernel's cycle stamps, linked with the ordinary
# objects of wavelet_monodepth_amd/csrc into tools/probes/_build/libwmd_hip_stamps.so (development aid, see stamps_probe.py)
set -e
cd "$(dirname "$0")/../.."
python wavelet_monodepth_amd/build.py
mkdir -p tools/probes/_build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-bitwise-instead-of-logical -Iinclude -DWMD_STAMPS"
for f in wmd_conv_wino32 wmd_conv_fwd; do
  /opt/rocm/bin/hipcc $F -c wavelet_monodepth_amd/csrc/$f.hip -o tools/probes/_build/${f}_st.o &
done
wait
OBJS=$(ls wavelet_monodepth_amd/csrc/*.o | grep -v -e wmd_conv_wino32.o -e wmd_conv_fwd.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/_build/libwmd_hip_stamps.so $OBJS tools/probes/_build/wmd_conv_wino32_st.o \
  tools/probes/_build/wmd_conv_fwd_st.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
ls -la tools/probes/_build/libwmd_hip_stamps.so
