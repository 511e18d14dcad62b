#!/bin/bash
# Build ablated variants of conv_wino32_kernel (W32_DBG bitmask) as separate libraries: gpurun_out/abl/libwmd_dbg<N>.so
# (development aid: tools/wino32_microbench.py under WMD_LIB_PATH times them; results of an ablated build are wrong by design)
set -e
cd "$(dirname "$0")/../wavelet_monodepth_amd"
mkdir -p ../build_abl
OBJS=$(ls csrc/*.o | grep -v wmd_conv_wino32.o)
for d in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -DW32_DBG=$d -c csrc/wmd_conv_wino32.hip -o ../build_abl/w32_dbg$d.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build_abl/libwmd_dbg$d.so $OBJS ../build_abl/w32_dbg$d.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib ) &
done
wait
ls -la ../build_abl/*.so
