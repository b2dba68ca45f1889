#!/bin/bash
# Lab builds (not product) of the persistent LDS-DMA propagation kernels (prop_dma.hip) -> jspsr_amd/lib_lab/:
#   libjspsr_k1d_stamps.so     -DK1D_STAMPS     per-phase cycle counters of the backward kernel (tools/lab/k1d_stamps.py)
#   libjspsr_k1d_nocompute.so  -DK1D_NOCOMPUTE  gather + arithmetic replaced by copies: the structure's streaming ceiling
# Loaded through JSPSR_LAB_LIB.  Results of the second are WRONG by construction.
set -e
cd "$(dirname "$0")/../../jspsr_amd/csrc"
make -s
mkdir -p ../lib_lab
for v in stamps nocompute; do
  D=$(echo $v | tr a-z A-Z)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DK1D_$D -c prop_dma.hip -o /tmp/k1d_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DK1D_$D -c prop.hip -o /tmp/k1d_prop_$v.o
  objs=$(ls _obj/*.o | grep -v "_obj/prop.o" | grep -v "_obj/prop_dma.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_lab/libjspsr_k1d_$v.so /tmp/k1d_$v.o /tmp/k1d_prop_$v.o $objs
done
ls -la ../lib_lab
