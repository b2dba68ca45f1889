#!/bin/bash
# Lab builds (not product) of the persistent LDS-DMA propagation kernels (prop_dma.hip) -> jspsr_amd/lib_lab/libjspsr_k1d_<name>.so
#   stamps     -DK1D_STAMPS     per-phase cycle counters of the backward kernel (tools/lab/k1d_stamps.py)
#   nocompute  -DK1D_NOCOMPUTE  gather + arithmetic replaced by copies: the structure's streaming ceiling (WRONG results)
#   any other name=flags pair, e.g.  tg9="-DK1D_TG=9"  plainst="-DK1D_NTS=0"  stfirst="-DK1D_STORE_FIRST=1"
# Loaded through JSPSR_LAB_LIB.  Usage: tools/lab/build_k1d_variants.sh stamps nocompute tg9="-DK1D_TG=9" ...
set -e
cd "$(dirname "$0")/../../jspsr_amd/csrc"
make -s
mkdir -p ../lib_lab
for v in "$@"; do
  name=${v%%=*}; flags=""
  case "$v" in *=*) flags=${v#*=};; stamps) flags="-DK1D_STAMPS";; nocompute) flags="-DK1D_NOCOMPUTE";; esac
  M0_FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags" ./check_m0.sh prop_dma.hip      # the lab -D switches change the code: same M0 guard as the product build
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c prop_dma.hip -o /tmp/k1d_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c prop.hip -o /tmp/k1d_prop_$name.o
  objs=$(ls _obj/*.o | grep -v "_obj/prop.o" | grep -v "_obj/prop_dma.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_lab/libjspsr_k1d_$name.so /tmp/k1d_$name.o /tmp/k1d_prop_$name.o $objs
done
ls -la ../lib_lab
