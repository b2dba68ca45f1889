#!/bin/bash
# Lab builds (not product) of the step kernels (prop_steps.hip) -> jspsr_amd/lib_lab/libjspsr_k1s_<name>.so, loaded through
# JSPSR_LAB_LIB.  Usage: tools/lab/build_k1s_variants.sh plain="-DK1S_LAB_PLAINSTORE" nofar="-DK1S_LAB_NOFAR" ...
set -e
cd "$(dirname "$0")/../../jspsr_amd/csrc"
make -s -j6
mkdir -p ../lib_lab
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c prop_steps.hip -o /tmp/k1s_$name.o
  objs=$(ls _obj/*.o | grep -v "_obj/prop_steps.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_lab/libjspsr_k1s_$name.so /tmp/k1s_$name.o $objs
done
ls -la ../lib_lab
