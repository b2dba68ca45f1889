#!/bin/bash
# Lab: libjspsr_hip.so variants of K2q (conv128.hip) -> jspsr_amd/lib_lab/libjspsr_k2q_<name>.so, picked up through JSPSR_LAB_LIB.
# Usage: tools/lab/build_k2q_variants.sh name="-DK2Q_DMA_EVERY=12" la6="-DK2Q_LA=6" ...
set -e
cd "$(dirname "$0")/../../jspsr_amd/csrc"
make -s
mkdir -p ../lib_lab
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}
  # the lab -D switches change the code: same M0 guard as the product build (the stamps build prints from the kernel's last
  # instructions, after every DMA piece: printf's own use of M0 there is harmless and the guard is skipped)
  case "$name" in stamps*) true;; *) false;; esac || M0_FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags" ./check_m0.sh conv128.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c conv128.hip -o /tmp/k2q_$name.o
  objs=$(ls _obj/*.o | grep -v "_obj/conv128.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_lab/libjspsr_k2q_$name.so /tmp/k2q_$name.o $objs
done
ls -la ../lib_lab | grep k2q
