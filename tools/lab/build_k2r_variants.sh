#!/bin/bash
# Lab: libjspsr_hip.so variants of K2r (conv64.hip) with different fragment look-ahead depths -> jspsr_amd/lib/lab_k2r_la<N>.so
# (picked up through JSPSR_LAB_LIB by tools/bench_conv.py).  Usage: tools/lab/build_k2r_variants.sh 3 4 6 8 10 [stamps]
set -e
cd "$(dirname "$0")/../../jspsr_amd/csrc"
make -s -j6
for la in "$@"; do
  extra=""; name=""
  if [ "$la" = "stamps" ]; then extra="-DK2R_STAMPS"; la=4; name=lab_k2r_stamps; fi      # per-phase cycle counters printed by two waves
  case "$la" in e*) extra="-DK2R_EARLY=${la#e}"; name=lab_k2r_$la; la=4;; esac       # e<N>: pieces requested by each of waves 0-3
  case "$la" in o*) o=${la#o}; extra="-DK2R_ORDER=${o%%e*} -DK2R_EARLY=${o##*e}"; name=lab_k2r_$la; la=4;; esac    # o<order>e<N>
  M0_FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DK2R_LA=$la $extra" ./check_m0.sh conv64.hip      # same M0 guard as the product build
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DK2R_LA=$la $extra -c conv64.hip -o _obj/conv64_la$la.o
  objs=$(ls _obj/*.o | grep -v "conv64")
  [ -z "$name" ] && name=lab_k2r_la$la
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/$name.so $objs _obj/conv64_la$la.o
  rm _obj/conv64_la$la.o
done
ls -la ../lib
