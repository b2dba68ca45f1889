#!/bin/bash
# Lab build (not product): libjspsr_hip.so with the K1 gather replaced by a constant (-DJSPSR_LAB_NOCOMPUTE), to read the
# streaming ceiling of K1's load/store pattern (tools/k1_sweep.sh loads it through JSPSR_LAB_LIB).  Results of this
# library are WRONG by construction.
set -e
cd "$(dirname "$0")/../../jspsr_amd/csrc"
make -s
mkdir -p ../lib_lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DJSPSR_LAB_NOCOMPUTE -c prop.hip -o /tmp/prop_nogather.o
objs=$(ls _obj/*.o | grep -v "_obj/prop.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_lab/libjspsr_nocompute.so /tmp/prop_nogather.o $objs
echo "built jspsr_amd/lib_lab/libjspsr_nocompute.so"
