#!/bin/bash
# sweep of K1 variants (one process each: the tile shape is read once per process); the no-gather lab library is built
# by tools/lab/build_k1_nogather.sh
cd "$(dirname "$0")/.."
python tools/k1_lab.py
for tw_th in "64 4" "64 16" "128 4" "128 8" "256 2" "256 4" "256 8"; do
  set -- $tw_th
  JSPSR_PROP_TW=$1 JSPSR_PROP_TH=$2 python tools/k1_lab.py
done
JSPSR_PROP_PX=2 python tools/k1_lab.py
JSPSR_PROP_PX=4 python tools/k1_lab.py
python tools/k1_lab.py 16 512 512
echo "--- no-gather lab build (streaming ceiling of the same load pattern; results are NOT the product kernel)"
JSPSR_LAB_LIB=jspsr_amd/lib_lab/libjspsr_nocompute.so python tools/k1_lab.py
JSPSR_LAB_LIB=jspsr_amd/lib_lab/libjspsr_nocompute.so JSPSR_PROP_TW=128 JSPSR_PROP_TH=8 python tools/k1_lab.py
