#!/bin/bash
# sweep of K1 variants (one process each: the shape is read once per process); the no-gather lab library is built by
# tools/lab/build_k1_nogather.sh
cd "$(dirname "$0")/.."
python tools/k1_lab.py
echo "--- no-gather lab build (streaming ceiling of the same load pattern; results are NOT the product kernel)"
JSPSR_LAB_LIB=jspsr_amd/lib_lab/libjspsr_nocompute.so python tools/k1_lab.py
JSPSR_LAB_LIB=jspsr_amd/lib_lab/libjspsr_nocompute.so JSPSR_LAB_LIB=jspsr_amd/lib_lab/libjspsr_nocompute.so 