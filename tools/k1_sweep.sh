#!/bin/bash
# sweep of K1 variants (one process each: the shape is read once per process)
cd "$(dirname "$0")/.."
python tools/k1_lab.py
JSPSR_PROP_PF=1 python tools/k1_lab.py
JSPSR_PROP_PF=1 JSPSR_PROP_TH=4 python tools/k1_lab.py
JSPSR_PROP_PF=1 JSPSR_PROP_TH=16 python tools/k1_lab.py
JSPSR_PROP_PF=1 JSPSR_PROP_TW=128 JSPSR_PROP_TH=8 python tools/k1_lab.py
JSPSR_PROP_PF=1 JSPSR_PROP_TW=128 JSPSR_PROP_TH=4 python tools/k1_lab.py
JSPSR_PROP_PF=1 JSPSR_PROP_TW=256 JSPSR_PROP_TH=4 python tools/k1_lab.py
JSPSR_PROP_PF=1 JSPSR_PROP_TW=256 JSPSR_PROP_TH=2 python tools/k1_lab.py
echo "--- no-gather lab build (streaming ceiling of the same load pattern; results are NOT the product kernel)"
JSPSR_LAB_LIB=jspsr_amd/lib_lab/libjspsr_nocompute.so python tools/k1_lab.py
JSPSR_LAB_LIB=jspsr_amd/lib_lab/libjspsr_nocompute.so JSPSR_PROP_PF=1 python tools/k1_lab.py
JSPSR_LAB_LIB=jspsr_amd/lib_lab/libjspsr_nocompute.so JSPSR_PROP_PF=1 JSPSR_PROP_TW=128 JSPSR_PROP_TH=8 python tools/k1_lab.py
