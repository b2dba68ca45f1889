#!/bin/bash
# K1d lab: rows per wave per tile (RP) in the symmetric form, both directions, one box; general kernels and streams beside
run() { echo "--- $*"; env "$@" timeout -k 10 120 python tools/k1_lab.py 2>&1 | grep -v amdgpu.ids | grep -v "head-fed"; }
run JSPSR_PROP_DMA=0
for rp in 1 2 4; do run JSPSR_PROP_SPLIT=0 JSPSR_PROP_RP=$rp; done
run JSPSR_PROP_SPLIT=1
for rp in 2 4; do run JSPSR_PROP_SPLIT=0 JSPSR_PROP_RP=$rp JSPSR_PROP_NTL=0; done
echo "== stream_lab (bare plane streams, same box)"; timeout -k 10 100 ./tools/lab/stream_lab | grep "pad=0\|copy"
for rp in 1 2 4; do run JSPSR_PROP_SPLIT=0 JSPSR_PROP_RP=$rp; done
