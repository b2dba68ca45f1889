#!/bin/bash
# K1d lab: symmetric / split forms, NW, load policy, against the one-pixel-per-lane kernels and bare plane streams ON ONE BOX
run() { echo "--- $*"; env "$@" timeout -k 10 120 python tools/k1_lab.py 2>&1 | grep -v amdgpu.ids | grep -v "head-fed"; }
L=jspsr_amd/lib_lab
run JSPSR_PROP_DMA=0
for split in 0 1; do for nw in 4 8; do for ntl in 1 0; do
  run JSPSR_PROP_SPLIT=$split JSPSR_PROP_NW=$nw JSPSR_PROP_NTL=$ntl
done; done; done
for split in 0 1; do run JSPSR_PROP_SPLIT=$split JSPSR_PROP_NW=4 JSPSR_PROP_NTL=1 JSPSR_LAB_LIB=$L/libjspsr_k1d_nocompute.so; done
echo "== stream_lab (bare plane streams, same box)"; timeout -k 10 100 ./tools/lab/stream_lab | grep "pad=0\|pad=64 \|copy\|bchw"
run JSPSR_PROP_DMA=0
