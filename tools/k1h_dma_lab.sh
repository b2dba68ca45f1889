#!/bin/bash
# K1h lab: the head-fed propagation step (what the models launch) -- the 4-lanes-per-pixel kernels (prop_head.hip) against
# the persistent LDS-DMA kernels (prop_head_dma.hip), same tool (tools/k1_lab.py), one box.
run() { echo "--- $*"; env "$@" timeout -k 10 120 python tools/k1_lab.py 2>&1 | grep "head-fed bfloat16"; }
run JSPSR_PROP_HEAD_DMA=0
for split in 0 1; do for wgs in 3 2 1; do
  run JSPSR_PROP_HEAD_DMA=1 JSPSR_PROP_HEAD_SPLIT=$split JSPSR_PROP_HEAD_WGS=$wgs
done; done
run JSPSR_PROP_HEAD_DMA=0
