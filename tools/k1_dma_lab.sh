#!/bin/bash
# K1 lab: the persistent LDS-DMA kernels (prop_dma.hip) against the one-pixel-per-lane kernels (prop.hip), same tool
# (tools/k1_lab.py: back-to-back C-ABI launches between events).  Usage: bash tools/k1_dma_lab.sh > gpurun_out/k1_dma_lab.txt
set -o pipefail
run() { echo "--- $*"; env "$@" timeout -k 10 120 python tools/k1_lab.py 2>&1 | grep -v amdgpu.ids | grep -v "head-fed"; }
run JSPSR_PROP_DMA=0
run JSPSR_PROP_DMA=1 JSPSR_PROP_NW=8 JSPSR_PROP_NTL=1
run JSPSR_PROP_DMA=1 JSPSR_PROP_NW=8 JSPSR_PROP_NTL=0
run JSPSR_PROP_DMA=1 JSPSR_PROP_NW=4 JSPSR_PROP_NTL=1
run JSPSR_PROP_DMA=1 JSPSR_PROP_NW=4 JSPSR_PROP_NTL=0
run JSPSR_PROP_DMA=1 JSPSR_PROP_NW=4 JSPSR_PROP_NTL=1 JSPSR_PROP_WGS=1
