#!/bin/bash
# K1 forms in steady state (300 warm-up launches into every timed block, tools/k1_lab.py), one box, two rounds interleaved.
for r in 1 2; do
  for v in "X=0" "JSPSR_PROP_SPLIT=1" "JSPSR_PROP_SPLIT=0" "JSPSR_PROP_RP=2" "JSPSR_PROP_RP=4" "JSPSR_PROP_NW=8" "JSPSR_PROP_WGS=1" "JSPSR_PROP_WGS=3" "JSPSR_PROP_NTL=0" "JSPSR_PROP_DMA=0"; do
    env $v timeout -k 10 120 python tools/k1_lab.py 2>/dev/null | grep "8x512x512" | sed "s/^\[[^]]*\]/[$v]/"
  done
done
