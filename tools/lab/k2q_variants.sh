#!/bin/bash
# Lab: stand-alone timings of K2q lab builds (tools/lab/build_k2q_variants.sh) on one box, two interleaved rounds.
# Usage (GPU box): tools/lab/k2q_variants.sh e12 e4 la6 ...      ("base" = the product library)
for round in 1 2; do
  for v in base "$@"; do
    lib=""; [ "$v" != base ] && lib=$(pwd)/jspsr_amd/lib_lab/libjspsr_k2q_$v.so
    echo "== $v (round $round)"
    JSPSR_LAB_LIB=$lib JSPSR_CONV_RESIDENT128_MIN=1 timeout -k 10 120 python tools/lab/k2q_check.py --child /tmp/k2q_v.pt time 2>&1 | grep "us " | grep -v "vs torch"
  done
done
