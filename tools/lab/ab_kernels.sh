#!/bin/bash
# Lab: single-stream kernel tables of the benchmark step under two environments -> gpurun_out/abk_<tag>.txt
# usage: tools/lab/ab_kernels.sh "VAR=a" "VAR=b"
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for arm in "$1" "$2"; do
  i=$((i+1))
  out=$root/gpurun_out/abk_$i
  rm -rf $out; mkdir -p $out
  env $arm JSPSR_BRANCH_STREAMS=0 JSPSR_WGRAD_ASYNC=0 rocprofv3 --kernel-trace --stats -f csv -d $out -o t -- python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-roofline --no-inference --no-graph > $out/log 2>&1
  (cd $root && echo "== $arm" && python3 tools/kstats.py $out 30) > $root/gpurun_out/abk_$i.txt
  find $out \( -name "*.db" -o -name "*.csv" \) -delete
done
