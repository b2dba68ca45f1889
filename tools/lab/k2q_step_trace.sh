#!/bin/bash
# Lab: the benchmark step with / without K2q (conv128.hip) under the kernel trace, multi-stream and single-stream:
# where the kernel's stand-alone gain goes inside the step.  Usage (on the GPU box): tools/lab/k2q_step_trace.sh <outdir>
out=${1:-gpurun_out/k2q_trace}; root=$(pwd); mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for arm in 1 0; do
  export JSPSR_CONV_RESIDENT128=$arm
  rocprofv3 --kernel-trace --stats -f csv -d $root/$out/ms$arm -o t -- python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-roofline --no-inference --no-graph > $root/$out/ms$arm.log 2>&1
  JSPSR_BRANCH_STREAMS=0 JSPSR_WGRAD_ASYNC=0 rocprofv3 --kernel-trace --stats -f csv -d $root/$out/ss$arm -o t -- python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-roofline --no-inference --no-graph > $root/$out/ss$arm.log 2>&1
done
cd $root
for d in ms1 ms0 ss1 ss0; do
  python3 tools/kstats.py $out/$d 14 > $out/$d.txt
  f=$(find $out/$d -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/ktrace.py $f 30 > $out/${d}_grid.txt
  head -1 $out/$d.txt; grep -h "ms_per_step" $out/$d.log | head -1 | cut -c1-120
done
find $out \( -name "*.db" -o -name "*.csv" \) -delete
