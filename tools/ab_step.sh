#!/bin/bash
# A/B of the benchmark step on ONE box: tools/ab_step.sh "VAR=a" "VAR=b" [reps] [steps]  -> ms/step of each arm, interleaved
A="$1"; B="$2"; N=${3:-3}; S=${4:-20}
for i in $(seq $N); do
  for arm in "$A" "$B"; do
    ms=$(env $arm timeout -k 10 300 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-fp32 --no-roofline --no-inference --no-graph 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$arm  $ms ms/step"
  done
done
