#!/bin/bash
# Round profile: bench line, rocprofv3 kernel stats (multi-stream and single-stream step), K1 micro-benchmark stats and
# the PMC passes behind profiles/k1_pmc.json.  Usage (on the GPU box): tools/profile_round.sh r02
set -e
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err
echo "[profile] bench line done"
rocprofv3 --kernel-trace --stats -f csv -d $out/step -o step -- python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-roofline --no-inference --no-graph > $out/step.log 2>&1
echo "[profile] multi-stream step traced"
JSPSR_BRANCH_STREAMS=0 JSPSR_WGRAD_ASYNC=0 rocprofv3 --kernel-trace --stats -f csv -d $out/step1s -o step1s -- python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-roofline --no-inference --no-graph > $out/step1s.log 2>&1
echo "[profile] single-stream step traced"
# the bench line's roofline leg under the profiler: the JSON line and the kernel averages come from ONE process
rocprofv3 --kernel-trace --stats -f csv -d $out/roof -o roof -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-inference --no-graph > $out/roof_line.json 2> $out/roof.err
echo "[profile] roofline leg traced"
rocprofv3 --kernel-trace --stats -f csv -d $out/k1 -o k1 -- python3 $root/tools/k1_lab.py > $out/k1.log 2>&1
echo "[profile] K1 micro-benchmark traced"
rocprofv3 --pmc FETCH_SIZE -f csv -d $out/k1_fetch -o k1f -- python3 $root/tools/k1_lab.py > $out/k1_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $out/k1_write -o k1w -- python3 $root/tools/k1_lab.py > $out/k1_write.log 2>&1
echo "[profile] PMC passes done"
cd $root
python3 tools/kstats.py $out/step 45 > $out/step_kernel_stats.txt
python3 tools/kstats.py $out/step1s 45 > $out/step1s_kernel_stats.txt
python3 tools/kstats.py $out/k1 12 > $out/k1_kernel_stats.txt
python3 tools/kstats.py $out/roof 400 | grep -i "prop_\|head_\|^#" > $out/roof_kernel_stats.txt
python3 - <<PY >> $out/roof_kernel_stats.txt
import json
l = json.loads(open("$out/roof_line.json").readline())["roofline"]
print("# the same process's JSON line: backward us_per_launch", l["us_per_launch"], l["us_per_launch_reps"], "frac", l["frac"],
      "| forward", l["forward"]["us_per_launch"], l["forward"]["us_per_launch_reps"], "frac", l["forward"]["frac"],
      "| public boundary bwd/fwd us", l["public_boundary"]["us_per_launch"], l["public_boundary"]["forward"]["us_per_launch"],
      "| K1c heads fwd/bwd us", l["head_conv"]["fwd_us"], l["head_conv"]["bwd_us"])
PY
python3 tools/pmc_summary.py $out/k1_fetch prop > $out/k1_pmc_counters.txt
python3 tools/pmc_summary.py $out/k1_write prop >> $out/k1_pmc_counters.txt
python3 tools/k1_pmc_json.py $out/k1_fetch $out/k1_write "round ${tag#r}" > $out/k1_pmc.json
f=$(find $out/step1s -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/ktrace.py $f 45 > $out/step1s_per_grid.txt
# keep the merged-back volume small: summaries only
find $out \( -name "*.db" -o -name "*.csv" -size +2M \) -delete
ls -la $out
