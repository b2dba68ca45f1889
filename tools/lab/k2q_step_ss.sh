#!/bin/bash
# Lab: K2q's durations inside the single-stream benchmark step (kernel trace).  Usage (GPU box): tools/lab/k2q_step_ss.sh <outdir>
out=${1:-gpurun_out/k2q_ss}; root=$(pwd); mkdir -p $out
cd /tmp && export TMPDIR=/tmp
JSPSR_BRANCH_STREAMS=0 JSPSR_WGRAD_ASYNC=0 rocprofv3 --kernel-trace --stats -f csv -d $root/$out/ss -o t -- python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-roofline --no-inference --no-graph > $root/$out/ss.log 2>&1
cd $root
python3 tools/kstats.py $out/ss 6 > $out/ss.txt
f=$(find $out/ss -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/ktrace.py $f 40 > $out/ss_grid.txt
head -1 $out/ss.txt; grep -i "conv128" $out/ss_grid.txt | cut -c1-200
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
prev = None
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    n = r["Kernel_Name"]
    if "conv128" in n:
        d[n[:40]].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, prev))
    prev = n[:50]
for k, v in d.items():
    print(k, " ".join(f"{x[0]:.0f}" for x in v))
    big = [x for x in v if x[0] > 400]
    print("   before the long ones:", collections.Counter(x[1] for x in big).most_common(4))
PY
find $out \( -name "*.db" -o -name "*.csv" \) -delete
