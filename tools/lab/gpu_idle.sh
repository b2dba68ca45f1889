#!/bin/bash
# Lab: how much of the multi-stream benchmark step the GPU is idle (no kernel running on any stream): the union of the kernel
# intervals of the last traced steps against their wall span.  Usage (GPU box): tools/lab/gpu_idle.sh
root=$(pwd); out=$root/gpurun_out/gpu_idle; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d $out/t -o t -- python3 $root/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-fp32 --no-roofline --no-inference --no-graph > $out/bench.log 2>&1
cd $root
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# steps end with the AdamW kernel
ends = [e for s, e, n in rows if "adamw" in n]
print("steps traced:", len(ends))
for a, b in zip(ends[-5:-1], ends[-4:]):
    iv = [(s, e) for s, e, n in rows if s >= a and e <= b]
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps = sorted(((s2 - e1) for (s1, e1), (s2, e2) in zip(iv, iv[1:]) if s2 > e1), reverse=True)
    print(f"step {(b - a) / 1e6:.2f} ms: some kernel running {busy / 1e6:.2f} ms, idle {(b - a - busy) / 1e6:.2f} ms; kernels {len(iv)}; sum of durations {sum(e - s for s, e in iv) / 1e6:.2f} ms")
PY
rm -rf $out/t
