"""Average a rocprofv3 --pmc counter per kernel: python tools/pmc_summary.py <dir> [filter]"""
import csv, glob, os, re, sys
from collections import defaultdict
path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        m = re.match(r"([\w:]+(<[^(]*?>)?)", name)
        name = (m.group(1) if m else name)[:70]
        if flt in name:
            acc[(name, r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for (n, c, g), v in sorted(acc.items()):
        print(f"{n:70s} grid {g:>9s} {c:12s} n={len(v):3d} mean={sum(v)/len(v):14.1f} min={min(v):14.1f} max={max(v):14.1f}")
