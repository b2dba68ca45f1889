"""Summarise rocprofv3 --kernel-trace --stats CSVs with short kernel names.
Usage: python tools/kstats.py <dir-or-csv> [top_n]"""
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*?>)?)", name)
    s = m.group(1) if m else name
    return s[:90]


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
    for f in files:
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        print(f"# {f}  total {tot/1e6:.3f} ms")
        for r in rows[:top]:
            print(f"{short(r['Name']):90s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}  "
                  f"{100*float(r['TotalDurationNs'])/tot:5.1f}%")


if __name__ == "__main__":
    main()
