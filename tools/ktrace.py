"""Per (kernel, grid) totals from a rocprofv3 --kernel-trace CSV: which launches of a kernel are the slow ones.
Usage: python tools/ktrace.py <kernel_trace.csv> [top_n]"""
import collections
import csv
import re
import statistics as st
import sys


def short(n):
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"EEvPK.*$", "", n)
    return n[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    d = collections.defaultdict(list)
    for r in rows:
        wg = max(int(r["Workgroup_Size_X"]), 1)
        d[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // wg, wg)].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in d.values())
    print(f"# total {tot/1e3:.1f} ms, {len(rows)} launches")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print(f"{k[0]:60s} grid {k[1]:7d}x{k[2]:<4d} n {len(v):4d} med {st.median(v):8.1f} max {max(v):8.1f} us  total {sum(v)/1e3:7.2f} ms {100*sum(v)/tot:5.1f}%")


if __name__ == "__main__":
    main()
