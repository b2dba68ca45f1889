"""profiles/k1_pmc.json from the two PMC passes of tools/profile_round.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on
tools/k1_lab.py): per-launch HBM bytes of the planar K1 kernels and of the head-fed K1h kernels.
FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte requests at 64 B on wide coalesced streams;
calibrated on tools/lab/stream_lab.hip: profiles/r01_stream_lab_pmc_calibration.txt); WRITE_SIZE is exact.
Usage: python tools/k1_pmc_json.py <fetch_dir> <write_dir> <tag> > profiles/k1_pmc.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def means(path, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def pick(d, *subs):
    for k, v in d.items():
        if all(s in k for s in subs):
            return v
    return None


def main():
    fd, wd, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    F, Wr = means(fd, "FETCH_SIZE"), means(wd, "WRITE_SIZE")
    px = 8 * 512 * 512
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, -f csv) on tools/k1_lab.py 8 512 512 1.5, {tag} "
                     "(tools/profile_round.sh -> tools/k1_pmc_json.py); FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md",
           "shape": "B=8 H=512 W=512, 16-channel offsets / 32-channel head"}
    import re
    # K1: the persistent LDS-DMA kernels, prop_dma_kernel<OC, NW, BWD, NTL, SPLIT, RP, SIG>: SIG = false at the public
    # boundary (keys fwd_* / bwd_*), SIG = true for the in-model logits entry (keys logits_fwd_* / logits_bwd_*)
    for prefix, sig in (("", "false"), ("logits_", "true")):
        for name, bwd in (("fwd", "false"), ("bwd", "true")):
            ks = [k for k in F if re.search(r"prop_dma_kernel<\d+, \d+, %s, \w+, \w+, \d+, %s>" % (bwd, sig), k)]
            if not ks or ks[0] not in Wr:
                continue
            k = ks[0]
            out[f"{prefix}{name}_kernel"] = re.search(r"prop_dma_kernel<[^>]*>", k).group(0)
            out[f"{prefix}{name}_fetch_kib_raw"] = round(F[k], 1)
            out[f"{prefix}{name}_write_kib"] = round(Wr[k], 1)
            out[f"{prefix}{name}_bytes_per_launch"] = int(round((2 * F[k] + Wr[k]) * 1024))
            out[f"{prefix}{name}_algorithmic_bytes"] = (108 if name == "fwd" else 208) * px
    # K1h: prop_head_dma_kernel<NW, BWD, SPLIT> (bf16 heads: what the models launch), prop_head_kernel<T, BWD> (fp32 heads)
    for dt, es, pat in (("bf16", 2, "__bf16"), ("f32", 4, "float")):
        for name, bwd in (("fwd", "false"), ("bwd", "true")):
            ks = [k for k in F if re.search(r"prop_head_kernel<%s, %s>" % (pat, bwd), k)]
            if dt == "bf16":
                ks = [k for k in F if re.search(r"prop_head_dma_kernel<\d+, %s, " % bwd, k)] or ks
                if ks:
                    out[f"head_bf16_{name}_kernel"] = re.search(r"prop_head(_dma)?_kernel<[^>]*>", ks[0]).group(0)
            if not ks or ks[0] not in Wr:
                continue
            k = ks[0]
            out[f"head_{dt}_{name}_bytes_per_launch"] = int(round((2 * F[k] + Wr[k]) * 1024))
            out[f"head_{dt}_{name}_moved_bytes_by_layout"] = ((32 if name == "fwd" else 64) * es + 8) * px
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
