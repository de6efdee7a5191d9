#!/usr/bin/env python3
"""Average a set of rocprofv3 PMC counters per kernel symbol (full-size launches only).
    python tools/pmc_counters.py <dir with the counter_collection csv files of one or more passes> [kernel substring ...]"""
import csv, glob, os, sys
from collections import defaultdict

def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    cut = name.find(">(")
    return name[:cut + 1].strip() if cut >= 0 else name.split("(")[0].strip()

def main():
    root, keys = sys.argv[1], sys.argv[2:] or ["fused_post_kernel", "fused_pre_kernel", "cgstep2_kernel", "dia_strip_kernel<double, float, double, 0, 5"]
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if any(s in k for s in keys):
                    acc[k][row["Counter_Name"]].append((int(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
    for k in sorted(acc):
        gmax = max(g for vals in acc[k].values() for g, _ in vals)
        print(k)
        for c in sorted(acc[k]):
            vals = [v for g, v in acc[k][c] if g == gmax]
            print(f"    {c:42s} {sum(vals) / len(vals):16.1f}   (n={len(vals)})")

if __name__ == "__main__":
    main()
