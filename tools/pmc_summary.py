#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch.
usage: python tools/pmc_summary.py <dir with p*/p*_counter_collection.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "p*", "*_counter_collection.csv"))):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                short = name.replace("void (anonymous namespace)::", "").split("(")[0]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc, key=lambda k: -len(acc[k])):
        if not ("k_stage" in k or "k_riccati" in k):
            continue
        print(f"== {k}")
        for c, v in sorted(acc[k].items()):
            # early-exit launches (converged batch) are excluded by taking dispatches above 10% of the max
            mx = max(v)
            vv = [x for x in v if x > 0.1 * mx] or v
            print(f"   {c:28s} n={len(vv):4d} mean={sum(vv) / len(vv):16.1f} max={mx:16.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
