#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 `--kernel-trace` CSV (…_kernel_trace.csv), with the launches of the persistent pipeline
kernel of the headline configuration (`k_pipeline<6>`) split by what they ran: `python bench.py` times converged-mode solves
(mean 8 iterations per instance), then a profiled pass of the same, then fixed-20 solves -- one kernel name, two very different
launches.  The split is by duration (the two clusters do not overlap: converged ~1.1 ms, fixed-20 ~1.45 ms).
usage: python tools/trace_split.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def main(path):
    rows = defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in rows.values())
    print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        short = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        print(f"{short[:72]:72s} {len(v):6d} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {100 * sum(v) / tot:6.2f}")
        if "k_pipeline<6>" in k:
            cut = 0.5 * (min(v) + max(v))
            lo, hi = [x for x in v if x <= cut], [x for x in v if x > cut]
            if lo and hi and min(hi) > 1.1 * max(lo):
                print(f"{'    converged-mode launches (the bench line)':72s} {len(lo):6d} {sum(lo) / len(lo):10.1f} {min(lo):10.1f} {max(lo):10.1f}")
                print(f"{'    fixed-20 launches':72s} {len(hi):6d} {sum(hi) / len(hi):10.1f} {min(hi):10.1f} {max(hi):10.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
