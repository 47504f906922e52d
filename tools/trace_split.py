#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 `--kernel-trace` CSV (…_kernel_trace.csv), with the launches of the two kernels of the
headline configuration's iteration loop split by what they ran -- one kernel name, very different launches:
  k_pipeline<6>   `python bench.py` times converged-mode solves (mean 8 iterations per instance; with the hybrid solve the pipeline
                  hands its tiles over after ~8 rounds), then a profiled pass of the same, then fixed-20 solves (no hand-over);
                  split by duration (the clusters do not overlap)
  k_solve_wg<6>   behind a converged-mode k_pipeline<6> launch it finishes the stragglers of the headline batch (the bench line);
                  on its own it solves a small batch alone (configuration 2, the sub-batches of a second chance); split by the
                  kernel that ran just before it
usage: python tools/trace_split.py <kernel_trace.csv>"""
import re
import csv
import sys
from collections import defaultdict


def short(k):
    k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    return re.sub(r"<(\d)(, \w+)+>", lambda m: "<" + m.group(1) + ">", k)          # (template arguments that do not matter here: kernel variant, second chance)


def main(path):
    rows = defaultdict(list)
    seq = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            rows[r["Kernel_Name"]].append((e - s) / 1e3)
            seq.append((s, short(r["Kernel_Name"]), (e - s) / 1e3))
    seq.sort()
    tot = sum(sum(v) for v in rows.values())
    pipe = [d for _, n, d in seq if n == "k_pipeline<6>"]
    cut = 0.5 * (min(pipe) + max(pipe)) if pipe else 0.0
    split_pipe = bool(pipe) and min([x for x in pipe if x > cut] or [0]) > 1.1 * max([x for x in pipe if x <= cut] or [1e30])
    behind, alone = [], []
    for i, (_, n, d) in enumerate(seq):
        if n == "k_solve_wg<6>":
            j = i - 1
            while j >= 0 and seq[j][1].startswith("__amd_rocclr_"):        # (runtime fills / copies between the two, if any, do not split a solve)
                j -= 1
            prev = seq[j] if j >= 0 else None
            (behind if prev and prev[1] == "k_pipeline<6>" and (not split_pipe or prev[2] <= cut) else alone).append(d)
    print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        print(f"{short(k)[:72]:72s} {len(v):6d} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {100 * sum(v) / tot:6.2f}")
        if short(k) == "k_pipeline<6>" and split_pipe:
            lo, hi = [x for x in v if x <= cut], [x for x in v if x > cut]
            print(f"{'    converged-mode launches (the bench line)':72s} {len(lo):6d} {sum(lo) / len(lo):10.1f} {min(lo):10.1f} {max(lo):10.1f}")
            print(f"{'    fixed-20 launches':72s} {len(hi):6d} {sum(hi) / len(hi):10.1f} {min(hi):10.1f} {max(hi):10.1f}")
        if short(k) == "k_solve_wg<6>":
            for label, w in (("    behind a converged-mode k_pipeline<6> launch (the bench line)", behind), ("    alone (small batches)", alone)):
                if w:
                    print(f"{label:72s} {len(w):6d} {sum(w) / len(w):10.1f} {min(w):10.1f} {max(w):10.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
