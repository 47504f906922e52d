#!/usr/bin/env python3
"""Timeline of ONE converged-mode solve of the headline batch from a rocprofv3 `--kernel-trace` CSV: every kernel between two
consecutive first kernels of a solve (k_start<6>; k_ingest<6> on the unfused path) around a k_pipeline<6> + k_solve_wg<6> pair, with start
offset, duration and the gap in front of it.
usage: python tools/solve_timeline.py <kernel_trace.csv> [which solve, default: the 10th such]"""
import re
import csv, sys

def short(k):
    k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    return re.sub(r"<(\d)(, \w+)+>", lambda m: "<" + m.group(1) + ">", k)          # (template arguments that do not matter here: kernel variant, second chance)

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
first = "k_start<6>" if any(r[2] == "k_start<6>" for r in rows) else "k_ingest<6>"
starts = [i for i, r in enumerate(rows) if r[2] == first]
solves = []
for a, b in zip(starts, starts[1:]):
    names = [r[2] for r in rows[a:b]]
    if "k_pipeline<6>" in names and "k_solve_wg<6>" in names and names.count(first) == 1 and len(names) < 16:
        solves.append((a, b))
which = int(sys.argv[2]) if len(sys.argv) > 2 else min(10, len(solves) - 1)
a, b = solves[which]
t0 = rows[a][0]
prev_end = None
print(f"solve #{which} of {len(solves)}: period to the next {first} {(rows[b][0] - t0) / 1e3:.1f} us")
for s, e, n in rows[a:b]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"  {n:34s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:8.1f} us  gap before {gap:6.1f} us")
    prev_end = e
# averages over all such solves
import collections
dur = collections.defaultdict(list); gaps = []; period = []
for a, b in solves[2:]:
    pe = None
    for s, e, n in rows[a:b]:
        dur[n].append((e - s) / 1e3)
        if pe is not None: gaps.append(0); gaps[-1] = (s - pe) / 1e3
        pe = e
    period.append((rows[b][0] - rows[a][0]) / 1e3)
print("mean over %d solves: period %.1f us; kernels: %s; sum of gaps inside a solve %.1f us" % (len(period), sum(period) / max(1, len(period)),
      ", ".join(f"{n} {sum(v) / len(v):.1f}" for n, v in dur.items()), sum(gaps) / max(1, len(period))))
