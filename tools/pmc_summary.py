#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch.
usage: python tools/pmc_summary.py <dir with p*/p*_counter_collection.csv> [--json profiles/pmc_traffic.json]

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (both in KiB): on gfx950 FETCH_SIZE counts 64 B per 128-B request
(MI355X_MICROARCH.md, HBM section); verified for this code's 8-byte-per-lane pattern with tools/ubench/ldpat.hip
(145.2 MB read -> FETCH_SIZE 71 553 KiB; 145.2 MB written -> WRITE_SIZE 143 884 KiB incl. the 1 MB result array)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, json_out=None):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "p*", "*_counter_collection.csv"))):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                short = name.replace("void (anonymous namespace)::", "").split("(")[0]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc, key=lambda k: -len(acc[k])):
        if not any(t in k for t in ("k_stage", "k_riccati", "k_pipeline", "k_solve_wg", "k_forces_qp", "k_forces_stage")):
            continue
        print(f"== {k}")
        for c, v in sorted(acc[k].items()):
            # early-exit launches (converged batch) are excluded by taking dispatches above 10% of the max
            mx = max(v)
            vv = [x for x in v if x > 0.1 * mx] or v
            print(f"   {c:28s} n={len(vv):4d} mean={sum(vv) / len(vv):16.1f} max={mx:16.1f}")


    if json_out:
        import json
        out = {}
        for k, cs in acc.items():
            if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
                continue
            name = ("k_stage" if "k_stage" in k and "true" not in k.split(",")[1] else
                    "k_riccati" if "k_riccati" in k else "k_pipeline" if "k_pipeline<6" in k else None)     # <6>: the metric configuration
            if name is None:
                continue

            def live(v):
                mx = max(v)
                return [x for x in v if x > 0.1 * mx] or v
            fe, wr = live(cs["FETCH_SIZE"]), live(cs["WRITE_SIZE"])
            if name == "k_pipeline":
                # one launch per solve: the bench command runs converged-mode solves (mean 8 iterations) and fixed-20
                # solves; `mean` is the converged-mode cluster (what the bench line's roofline object is quoted on)
                def low(v):
                    mid = 0.5 * (min(v) + max(v))
                    return [x for x in v if x <= mid] or v
                fe_lo, wr_lo = low(fe), low(wr)
                out[name] = dict(fetch_size_kib_mean=sum(fe_lo) / len(fe_lo), write_size_kib_mean=sum(wr_lo) / len(wr_lo),
                                 fetch_size_kib_max=max(fe), write_size_kib_max=max(wr),
                                 hbm_bytes_per_launch_mean=(2 * sum(fe_lo) / len(fe_lo) + sum(wr_lo) / len(wr_lo)) * 1024,
                                 hbm_bytes_per_launch_full=(2 * max(fe) + max(wr)) * 1024,
                                 note="2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes over `python bench.py --steps 2 --warmup 1`; one launch "
                                      "per solve: mean = converged-mode solves (mean 8 iterations per instance), full = 20 fixed iterations")
                continue
            out[name] = dict(fetch_size_kib_mean=sum(fe) / len(fe), write_size_kib_mean=sum(wr) / len(wr),
                             fetch_size_kib_max=max(fe), write_size_kib_max=max(wr),
                             hbm_bytes_per_launch_mean=(2 * sum(fe) / len(fe) + sum(wr) / len(wr)) * 1024,
                             hbm_bytes_per_launch_full=(2 * max(fe) + max(wr)) * 1024,
                             note="2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes over `python bench.py --steps 2 --warmup 1` "
                                  "(converged mode; launches below 10% of the maximum, i.e. early exits, excluded)")
        json.dump(out, open(json_out, "w"), indent=1)
        print("wrote", json_out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--json" else None)
