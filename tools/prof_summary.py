#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite output) as text for profiles/.
usage: python tools/prof_summary.py <results.db> [title]"""
import sqlite3
import sys


def main(db, title=""):
    con = sqlite3.connect(db)
    cur = con.cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary  {title}")
    print(f"# source: {db}")
    print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    rows = list(cur.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 from kernels "
        "group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1.0
    for n, c, t, a, mn, mx in rows:
        print(f"{n[:90]:90s} {c:6d} {t:12.1f} {a:10.2f} {mn:10.2f} {mx:10.2f} {100 * t / tot:6.2f}")
    print()
    print(f"{'kernel':90s} {'grid':>8s} {'wg':>5s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>8s}")
    for r in cur.execute("select name, max(grid_x), max(workgroup_x), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), "
                         "max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc"):
        print(f"{r[0][:90]:90s} {r[1]:8d} {r[2]:5d} {r[3]:5d} {r[4]:5d} {r[5]:5d} {r[6]:7d} {r[7]:8d}")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
