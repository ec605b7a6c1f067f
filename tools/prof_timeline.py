#!/usr/bin/env python3
"""Kernel dispatches of a rocprofv3 run (rocpd sqlite .db) in start order: start (us since the first listed one),
duration, gap to the previous end, name.  `--last N` keeps the last N dispatches, `--match RE` filters names.

  python tools/prof_timeline.py gpurun_out/x/bfs_results.db --last 120
"""
import argparse
import sqlite3
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import short, table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--last", type=int, default=0)
    ap.add_argument("--match", default="")
    args = ap.parse_args()
    import re
    db = sqlite3.connect(args.db)
    kd = table(db, "rocpd_kernel_dispatch")
    ks = table(db, "rocpd_info_kernel_symbol")
    rows = db.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
    if args.match:
        rows = [r for r in rows if re.search(args.match, r[2])]
    if args.last:
        rows = rows[-args.last:]
    if not rows:
        return
    t0 = rows[0][0]
    prev_end = t0
    names = {}
    print("| start us | dur us | gap us | kernel |")
    print("|---:|---:|---:|---|")
    for st, en, name in rows:
        if name not in names:
            names[name] = short(name)
        print("| %.1f | %.1f | %.1f | `%s` |" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, names[name][:90]))
        prev_end = max(prev_end, en)


if __name__ == "__main__":
    main()
