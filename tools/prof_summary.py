#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite .db) per kernel: calls, total/avg/min/max duration,
and PMC counter sums/averages when the run collected counters.  Writes markdown to stdout.

  python tools/prof_summary.py gpurun_out/prof22/r1a_results.db > profiles/r01_scale22_kernel_stats.md
"""
import re
import sqlite3
import subprocess
import sys


def table(db, prefix):
    for (n,) in db.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix):
            return n
    return None


def short(name):
    if name.startswith("_Z"):
        try:
            name = subprocess.check_output(["c++filt", name.replace(".kd", "")]).decode().strip()
        except Exception:
            pass
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("GraphMat::dev::", "").replace("void ", "")
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    kd = table(db, "rocpd_kernel_dispatch")
    ks = table(db, "rocpd_info_kernel_symbol")
    rows = db.execute("select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                      "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (kd, ks)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, t, mn, mx in rows[:int(__import__("os").environ.get("PROF_ROWS", "25"))]:
        print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (short(name), n, t / 1e6, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    pe = table(db, "rocpd_pmc_event")
    pi = table(db, "rocpd_info_pmc")
    if pe and db.execute("select count(*) from %s" % pe).fetchone()[0]:
        print("\nPMC counters (sum over dispatches / per dispatch):\n")
        print("| kernel | counter | dispatches | sum | per dispatch |")
        print("|---|---|---:|---:|---:|")
        q = ("select s.kernel_name, p.name, count(*), sum(e.value) from %s e join %s p on e.pmc_id = p.id "
             "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by 1,2 order by 1,2" % (pe, pi, kd, ks))
        for name, c, n, v in db.execute(q):
            print("| `%s` | %s | %d | %.4g | %.4g |" % (short(name), c, n, v, v / n))


if __name__ == "__main__":
    main(sys.argv[1])
