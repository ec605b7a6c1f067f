#!/usr/bin/env python3
"""PMC counters per kernel DISPATCH of a rocprofv3 rocpd database (prof_summary.py sums over dispatches):
python tools/pmc_per_dispatch.py results.db <kernel name substring> [min duration us]
one line per dispatch in time order: duration and every counter collected in that run."""
import re, sqlite3, subprocess, sys


def table(db, p):
    t = [n for (n,) in db.execute("select name from sqlite_master where type='table'") if n.startswith(p)]
    return t[0] if t else None


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    kd, ks = table(db, "rocpd_kernel_dispatch"), table(db, "rocpd_info_kernel_symbol")
    pe, pi = table(db, "rocpd_pmc_event"), table(db, "rocpd_info_pmc")
    rows = db.execute("select d.event_id, s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
    for ev, name, st, en in rows:
        if pat not in name or (en - st) / 1e3 < min_us:
            continue
        short = subprocess.check_output(["c++filt", name.replace(".kd", "")]).decode().strip() if name.startswith("_Z") else name
        short = re.sub(r"<.*$", "", re.sub(r"\(.*$", "", short)).replace("GraphMat::dev::", "").replace("void ", "")
        vals = db.execute("select p.name, sum(e.value) from %s e join %s p on e.pmc_id = p.id where e.event_id = ? group by 1 order by 1" % (pe, pi), (ev,)).fetchall()
        print("%-24s %9.1f us  %s" % (short[:24], (en - st) / 1e3, "  ".join("%s=%.4g" % (n, v) for n, v in vals)))


if __name__ == "__main__":
    main()
