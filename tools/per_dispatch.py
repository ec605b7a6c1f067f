#!/usr/bin/env python3
"""List kernel dispatches of a rocprofv3 rocpd database in time order (name, duration us)."""
import re, sqlite3, subprocess, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
def table(p):
    return [n for (n,) in db.execute("select name from sqlite_master where type='table'") if n.startswith(p)][0]
kd, ks = table("rocpd_kernel_dispatch"), table("rocpd_info_kernel_symbol")
for name, st, en in db.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)):
    if pat in name:
        short = subprocess.check_output(["c++filt", name.replace(".kd", "")]).decode().strip() if name.startswith("_Z") else name
        short = re.sub(r"\(.*$", "", short).replace("GraphMat::dev::", "").replace("void ", "")[:60]
        print("%-62s %10.1f us" % (short, (en - st) / 1e3))
