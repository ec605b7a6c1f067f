#!/usr/bin/env python3
"""Per-wave time stamps of the persistent multiply kernels' last launch (ablation build only).

    GRAPHMAT_HIP_LIBRARY=build/ablation/libgraphmat_hip.so python tools/wave_times_probe.py --scale 26

Runs a few PageRank iterations, then reads g_abl_wave_times: for k_spmv_wave16p and k_spmv_rowwave, per wave of the grid,
the 100 MHz time stamps at kernel entry, after the hot set is in LDS, and at the end."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphmat_amd import _lib, api  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--lib-option", action="append", default=[])
    args = ap.parse_args()
    L = _lib.lib()
    for kv in args.lib_option:
        k, v = kv.split("=")
        L.gm_set_option(k.encode(), int(v))
    nv, s, d, _ = api.rmat_on_device(args.scale, 16, 1)
    g = api.Graph(nv, s, d, None, keep_values=False)
    del s, d
    st = g.new_pr_state()
    g.run_degree(st)
    g.run_pagerank(st, args.iters)
    raw = C.CDLL(os.environ["GRAPHMAT_HIP_LIBRARY"])
    buf = np.zeros((2, 8192, 4), np.uint64)
    raw.gm_abl_wave_times.argtypes = [C.c_void_p, C.c_size_t]
    assert raw.gm_abl_wave_times(buf.ctypes.data, buf.nbytes) == 0
    for ki, name in enumerate(("k_spmv_wave16p", "k_spmv_rowwave")):
        t = buf[ki].astype(np.int64)
        used = t[:, 2] > 0
        t = t[used]
        if len(t) == 0:
            print(name, "no stamps")
            continue
        base = t[:, 0].min()
        start = (t[:, 0] - base) / 100.0
        hot = (t[:, 1] - t[:, 0]) / 100.0
        work = (t[:, 2] - t[:, 1]) / 100.0
        end = (t[:, 2] - base) / 100.0
        q = lambda a: "min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
        print("%s: %d waves, last launch (us): kernel span %.1f" % (name, len(t), end.max()))
        print("   wave start after the first wave's: " + q(start))
        print("   hot set load + barrier:            " + q(hot))
        print("   work after the barrier:            " + q(work))
        print("   wave end:                          " + q(end))


if __name__ == "__main__":
    main()
