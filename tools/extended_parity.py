#!/usr/bin/env python3
"""Extended parity sweep against the oracle (longer than the regular -m gpu tests): several seeds,
reference thread counts (layout parameter), both device layouts, PageRank / BFS / SSSP / Degree.
  python tools/extended_parity.py > profiles/r01_extended_parity.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def main():
    from graphmat_amd import api, generators as gen
    from oracle import binding as ob
    ob.lib().gmo_set_num_threads(32)
    ok_all = True
    t00 = time.time()
    for scale in (17, 19, 21):
        for seed in (1, 2, 3):
            nv, s, d, v = gen.rmat_edges(scale, 16, seed=seed, weights="hash") if scale < 21 else (None,) * 4
            if nv is None:
                dnv, ds, dd, dv = api.rmat_on_device(scale, 16, seed, weights=True)
                nv, s, d, v = dnv, ds.cpu().numpy(), dd.cpu().numpy(), dv.cpu().numpy()
            for threads in ((1, 4) if scale < 21 else (2,)):
                og = ob.OracleGraph(nv, s, d, v, threads)
                odeg = og.degree()
                opr, oit, _ = og.pagerank(8, degree=odeg)
                for layout in (0, 1):
                    g = api.Graph(nv, s, d, v, ref_threads=threads, layout=layout)
                    pr, deg, it = g.pagerank(8)
                    ok = bool((deg == odeg).all()) and it == oit and bool((pr.view(np.uint32) == opr.view(np.uint32)).all())
                    srcs = (1, int(s[seed * 7]), int(d[seed * 13]))
                    for src in srcs:
                        depth, parent, itb = g.bfs(src)
                        od, op, oitb, _ = og.bfs(src)
                        ok &= itb == oitb and bool((depth == od).all()) and bool((parent == op).all())
                    dist, its = g.sssp(srcs[1])
                    odist, oits = og.sssp(srcs[1])
                    ok &= its == oits and bool((dist == odist).all())
                    ok_all &= ok
                    print("scale=%d seed=%d ref_threads=%d layout=%s: PageRank(8 it) bits, Degree, BFS x3 (depth+parent), SSSP (weighted): %s"
                          % (scale, seed, threads, "degree" if layout else "native", "PASS" if ok else "FAIL"), flush=True)
                    g.close()
                del og
    print("%s in %.0f s" % ("ALL PASS" if ok_all else "SOME FAILED", time.time() - t00))
    sys.exit(0 if ok_all else 1)

if __name__ == "__main__":
    main()
