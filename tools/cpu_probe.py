import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from graphmat_amd import generators
from oracle import binding as ob
nv, s, d, v = generators.rmat_edges(20, 16, seed=1)
print("cores", os.cpu_count())
for t in (8, 16, 32, 64, 128):
    ob.lib().gmo_set_num_threads(t)
    t0 = time.time(); og = ob.OracleGraph(nv, s, d, None, ref_threads=t); deg = og.degree(); b = time.time() - t0
    og.pagerank(1, degree=deg)
    t0 = time.time(); og.pagerank(10, degree=deg); dt = time.time() - t0
    print("threads=%d build %.1fs 10 iters %.2fs -> %.3f GTEPS" % (t, b, dt, len(s) * 10 / dt / 1e9), flush=True)
    del og
