#!/usr/bin/env python3
"""Run the reference's UNCHANGED application binaries (build/ref_apps) on a synthetic RMAT file.
Usage: python tools/app_at_scale.py <scale> [uniform]   (uniform: 2^scale vertices x 16 uniformly drawn out-edges, values 1: a graph without skew; PageRank only)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphmat_amd import api
from graphmat_amd.mtx import write_mtx_bin
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
uniform = len(sys.argv) > 2 and sys.argv[2] == "uniform"
if uniform:
    import torch
    nv, s, d, _ = api.uniform_on_device(scale, 16, 1)
    v = torch.ones_like(s)
else:
    nv, s, d, v = api.rmat_on_device(scale, 16, 1, weights=True)
path = "/tmp/%s%d.bin.mtx" % ("uniform" if uniform else "rmat", scale)
write_mtx_bin(path, nv, s.cpu().numpy(), d.cpu().numpy(), v.cpu().numpy())
for app, args in ((("PageRank", []),) if uniform else (("PageRank", []), ("BFS", ["1"]), ("SSSP", ["1"]))):
    exe = os.path.join(ROOT, "build", "ref_apps", app)
    if not os.path.exists(exe):
        print(app, "binary not prebuilt"); continue
    t0 = time.time()
    out = subprocess.run([exe, path] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    keep = [l for l in out.splitlines() if any(k in l for k in ("Completed", "Time", "Reachable", "construction"))]
    print("== unchanged %s.cpp on %s-%d (wall %.1fs incl. host loader): %s" % (app, "uniform" if uniform else "RMAT", scale, time.time() - t0, " | ".join(keep)))
