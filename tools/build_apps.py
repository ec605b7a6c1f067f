#!/usr/bin/env python3
"""Builds application binaries against include/ with `hipcc --hipstdpar` (gfx950):

  * apps/*.cpp            -- this project's own example programs (always)
  * <reference>/src/{PageRank,BFS,SGD,SSSP,IncrementalPageRank,TopologicalSort,DeltaStepping}.cpp -- the reference's UNCHANGED application
    sources, compiled where they lie when the reference tree is present (build container
    only).  Outputs go to build/ref_apps/ (git-ignored; they travel to the GPU box like any
    other built artefact).  Nothing from the reference is copied into the repository.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GRAPHMAT_REFERENCE", "/root/reference")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "--hipstdpar", "-ffp-contract=off", "-w", "-DGRAPHMAT_NO_MPI",
         "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "graphmat_amd"), "-lgraphmat_hip"]


def _newest_header():
    newest = 0.0
    for d, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            newest = max(newest, os.path.getmtime(os.path.join(d, f)))
    return newest


def _compile(src, out, rpath, extra=()):
    if os.path.exists(out) and os.path.getmtime(out) >= max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "graphmat_amd", "libgraphmat_hip.so")), _newest_header()):
        return
    cmd = [HIPCC] + FLAGS + list(extra) + [src, "-o", out, "-Wl,-rpath," + rpath]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode())
        raise RuntimeError("hipcc failed on %s" % src)


def build_one(name):
    """Build a single program of apps/ (used by tests that need just one tool)."""
    outdir = os.path.join(ROOT, "build", "apps")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, name)
    _compile(os.path.join(ROOT, "apps", name + ".cpp"), out, "$ORIGIN/../../graphmat_amd")
    return out


def build(verbose=False, jobs=None):
    """Compile every application (a few at a time: each hipcc run is single-threaded)."""
    from concurrent.futures import ThreadPoolExecutor
    work = []
    appdir = os.path.join(ROOT, "apps")
    outdir = os.path.join(ROOT, "build", "apps")
    os.makedirs(outdir, exist_ok=True)
    for f in sorted(os.listdir(appdir)):
        if f.endswith(".cpp"):
            work.append((os.path.join(appdir, f), os.path.join(outdir, f[:-4])))
    if os.path.isdir(os.path.join(REF, "src")):
        outdir = os.path.join(ROOT, "build", "ref_apps")
        os.makedirs(outdir, exist_ok=True)
        for app in ("PageRank", "BFS", "SGD", "SSSP", "IncrementalPageRank", "TopologicalSort", "DeltaStepping"):
            work.append((os.path.join(REF, "src", app + ".cpp"), os.path.join(outdir, app)))
        # the reference's own tracing build (-D__TIMING, its Makefile's `timing` flavour): per-iteration lines
        work.append((os.path.join(REF, "src", "PageRank.cpp"), os.path.join(outdir, "PageRank__TIMING"), ("-D__TIMING",)))
    jobs = jobs or max(1, min(6, (os.cpu_count() or 2) - 1))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(lambda w: _compile(w[0], w[1], "$ORIGIN/../../graphmat_amd", w[2] if len(w) > 2 else ()), work))
    built = [w[1] for w in work]
    if verbose:
        print("\n".join(built))
    return built


if __name__ == "__main__":
    build(verbose=True)
