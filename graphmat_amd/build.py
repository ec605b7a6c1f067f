"""Builds libgraphmat_hip.so (gfx950) in-tree with hipcc.  No GPU needed to build."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libgraphmat_hip.so")
SOURCES = ["gm_core.hip", "gm_graph.hip", "gm_programs.hip", "gm_dist.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-DGRAPHMAT_NO_MPI"]


def _deps():
    out = []
    for d in (CSRC, os.path.join(ROOT, "include"), os.path.join(ROOT, "include", "graphmat")):
        for f in os.listdir(d):
            if f.endswith((".hip", ".hpp", ".h")):
                out.append(os.path.join(d, f))
    out.append(os.path.abspath(__file__))
    return out


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    ablation = os.environ.get("GRAPHMAT_ABLATION") == "1"
    # in-kernel ablation switches: a separate library under build/ablation/ (experiments load it with
    # GRAPHMAT_HIP_LIBRARY=...), the product library is never an ablation build
    flags = FLAGS + (["-DGRAPHMAT_ABLATION"] if ablation else [])
    objdir = os.path.join(ROOT, "build", "ablation") if ablation else CSRC
    so = os.path.join(objdir, "libgraphmat_hip.so") if ablation else SO
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    deps_mtime = max(os.path.getmtime(p) for p in _deps())
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if (not force) and os.path.exists(obj) and os.path.getmtime(obj) >= deps_mtime:
            continue
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % src)
        elif verbose and out:
            sys.stderr.write(out.decode())
    if procs or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + ["-ldl", "-lrt"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return so


def build_hooks(verbose=False):
    """The TEST-HOOKS variant of the library (build/hooks/libgraphmat_hip.so): the product objects with gm_dist.hip compiled
    -DGM_TEST_HOOKS, which adds the fault injection the negative control of the multi-rank tests needs (GRAPHMAT_DEBUG_DROP_WAIT).
    Test infrastructure: loaded only through GRAPHMAT_HIP_LIBRARY, never by the product."""
    build(verbose=verbose)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(ROOT, "build", "hooks")
    os.makedirs(objdir, exist_ok=True)
    so = os.path.join(objdir, "libgraphmat_hip.so")
    obj = os.path.join(objdir, "gm_dist.o")
    deps_mtime = max(os.path.getmtime(p) for p in _deps())
    if not os.path.exists(obj) or os.path.getmtime(obj) < deps_mtime:
        subprocess.check_call([hipcc] + FLAGS + ["-DGM_TEST_HOOKS", "-c", os.path.join(CSRC, "gm_dist.hip"), "-o", obj])
    objs = [os.path.join(CSRC, src.replace(".hip", ".o")) for src in SOURCES if src != "gm_dist.hip"] + [obj]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(o) for o in objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + ["-ldl", "-lrt"])
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
