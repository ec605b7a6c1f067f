"""Builds tests/support/libgm_shm_transport.so: the test suite's stand-in for librccl (host shared memory), loaded by
the product only when a test sets GRAPHMAT_RCCL_LIBRARY to it.  Test infrastructure -- see shm_transport.hip."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "shm_transport.hip")
SO = os.path.join(HERE, "libgm_shm_transport.so")


def build():
    if os.path.exists(SO) and os.path.getmtime(SO) >= os.path.getmtime(SRC):
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", SO, "-lrt"])
    return SO


if __name__ == "__main__":
    print(build())
