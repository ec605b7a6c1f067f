"""The sharded (one process per shard) path end to end on the GPU: 2 and 3 shards, bit-exact
PageRank / BFS / SSSP against the oracle.  On the 1-GPU test box the shards share cuda:0 and
the collectives go over gloo; the data path (sharded CSR, slice exchange, convergence
all-reduce) is the one the multi-GPU bench uses."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_runs_match_oracle(world):
    from graphmat_amd import build
    build.build()
    from oracle import binding
    binding.build()
    env = dict(os.environ, GM_BACKEND="gloo", GM_SCALE="13")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "multi_check.py")]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env, cwd=ROOT)
    text = out.stdout.decode()
    assert out.returncode == 0 and "MULTI_OK" in text, text[-3000:]


def test_native_rccl_exchange_single_rank():
    """The library's own RCCL exchange (gm_dist.hip: ncclAllGather / ncclAllReduce on HIP streams, overlapped
    parts on a side stream).  RCCL wants one rank per GPU, so on this 1-GPU box the world has one rank: the
    communicator, the staged live-prefix gather, the two-stage PART/WAIT schedule, the flag all-reduce and
    the 512-byte-row SGD exchange all run for real, and every result must equal the oracle's."""
    from graphmat_amd import build
    build.build()
    from oracle import binding
    binding.build()
    env = dict(os.environ, GM_BACKEND="gloo", GM_SCALE="13", GM_EXCHANGE="native")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "multi_check.py")]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env, cwd=ROOT)
    text = out.stdout.decode()
    assert out.returncode == 0 and "MULTI_OK" in text and "native-rccl" in text, text[-3000:]


@pytest.mark.parametrize("world", [2, 3])
def test_native_exchange_several_ranks_over_shared_memory(world):
    """The native exchange's rank arithmetic (slice offsets, staged live prefixes, part regions, sparse blocks) with
    more than one participant: RCCL cannot put two ranks on one GPU, so the library's shared-memory test transport
    (GRAPHMAT_DIST_TRANSPORT=shm: the same all-gather / all-reduce entry points over host memory) carries the bytes
    while gm_dist.hip's exchange code runs unchanged.  Every result must equal the oracle's."""
    from graphmat_amd import build
    build.build()
    from oracle import binding
    binding.build()
    env = dict(os.environ, GM_BACKEND="gloo", GM_SCALE="13", GM_EXCHANGE="native", GRAPHMAT_DIST_TRANSPORT="shm")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "multi_check.py")]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env, cwd=ROOT)
    text = out.stdout.decode()
    assert out.returncode == 0 and "MULTI_OK" in text and "native-rccl" in text, text[-3000:]
