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


def _shm_lib():
    """tests/support/libgm_shm_transport.so (built on demand): the test suite's stand-in for librccl"""
    sys.path.insert(0, ROOT)
    from tests.support import build as shm_build
    return shm_build.build()


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


def test_sharded_runs_with_the_persistent_kernels_match_oracle():
    """The persistent multiply kernels (k_spmv_rowwave / k_spmv_wave16p) keep, on a sharded graph, the first entries
    of EVERY shard's slice of x in LDS (the degree ranking is dealt over the slices).  The library picks them for
    large graphs only (what bench.py --gpus N runs at RMAT-26), so here they are forced on a small one: 3 shards,
    every program of tools/multi_check.py against the oracle."""
    from graphmat_amd import build
    build.build()
    from oracle import binding
    binding.build()
    env = dict(os.environ, GM_BACKEND="gloo", GM_SCALE="14", GM_FORCE_FORMS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
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
    more than one participant: RCCL cannot put two ranks on one GPU, so the test suite's shared-memory stand-in for librccl
    (GRAPHMAT_RCCL_LIBRARY=tests/support/libgm_shm_transport.so: the same entry points over host memory) carries the bytes
    while gm_dist.hip's exchange code runs unchanged.  Every result must equal the oracle's."""
    from graphmat_amd import build
    build.build()
    from oracle import binding
    binding.build()
    env = dict(os.environ, GM_BACKEND="gloo", GM_SCALE="13", GM_EXCHANGE="native", GRAPHMAT_RCCL_LIBRARY=_shm_lib())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "multi_check.py")]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env, cwd=ROOT)
    text = out.stdout.decode()
    assert out.returncode == 0 and "MULTI_OK" in text and "native-rccl" in text, text[-3000:]


@pytest.mark.parametrize("world,exchange,scale", [(2, "callback", 15), (3, "callback", 15), (2, "native", 16), (3, "native", 16)])
def test_sharded_sweep_matches_oracle(world, exchange, scale):
    """The row-stationary sweep on a SHARD's rows (graphmat_hip.h: gm_sweep_t.nsub; kernels.hpp: k_spmv_sell_sharded; round 6): every
    owner's range of the device order is [slice][degree rank], slices are ascending native ranges, the structure is built from the
    shard's own rows (also by the distributed build), and PageRank through it -- with and without edge values, several launches,
    long rows staged in one or several rounds, either giant-row form -- has the oracle's bits with 2 and 3 ranks, over the
    torch.distributed callback and over the library's native exchange on the shared-memory stand-in for librccl
    (tools/multi_sweep_check.py)."""
    from graphmat_amd import build
    build.build()
    from oracle import binding
    binding.build()
    env = dict(os.environ, GM_BACKEND="gloo", GM_SCALE=str(scale), GM_EXCHANGE=exchange)
    if exchange == "native":
        env["GRAPHMAT_RCCL_LIBRARY"] = _shm_lib()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "multi_sweep_check.py")]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env, cwd=ROOT)
    text = out.stdout.decode()
    assert out.returncode == 0 and "SWEEP_MULTI_OK" in text, text[-3000:]


def test_missing_stream_wait_is_caught():
    """Negative control for the multi-rank tests.  The shared-memory stand-in for librccl is stream-ordered (its collectives only
    enqueue copies, reductions and a spinning rendezvous kernel on the stream they are handed), so the ordering between
    gm_dist.hip's side stream and the run stream is really exercised: with ONE hipStreamWaitEvent of the two-stage
    schedule left out (GRAPHMAT_DEBUG_DROP_WAIT=1: the run stream no longer waits for the side stream's all-gathers
    before it copies the parts into x) a 3-rank PageRank must differ from the oracle.  A blocking transport would let
    this bug pass."""
    from graphmat_amd import build
    hooks_so = build.build_hooks()
    from oracle import binding
    binding.build()
    # (the fault injection exists in the test-hooks build of the library only: build/hooks/libgraphmat_hip.so)
    env = dict(os.environ, GM_BACKEND="gloo", GM_SCALE="16", GM_EXCHANGE="native", GRAPHMAT_RCCL_LIBRARY=_shm_lib(), GM_NEGATIVE="1",
               GRAPHMAT_DEBUG_DROP_WAIT="1", GRAPHMAT_HIP_LIBRARY=hooks_so)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "multi_check.py")]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env, cwd=ROOT)
    text = out.stdout.decode()
    assert out.returncode == 0 and "NEGATIVE_CAUGHT" in text, text[-3000:]
    # ... with the dependency in place the same run equals the oracle (the positive tests above at scale 13 / 14) ...
    env.pop("GRAPHMAT_DEBUG_DROP_WAIT")
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env, cwd=ROOT)
    text = out.stdout.decode()
    assert out.returncode == 0 and "NEGATIVE_MISSED" in text, text[-3000:]
    # ... and the PRODUCT library has no such switch: the variable changes nothing there
    env.pop("GRAPHMAT_HIP_LIBRARY")
    env["GRAPHMAT_DEBUG_DROP_WAIT"] = "1"
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env, cwd=ROOT)
    text = out.stdout.decode()
    assert out.returncode == 0 and "NEGATIVE_MISSED" in text, text[-3000:]


def test_bench_two_ranks_distributed_build_and_native_exchange():
    """bench.py as the driver launches it at N > 1 (torch.distributed.run, one rank per process), on this 1-GPU box with
    gloo for torch.distributed and the shared-memory stand-in for librccl for the library's own communicator: every rank generates
    half of the edge list, the library shuffles the edges to their shards (gm_graph_desc_t.edges_local), the native
    exchange is cross-checked against the callback inside bench.py, and the JSON line says what ran.  The value must
    match a run that builds every shard from the whole edge list (GM_BENCH_BUILD=whole) in everything but timing."""
    import json
    from graphmat_amd import build
    build.build()
    lines = {}
    for mode in ("local", "whole"):
        env = dict(os.environ, GM_BENCH_BACKEND="gloo", GRAPHMAT_RCCL_LIBRARY=_shm_lib(), GM_BENCH_BUILD=mode)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "16", "--steps", "3",
               "--warmup", "1", "--cpu-scale", "0"]
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env, cwd=ROOT)
        text = out.stdout.decode()
        assert out.returncode == 0, (text + out.stderr.decode())[-3000:]
        js = [json.loads(l) for l in text.splitlines() if l.startswith("{")]
        assert len(js) == 1, text[-2000:]
        lines[mode] = js[0]
    a, b = lines["local"], lines["whole"]
    assert a["n_gpus"] == 2 and a["config"]["graph_build"].startswith("distributed") and "native exchange" in a["config"]["exchange"]
    assert b["config"]["graph_build"].startswith("every rank sorts")
    for k in ("E", "V", "rows_per_shard", "exchanged_rows_per_shard", "max_in_degree_rank0"):
        assert a["config"][k] == b["config"][k], k
    assert a["value"] > 0 and a["steps"] == 3


def test_bench_four_ranks_sharded_sweep_both_forms():
    """bench.py at N = 4 on this 1-GPU box (gloo + the shared-memory stand-in for librccl) with --col-tiles 3, so that the shards' device
    order is sliced and their rows go through the SHARDED SWEEP (gm_sweep_t.nsub = 4): the line must say so, carry both schedule forms
    (the sharded swept schedule, whose all-gather starts before the giant rows are folded, and the plain loop -- bench.py itself checks that
    they give the same bits) with the exchange time each exposes, and a roofline entry for the sweep kernel on rank 0's shard."""
    import json
    from graphmat_amd import build
    build.build()
    env = dict(os.environ, GM_BENCH_BACKEND="gloo", GRAPHMAT_RCCL_LIBRARY=_shm_lib())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--scale", "18", "--steps", "3",
           "--warmup", "1", "--cpu-scale", "0", "--col-tiles", "3"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env, cwd=ROOT)
    text = out.stdout.decode()
    assert out.returncode == 0, (text + out.stderr.decode())[-3000:]
    js = [json.loads(l) for l in text.splitlines() if l.startswith("{")]
    assert len(js) == 1, text[-2000:]
    j = js[0]
    assert j["n_gpus"] == 4 and "native exchange" in j["config"]["exchange"] and "sharded swe" in j["config"]["exchange"], j["config"]["exchange"]
    forms = j["multi_gpu"]["forms"]
    assert set(forms) == {"plain", "overlapped"} and all("exchange_ms_exposed" in f for f in forms.values())
    assert j["multi_gpu"]["overlapped_parts_total"] > 0  # (the sharded swept schedule ran: one part per iteration but the last)
    assert "disagrees" not in out.stderr.decode()
    assert j["roofline"] is not None and "k_spmv_sell" in j["roofline"]["kernel"]
