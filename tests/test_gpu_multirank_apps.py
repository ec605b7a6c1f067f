"""The reference's UNCHANGED application sources, one process per shard.

include/graphmat/mpi_single.h turns the applications' MPI_Init / MPI_Comm_rank / MPI_Barrier into the library's own
communicator (gm_dist_init_from_env), Graph<V,E> then holds one shard per rank and the messages travel through
gm_dist.hip.  RCCL wants one rank per GPU, so on the 1-GPU test box the ranks use the shared-memory test transport
(GRAPHMAT_RCCL_LIBRARY=tests/support/libgm_shm_transport.so, same entry points); a single rank runs over RCCL itself.  Expected values come from the
oracle with the matching layout parameter (the reference's id permutation depends on the number of ranks:
nparts = threads * 16 * nranks, include/Graph.h:117)."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_APPS = os.path.join(ROOT, "build", "ref_apps")


def _launch(exe, args, nranks, tmp_path, transport):
    procs = []
    rdv = str(tmp_path / "rendezvous")
    for r in range(nranks):
        env = dict(os.environ, GRAPHMAT_RANK=str(r), GRAPHMAT_NRANKS=str(nranks), GRAPHMAT_LOCAL_RANK="0", GRAPHMAT_RENDEZVOUS=rdv)
        if transport:  # "shm": the test suite's stand-in for librccl (tests/support/shm_transport.hip)
            from tests.support import build as shm_build
            env["GRAPHMAT_RCCL_LIBRARY"] = shm_build.build()
        procs.append(subprocess.Popen([exe] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            outs.append(out.decode())
    except subprocess.TimeoutExpired:
        for p in procs:  # a rank stuck in a collective would otherwise outlive the test
            if p.poll() is None:
                p.kill()
        raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    return outs


def _need(name):
    path = os.path.join(REF_APPS, name)
    if not os.path.exists(path):
        pytest.skip("%s was not prebuilt (reference tree absent at build time)" % name)
    return path


@pytest.mark.parametrize("nranks,transport", [(1, None), (2, "shm"), (3, "shm")])
def test_unchanged_pagerank_one_process_per_shard(golden_dir, tmp_path, nranks, transport):
    """src/PageRank.cpp: every vertex is printed by exactly one rank (its owner) and the values are the oracle's."""
    from graphmat_amd.mtx import read_mtx_bin
    from oracle import binding as ob
    fixture = os.path.join(golden_dir, "2_10_upper_triangle.bin.mtx")
    nv, s, d, v = read_mtx_bin(fixture)
    # one rank goes through the launcher path too (GRAPHMAT_NRANKS=1 leaves the process alone)
    outs = _launch(_need("PageRank"), [fixture], nranks, tmp_path, transport)
    og = ob.OracleGraph(nv, s, d, v, ref_threads=nranks)  # same nparts = 16 * nranks
    opr, oit, _ = og.pagerank(-1)
    odeg = og.degree()
    seen = {}
    for r, text in enumerate(outs):
        assert "Completed %d iterations" % oit in text, text[-2000:]
        for vid, deg, pr in re.findall(r"^(\d+) : (\d+) ([0-9.]+)$", text, flags=re.M):
            assert int(vid) not in seen, "vertex %s printed by two ranks" % vid
            seen[int(vid)] = (int(deg), pr)
    assert sorted(seen) == list(range(1, 26))
    for vid, (deg, pr) in seen.items():
        assert deg == int(odeg[vid - 1]) and pr == "%.6f" % opr[vid - 1], (vid, deg, pr)


@pytest.mark.parametrize("nranks,transport", [(2, "shm")])
def test_unchanged_bfs_one_process_per_shard(golden_dir, tmp_path, nranks, transport):
    """src/BFS.cpp: depths, parents (which depend on the rank count through the id permutation) and the reachable
    count (a map-reduce combined over the ranks) equal the oracle's."""
    from graphmat_amd.mtx import read_mtx_bin
    from oracle import binding as ob
    fixture = os.path.join(golden_dir, "2_10_upper_triangle.bin.mtx")
    nv, s, d, v = read_mtx_bin(fixture)
    outs = _launch(_need("BFS"), [fixture, 1], nranks, tmp_path, transport)
    od, op, oit, _ = ob.OracleGraph(nv, s, d, v, ref_threads=nranks).bfs(1)
    reach = int((od != 0xFFFFFFFF).sum())
    seen = {}
    assert "Reachable vertices = %d" % reach in outs[0], outs[0][-2000:]  # (printed by rank 0 only)
    for text in outs:
        assert "Completed %d iterations" % oit in text, text[-2000:]
        for vid, depth, parent in re.findall(r"^Depth (\d+) : (\d+) parent: (-?\d+)$", text, flags=re.M):
            assert int(vid) not in seen
            seen[int(vid)] = (int(depth), int(parent))
    assert len(seen) >= 10
    for vid, (depth, parent) in seen.items():
        assert depth == int(od[vid - 1]) and parent == int(np.int64(op[vid - 1])), (vid, depth, parent)


@pytest.mark.parametrize("nranks,transport", [(2, "shm")])
def test_unchanged_delta_stepping_one_process_per_shard(golden_dir, tmp_path, nranks, transport):
    """src/DeltaStepping.cpp with two ranks: two sharded graphs (light / heavy edges) share one vertex-property vector
    (Graph::shareVertexProperty is then a collective relayout: every rank's edges travel to the shard owning their row in
    the other graph's order).  Distances and the reachable count equal the SSSP oracle's."""
    from graphmat_amd.mtx import read_mtx_bin
    from oracle import binding as ob
    fixture = os.path.join(golden_dir, "2_10_upper_triangle.bin.mtx")
    nv, s, d, v = read_mtx_bin(fixture)
    dist, _ = ob.OracleGraph(nv, s, d, v, nranks).sssp(1)
    for delta in (10, 40):
        outs = _launch(_need("DeltaStepping"), [fixture, delta, 1], nranks, tmp_path, transport)
        assert "Reachable vertices = %d" % int((dist != 0xFFFFFFFF).sum()) in outs[0], outs[0][-2000:]
        seen = {}
        for text in outs:
            for vtx, dv in re.findall(r"^(\d+) : distance = (\d+|INF)$", text, flags=re.M):
                assert int(vtx) not in seen
                seen[int(vtx)] = dv
        assert len(seen) >= 20
        for vtx, dv in seen.items():
            exp = dist[vtx - 1]
            assert (dv == "INF" and exp == 0xFFFFFFFF) or int(dv) == exp, (vtx, dv, exp)


@pytest.mark.parametrize("nranks,transport", [(1, None), (2, "shm"), (3, "shm")])
def test_apply_to_all_edges_with_several_ranks(tmp_path, nranks, transport):
    """apps/sharded_edge_ops.cpp: applyToAllEdges (function pointer and device functor) when the other endpoint's
    property lives on another shard; every edge is checked by the rank owning its source row, then a weighted multiply
    reads the rewritten values."""
    exe = os.path.join(ROOT, "build", "apps", "sharded_edge_ops")
    if not os.path.exists(exe):
        pytest.skip("sharded_edge_ops was not prebuilt")
    outs = _launch(exe, [], nranks, tmp_path, transport)
    total = 0
    for r, text in enumerate(outs):
        m = re.search(r"SHARDEDEDGES rank (\d+) ok (\d+) edges of (\d+)", text)
        assert m, text[-2000:]
        total += int(m.group(2))
        want_total = int(m.group(3))
    assert total == want_total


@pytest.mark.parametrize("nranks,transport", [(2, "shm"), (3, "shm")])
def test_programs_that_change_in_do_every_iteration_with_several_ranks(tmp_path, nranks, transport):
    """apps/mutating_program.cpp with one process per shard: the sharded fixed-count runs take the two-stage schedule,
    which sends the next iteration's messages before do_every_iteration has run -- a program whose send_message depends
    on what do_every_iteration changes must get them re-sent (and the plain loop's fused apply + send must be dropped)."""
    exe = os.path.join(ROOT, "build", "apps", "mutating_program")
    if not os.path.exists(exe):
        pytest.skip("mutating_program was not prebuilt")
    outs = _launch(exe, [], nranks, tmp_path, transport)
    for r, text in enumerate(outs):
        assert "MUTATING PASS (rank %d of %d)" % (r, nranks) in text, text[-2500:]
        assert "fuse_apply_send=1: steady ok, every-time ok, sometimes ok" in text


def test_a_rank_that_waits_for_missing_peers_gets_an_error_not_a_hang(tmp_path):
    """gm_dist_init_from_env bounds the set-up of the communicator (GRAPHMAT_INIT_TIMEOUT): rank 0 of a declared world of
    two whose peer never starts must stop with a message, and a rank > 0 that finds only a STALE rendezvous file (older
    than the launch) must not take its id."""
    import time
    from tests.support import build as shm_build
    exe = os.path.join(ROOT, "build", "apps", "api_selftest")
    if not os.path.exists(exe):
        pytest.skip("api_selftest was not prebuilt")
    rdv = str(tmp_path / "rendezvous")
    env = dict(os.environ, GRAPHMAT_RANK="0", GRAPHMAT_NRANKS="2", GRAPHMAT_LOCAL_RANK="0", GRAPHMAT_RENDEZVOUS=rdv,
               GRAPHMAT_RCCL_LIBRARY=shm_build.build(), GRAPHMAT_INIT_TIMEOUT="3")
    t0 = time.time()
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=120)
    assert out.returncode == 1 and b"was not set up within 3 s" in out.stdout, out.stdout[-1500:]
    assert time.time() - t0 < 60
    # rank 1 with a stale file (written "long ago": the header carries rank 0's start time)
    import struct
    with open(rdv, "wb") as f:
        f.write(struct.pack("<8sqii", b"GMRDV01\0", int(time.time()) - 3600, 2, 0) + b"\0" * 128)
    env["GRAPHMAT_RANK"] = "1"
    t0 = time.time()
    try:
        out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=15)
        text = out.stdout
    except subprocess.TimeoutExpired as e:  # still politely waiting for a FRESH file (up to 60 s): what it should do
        text = e.stdout or b""
        out = None
    assert out is None or (out.returncode == 1 and b"never published a fresh" in text), text[-1500:]
