"""The drop-in claim, end to end: application sources written against GraphMat's C++ surface
(the reference's own UNCHANGED src/PageRank.cpp, BFS.cpp, SGD.cpp, SSSP.cpp, compiled in the
build container where the reference tree exists, plus this project's apps/) run against
include/*.h + libgraphmat_hip.so and print the reference's golden outputs."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_APPS = os.path.join(ROOT, "build", "ref_apps")
OWN_APPS = os.path.join(ROOT, "build", "apps")


def test_apps_compile_for_gfx950():
    """CPU-side: hipcc --hipstdpar builds every app (the reference's too, when its tree is present)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from graphmat_amd import build
    build.build()
    import build_apps
    built = build_apps.build()
    assert any(p.endswith("label_propagation") for p in built) and any(p.endswith("bfs_bottom_up") for p in built)
    if os.path.isdir("/root/reference/src"):
        for app in ("PageRank", "BFS", "SGD", "SSSP"):
            assert os.path.exists(os.path.join(REF_APPS, app))


def _run(exe, *args, env=None):
    out = subprocess.run([exe] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                         env=(dict(os.environ, **env) if env else None))
    text = out.stdout.decode()
    assert out.returncode == 0, text
    return text


def _need(path):
    if not os.path.exists(path):
        pytest.skip("%s was not prebuilt (reference tree absent at build time)" % os.path.basename(path))
    return path


@pytest.fixture(scope="module")
def ref(golden_dir):
    return json.load(open(os.path.join(golden_dir, "reference_outputs.json")))


@pytest.mark.gpu
def test_reference_pagerank_app_unchanged(golden_dir, ref):
    g1 = ref["G1_pagerank_test_bin_mtx"]
    text = _run(_need(os.path.join(REF_APPS, "PageRank")), os.path.join(golden_dir, g1["file"]))
    assert text.count("Completed 1 iterations") == 1 and "Completed %d iterations" % g1["pagerank_iterations"] in text
    rows = re.findall(r"^(\d+) : (\d+) ([0-9.]+)$", text, flags=re.M)
    assert [int(r[1]) for r in rows] == g1["out_degree"]
    assert [r[2] for r in rows] == g1["pagerank_6dp"]


@pytest.mark.gpu
def test_reference_bfs_app_unchanged(golden_dir, ref):
    for key in ("G2_bfs_test_bin_mtx", "G2_bfs_2_10_upper_triangle"):
        g2 = ref[key]
        text = _run(_need(os.path.join(REF_APPS, "BFS")), os.path.join(golden_dir, g2["file"]), g2["source"])
        assert "Completed %d iterations" % g2["iterations"] in text
        assert "Reachable vertices = %d" % g2["reachable"] in text
        rows = re.findall(r"^Depth (\d+) : (\d+) parent: (-?\d+)$", text, flags=re.M)
        depth = g2.get("depth", g2.get("first10_depth"))[:10]
        parent = g2.get("parent", g2.get("first10_parent"))[:10]
        assert [int(r[1]) for r in rows] == depth
        assert [int(r[2]) for r in rows] == parent


@pytest.mark.gpu
def test_reference_sgd_app_unchanged(golden_dir, ref):
    g3 = ref["G3_sgd_ratings7"]
    text = _run(_need(os.path.join(REF_APPS, "SGD")), os.path.join(golden_dir, g3["file"]))
    rmse = re.findall(r"RMSE error = ([0-9.]+) per edge", text)
    assert rmse == [g3["rmse_before_6dp"], g3["rmse_after_6dp"]]


@pytest.mark.gpu
def test_reference_sssp_app_unchanged(golden_dir):
    from graphmat_amd.mtx import read_mtx_bin
    from oracle import binding as ob
    path = os.path.join(golden_dir, "2_10_upper_triangle.bin.mtx")
    text = _run(_need(os.path.join(REF_APPS, "SSSP")), path, 1)
    nv, s, d, v = read_mtx_bin(path)
    dist, it = ob.OracleGraph(nv, s, d, v, 1).sssp(1)
    assert "Completed %d iterations" % it in text
    assert "Reachable vertices = %d" % int((dist != 0xFFFFFFFF).sum()) in text
    got = re.findall(r"^(\d+) : distance = (\d+|INF)$", text, flags=re.M)
    assert len(got) == 25
    for vtx, dv in got:
        exp = dist[int(vtx) - 1]
        assert (dv == "INF" and exp == 0xFFFFFFFF) or int(dv) == exp


@pytest.mark.gpu
def test_own_generic_program_label_propagation(golden_dir, tmp_path):
    """An un-annotated user program outside the fixed menu: components equal scipy's."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from graphmat_amd import generators as gen
    from graphmat_amd.mtx import write_mtx_bin
    nv, s, d, v = gen.rmat_edges(11, 2, seed=9)  # sparse enough to have many components
    path = str(tmp_path / "g.bin.mtx")
    write_mtx_bin(path, nv, s, d, v)
    text = _run(_need(os.path.join(OWN_APPS, "label_propagation")), path)
    lab = np.array([int(x[1]) for x in re.findall(r"^component (\d+) (\d+)$", text, flags=re.M)])
    assert lab.size == nv
    ncomp, ref_lab = connected_components(coo_matrix((np.ones(len(s)), (s - 1, d - 1)), shape=(nv, nv)), directed=False)
    # same partition, and every label is the smallest vertex id of its component
    for c in range(ncomp):
        members = np.where(ref_lab == c)[0]
        assert (lab[members] == members.min() + 1).all()


@pytest.mark.gpu
def test_own_bfs_with_row_filter_trait(golden_dir, ref):
    """apps/bfs_bottom_up.cpp: a user-written BFS whose only extras are two traits: program_row_filter (bottom-up
    levels) and program_traits::reduce = REDUCE_LAST (the declared reduction strategy).  Golden depths/parents of G2, and the oracle
    on an RMAT graph large enough to have row-blocks, wave rows and top-down steps."""
    from graphmat_amd import generators as gen
    from graphmat_amd.mtx import write_mtx_bin
    from oracle import binding as ob
    ob.build()
    exe = _need(os.path.join(OWN_APPS, "bfs_bottom_up"))
    g2 = ref["G2_bfs_test_bin_mtx"]
    text = _run(exe, os.path.join(golden_dir, g2["file"]), g2["source"])
    rows = re.findall(r"^vertex (\d+) depth (\d+) parent (-?\d+)$", text, flags=re.M)
    assert [int(r[1]) for r in rows] == g2["depth"] and [int(r[2]) for r in rows] == g2["parent"]
    assert "Completed %d iterations" % g2["iterations"] in text
    nv, s, d, v = gen.rmat_edges(14, 16, seed=5)
    path = "/tmp/bfs_bottom_up_rmat14.bin.mtx"
    write_mtx_bin(path, nv, s, d, v)
    text = _run(exe, path, 3)
    got = {int(a): (int(b), int(c)) for a, b, c in re.findall(r"^vertex (\d+) depth (\d+) parent (-?\d+)$", text, flags=re.M)}
    od, op, oit, _ = ob.OracleGraph(nv, s, d, v, ref_threads=1).bfs(3)
    reached = np.where(od != 0xFFFFFFFF)[0]
    assert len(got) == reached.size and "Completed %d iterations" % oit in text
    for i in reached:
        exp_parent = -1 if i + 1 == 3 else int(op[i])
        assert got[i + 1] == (int(od[i]), exp_parent)


@pytest.mark.gpu
def test_untraited_programs_on_a_graph_with_giant_rows():
    """apps/untraited_programs.cpp: what an UNCHANGED application gets by default -- programs without program_traits, i.e. the
    ordered fold for every reduce_function -- on RMAT-16 (hub rows of 13 K in-edges: giant rows, whose ordered fold runs in
    two passes, k_giant_terms + k_giant_fold_ordered): BFS with 8-byte a = b messages over a sparse message vector, SSSP with a
    4-byte min over hashed edge weights, and PageRank's float sum over a dense vector, each against the oracle -- depths and
    parents, distances, out-degrees and fp32 bits of every vertex."""
    from graphmat_amd import generators as gen
    from graphmat_amd.mtx import write_mtx_bin
    from oracle import binding as ob
    ob.build()
    exe = _need(os.path.join(OWN_APPS, "untraited_programs"))
    nv, s, d, v = gen.rmat_edges(16, 16, seed=11, weights="hash")
    path = "/tmp/untraited_rmat16.bin.mtx"
    write_mtx_bin(path, nv, s, d, v)
    text = _run(exe, path, 5, 6)
    og = ob.OracleGraph(nv, s, d, v, ref_threads=1)
    od, op, oit, _ = og.bfs(5)
    got = {int(a): (int(b), int(c)) for a, b, c in re.findall(r"^bfs (\d+) (\d+) (-?\d+)$", text, flags=re.M)}
    reached = np.where(od != 0xFFFFFFFF)[0]
    assert len(got) == reached.size and reached.size > nv // 4
    for i in reached:
        assert got[i + 1] == (int(od[i]), -1 if i + 1 == 5 else int(op[i])), i + 1
    odist, _ = og.sssp(5)
    gots = {int(a): int(b) for a, b in re.findall(r"^sssp (\d+) (\d+)$", text, flags=re.M)}
    rs = np.where(odist != 0xFFFFFFFF)[0]
    assert len(gots) == rs.size
    for i in rs:
        assert gots[i + 1] == int(odist[i]), i + 1
    opr, oit2, _ = og.pagerank(6)
    odeg = og.degree()
    gotp = re.findall(r"^pr (\d+) (\d+) ([0-9a-f]{8})$", text, flags=re.M)
    assert len(gotp) == nv and oit2 == 6
    bits = np.array([int(b, 16) for _, _, b in gotp], dtype=np.uint32)
    degs = np.array([int(x) for _, x, _ in gotp], dtype=np.int64)
    assert (degs == odeg).all()
    assert (bits == np.ascontiguousarray(opr, np.float32).view(np.uint32)).all()
    # The GUIDED PULL (engine.hpp; round 6): on large graphs the levels of an undeclared ACTIVE_ONLY program whose active set owns few
    # out-edges only fold the rows that set reaches (marked first, then the same kernels in the same order).  Forced here on the small
    # graph (guided_pull = 2): every line of the output must be the same, and the path must really have been taken.
    text2 = _run(exe, path, 5, 6, env={"GRAPHMAT_OPTIONS": "guided_pull=2", "GRAPHMAT_VERBOSE": "1"})
    assert "guided pull: " in text2 and text2.count("   guided pull:") >= 4, text2[:2000]
    keep = lambda t: sorted(l for l in t.splitlines() if re.match(r"^(bfs|sssp|pr) ", l))
    assert keep(text2) == keep(text)
    text0 = _run(exe, path, 5, 6, env={"GRAPHMAT_OPTIONS": "guided_pull=0", "GRAPHMAT_VERBOSE": "1"})
    assert "guided pull" not in text0 and keep(text0) == keep(text)
    # ... and on a graph with slices (forced here: GRAPHMAT_COL_TILES) the undeclared SSSP's sparse message vector goes through the sweep
    # (k_spmv_sell_sparse: 4-byte messages, int edge values in the structure, the program's own min as an ordered fold): same lines again
    text3 = _run(exe, path, 5, 6, env={"GRAPHMAT_COL_TILES": "3", "GRAPHMAT_VERBOSE": "1", "GRAPHMAT_OPTIONS": "sweep_form=64"})
    assert "sparse message vector through the sweep" in text3 and keep(text3) == keep(text)


@pytest.mark.gpu
def test_cpp_surface_selftest():
    """apps/api_selftest.cpp: Graph<V,E> get/set, getEdgelist, applyToAll*, activity, a (mul,add)
    SpMV and chain BFS through include/*.h -- the reference's unit tests re-expressed."""
    text = _run(_need(os.path.join(OWN_APPS, "api_selftest")))
    assert "SELFTEST PASS" in text, text


# ---- three more of the reference's applications, unchanged ---------------
@pytest.mark.gpu
def test_reference_topological_sort_app_unchanged(golden_dir):
    """src/TopologicalSort.cpp on the DAG fixture: order = Kahn level (1 + max over in-neighbours)."""
    from graphmat_amd.mtx import read_mtx_bin
    path = os.path.join(golden_dir, "2_10_upper_triangle.bin.mtx")
    text = _run(_need(os.path.join(REF_APPS, "TopologicalSort")), path)
    nv, s, d, v = read_mtx_bin(path)
    level = np.zeros(nv + 1, np.int64)
    for a, b in sorted(zip(s.tolist(), d.tolist())):  # src < dst in this fixture: one pass in src order suffices
        level[b] = max(level[b], level[a] + 1)
    got = re.findall(r"^Top Sort order (\d+) : (\d+)$", text, flags=re.M)
    assert len(got) == 10
    assert [int(o) for _, o in got] == level[1:11].tolist()


@pytest.mark.gpu
def test_reference_delta_stepping_app_unchanged(golden_dir):
    """src/DeltaStepping.cpp: two graphs (light/heavy edges) sharing one vertex-property vector
    (Graph::shareVertexProperty); distances equal the SSSP oracle's."""
    from graphmat_amd.mtx import read_mtx_bin
    from oracle import binding as ob
    path = os.path.join(golden_dir, "2_10_upper_triangle.bin.mtx")
    nv, s, d, v = read_mtx_bin(path)
    dist, _ = ob.OracleGraph(nv, s, d, v, 1).sssp(1)
    for delta in (10, 40):
        text = _run(_need(os.path.join(REF_APPS, "DeltaStepping")), path, delta, 1)
        assert "Reachable vertices = %d" % int((dist != 0xFFFFFFFF).sum()) in text
        got = re.findall(r"^(\d+) : distance = (\d+|INF)$", text, flags=re.M)
        assert len(got) == 25
        for vtx, dv in got:
            exp = dist[int(vtx) - 1]
            assert (dv == "INF" and exp == 0xFFFFFFFF) or int(dv) == exp


@pytest.mark.gpu
def test_reference_incremental_pagerank_app_unchanged(golden_dir):
    """src/IncrementalPageRank.cpp (fp64 deltas, ACTIVE_ONLY): printed ranks equal a numpy
    restatement of the same recurrence."""
    from graphmat_amd.mtx import read_mtx_bin
    path = os.path.join(golden_dir, "test.bin.mtx")
    text = _run(_need(os.path.join(REF_APPS, "IncrementalPageRank")), path)
    nv, s, d, v = read_mtx_bin(path)
    outdeg = np.bincount(s - 1, minlength=nv)
    delta = np.full(nv, 0.3)
    pr = np.full(nv, 0.3)
    active = np.ones(nv, bool)
    alpha = 0.3
    iters = 0
    while True:
        msg = np.where(outdeg > 0, delta / np.maximum(outdeg, 1), 0.0)
        y = np.zeros(nv)
        got = np.zeros(nv, bool)
        for a, b in zip(s - 1, d - 1):
            if active[a]:
                y[b] += msg[a]
                got[b] = True
        old = pr.copy()
        for i in np.where(got)[0]:
            if abs(delta[i]) > 1e-8:
                delta[i] = 0.0
            delta[i] += (1.0 - alpha) * y[i]
            if abs(delta[i]) > 1e-8:
                pr[i] += delta[i]
        active = np.abs(pr - old) > 1e-8
        iters += 1
        if not active.any():
            break
    rows = re.findall(r"^(\d+) : (\d+) ([0-9.]+)$", text, flags=re.M)
    assert [int(r[1]) for r in rows] == outdeg.tolist()
    assert [r[2] for r in rows] == ["%f" % x for x in pr]
    assert "Completed %d iterations" % iters in text


@pytest.mark.gpu
def test_sparse_float_sum_exactness_app():
    """apps/active_float_sum.cpp: REDUCE_F32_ADD user program with a sparse message vector; giant,
    wave and short rows all equal a host fold in native order, bit for bit."""
    text = _run(_need(os.path.join(OWN_APPS, "active_float_sum")))
    assert "FLOATSUM PASS" in text, text[-1500:]


@pytest.mark.gpu
def test_last_writer_program_with_changing_senders():
    """apps/last_writer.cpp: an a=b program whose active vertices are rewritten in the step they send in
    (list-based top-down steps must use the messages of BEFORE the step); equals a host restatement."""
    text = _run(_need(os.path.join(OWN_APPS, "last_writer")))
    assert "LASTWRITER PASS" in text, text[-1500:]


@pytest.mark.gpu
def test_unmirrorable_globals_are_fatal(tmp_path):
    """include/graphmat/device_globals.hpp: an application whose image cannot be parsed (here: a stripped copy
    of the reference-style BFS) must not run with zeroed device copies of its host globals: message + exit(1);
    GRAPHMAT_ALLOW_UNMIRRORED_GLOBALS=1 downgrades that to a warning."""
    import shutil
    src = _need(os.path.join(OWN_APPS, "bfs_bottom_up"))
    exe = str(tmp_path / "bfs_stripped")
    shutil.copy(src, exe)
    if subprocess.run(["strip", exe]).returncode != 0:
        pytest.skip("strip is not available")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "graphmat_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    fixture = os.path.join(ROOT, "tests", "golden", "test.bin.mtx")
    out = subprocess.run([exe, fixture, "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, env=env)
    assert out.returncode == 1 and b"cannot mirror host namespace-scope variables" in out.stdout, out.stdout[-800:]
    env["GRAPHMAT_ALLOW_UNMIRRORED_GLOBALS"] = "1"
    out = subprocess.run([exe, fixture, "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, env=env)
    assert b"warning" in out.stdout


@pytest.mark.gpu
def test_probed_reduce_strategies_are_cross_checked():
    """apps/reduce_probe_cases.cpp: unannotated programs whose reduce_function is not float a+b (saturating add,
    float max, a+b with a quirk at one value) must end up with the ordered fold -- by the probe's adversarial
    operands or by the device-side cross-check -- and every result must equal a host fold with the program's function."""
    exe = _need(os.path.join(OWN_APPS, "reduce_probe_cases"))
    # without the opt-in nothing is probed: every unannotated program gets the ordered fold (exact by default)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=dict(os.environ, GRAPHMAT_VERBOSE="1"))
    text = out.stdout.decode()
    assert out.returncode == 0 and "PROBECASES PASS" in text, text[-3000:]
    assert "reduce strategy 3" not in text and "reduce strategy 2" not in text and text.count("reduce strategy 0") >= 4
    for tiles in ("1", "3"):  # untiled, and with column tiles (the cross-check then compares whole rows after the last tile)
        out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                             # (sweep_form bit 8: with slices the short rows of these small graphs ride the sweep and are folded by k_short_fold with the programs' own functions)
                             env=dict(os.environ, GRAPHMAT_VERBOSE="1", GRAPHMAT_COL_TILES=tiles, GRAPHMAT_TRUST_PROBE="1", GRAPHMAT_OPTIONS="sweep_form=256"))
        text = out.stdout.decode()
        assert out.returncode == 0 and "PROBECASES PASS" in text, text[-3000:]
        sect = {name: body for name, body in re.findall(r"== (\w+)\n(.*?)(?=\n== |\nPROBECASES)", text, flags=re.S)}
        assert "reduce strategy 3" in sect["PlainAdd"] and "0 mismatching rows" in sect["PlainAdd"]
        assert "reduce strategy 0" in sect["SatAdd"] and "reduce strategy 0" in sect["FloatMax"]
        assert "reduce strategy 3" in sect["QuirkAdd"] and "using the ordered fold" in sect["QuirkAdd"]
        for name in sect:
            assert "results-ok" in sect[name], (tiles, name)


@pytest.mark.gpu
def test_reference_unit_test_closed_forms_on_the_device():
    """apps/closed_forms.cpp: the closed forms of the reference's unit tests for this path (test/test_reduce.cpp:39-65
    MapReduce = 2000, test/test_apply_edges.cpp:39-112 val = src + s*dst on identity and random graphs,
    test/test_graph_basics.cpp:56-81 set/get through the permutation), function-pointer and device-functor forms."""
    text = _run(_need(os.path.join(OWN_APPS, "closed_forms")))
    assert "CLOSEDFORMS PASS" in text, text[-2000:]
    lines = re.findall(r"^CLOSED (\w+) (\w+)$", text, flags=re.M)
    assert len(lines) == 10 and all(v == "ok" for _, v in lines), lines


@pytest.mark.gpu
def test_fused_apply_and_send_respects_do_every_iteration():
    """apps/mutating_program.cpp: the apply pass also writes the next iteration's messages (k_apply_send); they may be
    used only while do_every_iteration leaves the program unchanged.  Programs that never / always / sometimes change
    what send_message computes, against a host evaluation of the reference's loop, with the fusion on and off."""
    text = _run(_need(os.path.join(OWN_APPS, "mutating_program")))
    assert "MUTATING PASS" in text, text[-2000:]
    assert "fuse_apply_send=1: steady ok, every-time ok, sometimes ok" in text and "fuse_apply_send=0: steady ok" in text


@pytest.mark.gpu
def test_edge_updates_and_shared_properties_on_tiled_graphs():
    """apps/tiled_edge_update.cpp: applyToAllEdges (device functor and host function pointer) on a graph whose multiply
    runs on column tiles -- large graphs are tiled automatically, so the functor form must not refuse them and the tile
    copies of the edge values must follow -- and shareVertexProperty between a tiled graph and graphs whose vertices
    with edges are / are not a subset of its own (gm_graph_relayout_like keeps the tiles only when they stay contiguous
    native ranges)."""
    text = _run(_need(os.path.join(OWN_APPS, "tiled_edge_update")))
    assert "TILEDEDGES PASS" in text, text[-2000:]
    assert re.search(r"graph 1: [2-9] column tiles", text)
    assert re.search(r"B1 \(not a subset\) 1, B2 \(subset\) [2-9]", text)


@pytest.mark.gpu
def test_edge_values_through_the_sweep():
    """apps/swept_edge_values.cpp: a weighted float SpMV (process_message = message * edge value) on a graph with medium, long and
    giant rows whose device order is sliced: the rows above 64 edges go through the row-stationary sweep WITH edge values
    (k_spmv_sell, gm_sweep_t.sval / lval / gval), also after applyToAllEdges rewrote the values (device functor and host
    function pointer: gm_graph_sync_tile_vals brings the sweep's copies over) and on a graph relayouted by shareVertexProperty;
    every vertex against a host evaluation."""
    text = _run(_need(os.path.join(OWN_APPS, "swept_edge_values")))
    assert "SWEPTEDGES PASS" in text, text[-2000:]
    assert re.search(r"sweep: [1-9][0-9]* rows \([1-9][0-9]* long\), value bytes 4, [1-9][0-9]* giant-row edges gathered by the sweep", text), text[-2000:]
    # ... and with the rows of at most 64 edges riding the sweep too (gm_sweep_t.nstream; sweep_form bit 8: on a graph of any size): their stream groups
    # carry the edge values, the rewritten ones after gm_graph_sync_tile_vals, also on the relayouted graph
    text = _run(_need(os.path.join(OWN_APPS, "swept_edge_values")), env={"GRAPHMAT_OPTIONS": "sweep_form=256", "GRAPHMAT_VERBOSE": "1"})
    assert "SWEPTEDGES PASS" in text and "ride the sweep (k_short_fold)" in text, text[-2000:]


@pytest.mark.gpu
def test_edge_values_through_the_column_blocked_stream():
    """apps/swept_edge_values.cpp blocked: the same checks with the rows of at most 64 edges in the column-blocked stream WITH their edge
    values (k_spmv_blocked<HAS_VALS>, gm_blocked_t.eval / epos; the sweep and the giant rows' passes in front of it): the values the
    stream reads after applyToAllEdges rewrote them, and on the relayouted graphs; every vertex against a host evaluation."""
    exe = _need(os.path.join(OWN_APPS, "swept_edge_values"))
    out = subprocess.run([exe, "blocked"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=dict(os.environ, GRAPHMAT_VERBOSE="1"))
    text = out.stdout.decode()
    assert out.returncode == 0 and "SWEPTEDGES PASS" in text, text[-2000:]
    assert re.search(r"column-blocked stream: [1-9][0-9]* short rows, [1-9][0-9]* entries, value bytes 4", text), text[-2000:]
    assert "the short rows take the column-blocked stream" in text and "with their edge values" in text, text[-2000:]


@pytest.mark.gpu
def test_reference_pagerank_timing_build_prints_the_per_iteration_lines(golden_dir, ref):
    """The reference's tracing flavour (-D__TIMING) of the UNCHANGED src/PageRank.cpp: per iteration the phase lines and
    "Iteration %d :: %f msec :: updated %d vertices :: changed %d vertices" (include/GraphMatRuntime.h:150-248 of the
    reference).  The counts of the PageRank run are G1's changed_per_iteration; "updated" = vertices with an in-edge;
    the Degree pass before it changes nothing under PR's operator!= (SURVEY section 8 note 3)."""
    g1 = ref["G1_pagerank_test_bin_mtx"]
    text = _run(_need(os.path.join(REF_APPS, "PageRank__TIMING")), os.path.join(golden_dir, g1["file"]))
    lines = re.findall(r"^Iteration (\d+) :: ([0-9.]+) msec :: updated (-?\d+) vertices :: changed (-?\d+) vertices", text, flags=re.M)
    n_deg, n_pr = g1["degree_iterations"], g1["pagerank_iterations"]
    assert len(lines) == n_deg + n_pr, text[-2000:]
    assert [int(l[0]) for l in lines] == list(range(n_deg)) + list(range(n_pr))
    assert [int(l[3]) for l in lines[n_deg:]] == g1["changed_per_iteration"]
    assert [int(l[3]) for l in lines[:n_deg]] == [0] * n_deg
    from graphmat_amd.mtx import read_mtx_bin
    nv, s, d, v = read_mtx_bin(os.path.join(golden_dir, g1["file"]))
    with_in_edge = len(set(d.tolist()))
    with_out_edge = len(set(s.tolist()))
    assert [int(l[2]) for l in lines[n_deg:]] == [with_in_edge] * n_pr
    assert [int(l[2]) for l in lines[:n_deg]] == [with_out_edge] * n_deg   # Degree runs over IN_EDGES: rows = sources
    for name in ("Send message time", "SPMV time", "Apply time", "Do every iteration time"):
        assert text.count(name + " = ") == n_deg + n_pr, name
    rows = re.findall(r"^(\d+) : (\d+) ([0-9.]+)$", text, flags=re.M)
    assert [r[2] for r in rows] == g1["pagerank_6dp"]


@pytest.mark.gpu
def test_undeclared_float_sums_speculated_and_proven():
    """apps/speculated_float_sum.cpp: giant rows of programs that declare nothing.  A reduce_function that answers like a float addition
    gets the exact parallel replay of the sum, every 8192-product chunk of which is proven with the program's own function
    (kernels.hpp: k_giant_verify_chunks); one that answers like an addition but is none on the data at hand must be caught by that proof
    and folded in order; results compared bit for bit with a host fold in the reference's order (SPMV.h:54-59)."""
    text = _run(_need(os.path.join(OWN_APPS, "speculated_float_sum")))
    assert "SPECULATED PASS" in text, text[-2000:]

