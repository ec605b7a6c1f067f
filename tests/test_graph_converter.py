"""Graph-file formats and the converter: every edge-list variant the reference's
loader/writer handles (include/GMDP/utils/edgelist.h:89-334), the clean-up steps of
include/GMDP/utils/edgelist_transformation.h, and the command line of src/graph_converter.cpp, checked
against numpy restatements of what each step means.  CPU-only except the graph-file round trip."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "apps", "graph_converter")


@pytest.fixture(scope="module")
def converter():
    from graphmat_amd import build
    build.build()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_apps
    build_apps.build_one("graph_converter")
    return EXE


def _run(exe, *args):
    out = subprocess.run([exe] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0, out.stdout.decode()
    return out.stdout.decode()


def _graph(seed=3, nv=40, ne=400):
    rng = np.random.default_rng(seed)
    src = rng.integers(1, nv + 1, ne).astype(np.int32)
    dst = rng.integers(1, nv + 1, ne).astype(np.int32)
    src[:20] = dst[:20]          # self loops
    src[20:60] = src[60:100]     # duplicates
    dst[20:60] = dst[60:100]
    val = rng.integers(1, 1000, ne).astype(np.uint32)
    return nv, src, dst, val


def test_every_file_variant_round_trips(tmp_path):
    from graphmat_amd import mtx
    nv, src, dst, _ = _graph()
    for dt in (np.int32, np.uint32, np.float32, np.float64):
        val = (np.arange(src.size) % 97 + 0.25).astype(dt) if np.dtype(dt).kind == "f" else (np.arange(src.size) % 97).astype(dt)
        for binary in (0, 1):
            for header in (0, 1):
                for weights in (0, 1):
                    p = tmp_path / "g"
                    mtx.write_edgelist(p, nv, nv, src, dst, val, binary=binary, header=header, weights=weights)
                    m, n, s2, d2, v2 = mtx.read_edgelist(p, binary=binary, header=header, weights=weights, val_dtype=dt)
                    assert (m, n) == ((nv, nv) if header else (int(src.max()), int(dst.max())))
                    assert np.array_equal(s2, src) and np.array_equal(d2, dst)
                    assert np.array_equal(v2, val if weights else np.ones_like(val))


def test_binary_variant_matches_the_golden_fixture_reader(golden_dir):
    from graphmat_amd import mtx
    path = os.path.join(golden_dir, "test.bin.mtx")
    nv, s, d, v = mtx.read_mtx_bin(path)
    m, n, s2, d2, v2 = mtx.read_edgelist(path, val_dtype=np.int32)
    assert max(m, n) == nv and np.array_equal(s, s2) and np.array_equal(d, d2) and np.array_equal(v, v2)


def test_text_header_count_governs_and_short_file_fails(tmp_path):
    from graphmat_amd import mtx
    p = tmp_path / "t"
    p.write_text("5 5 2\n1 2 7\n2 3 8\n3 4 9\n")
    m, n, s, d, v = mtx.read_edgelist(p, binary=False, val_dtype=np.int32)
    assert (m, n, list(s), list(d), list(v)) == (5, 5, [1, 2], [2, 3], [7, 8])
    p.write_text("5 5 4\n1 2 7\n2 3 8\n")
    with pytest.raises(RuntimeError):
        mtx.read_edgelist(p, binary=False, val_dtype=np.int32)
    with pytest.raises(RuntimeError):
        mtx.read_edgelist(tmp_path / "missing", binary=False)


def _read_out(path, binary=True, header=True, weights=True, dt=np.uint32):
    from graphmat_amd import mtx
    return mtx.read_edgelist(path, binary=binary, header=header, weights=weights, val_dtype=dt)


def test_default_conversion_text_to_binary_cleans_up(converter, tmp_path):
    """defaults: text in, binary out, self loops and duplicates removed (sorted by (src,dst), first value kept)"""
    from graphmat_amd import mtx
    nv, src, dst, val = _graph()
    mtx.write_edgelist(str(tmp_path / "in0"), nv, nv, src, dst, val, binary=False)
    _run(converter, tmp_path / "in", tmp_path / "out")
    m, n, s, d, v = _read_out(tmp_path / "out0")
    keep = src != dst
    ks, kd, kv = src[keep], dst[keep], val[keep]
    order = np.lexsort((kd, ks))  # stable: ties stay in input order
    ks, kd, kv = ks[order], kd[order], kv[order]
    first = np.ones(ks.size, bool)
    first[1:] = (ks[1:] != ks[:-1]) | (kd[1:] != kd[:-1])
    assert (m, n) == (nv, nv)
    assert np.array_equal(s, ks[first]) and np.array_equal(d, kd[first]) and np.array_equal(v, kv[first])


def test_keep_everything_is_the_identity(converter, tmp_path):
    from graphmat_amd import mtx
    nv, src, dst, val = _graph(5)
    mtx.write_edgelist(str(tmp_path / "in0"), nv, nv, src, dst, val, binary=True)
    _run(converter, "--selfloops", 1, "--duplicatededges", 1, "--inputformat", 0, "--outputformat", 1, tmp_path / "in", tmp_path / "out")
    m, n, s, d, v = _read_out(tmp_path / "out0", binary=False)
    assert np.array_equal(s, src) and np.array_equal(d, dst) and np.array_equal(v, val)
    text = (tmp_path / "out0").read_text().splitlines()
    assert text[0] == "%d %d %d" % (nv, nv, src.size) and text[1] == "%d %d %d" % (src[0], dst[0], val[0])


def test_bidirectional_and_uppertriangular(converter, tmp_path):
    from graphmat_amd import mtx
    nv, src, dst, val = _graph(7)
    mtx.write_edgelist(str(tmp_path / "in0"), nv, nv, src, dst, val, binary=True)
    _run(converter, "--inputformat", 0, "--bidirectional", "--selfloops", 1, "--duplicatededges", 1, tmp_path / "in", tmp_path / "bi")
    _, _, s, d, v = _read_out(tmp_path / "bi0")
    assert s.size == 2 * src.size
    assert np.array_equal(s[0::2], src) and np.array_equal(d[0::2], dst) and np.array_equal(s[1::2], dst) and np.array_equal(d[1::2], src)
    assert np.array_equal(v[0::2], val) and np.array_equal(v[1::2], val)
    _run(converter, "--inputformat", 0, "--uppertriangular", "--selfloops", 1, "--duplicatededges", 1, tmp_path / "in", tmp_path / "ut")
    _, _, s, d, v = _read_out(tmp_path / "ut0")
    assert np.array_equal(s, np.minimum(src, dst)) and np.array_equal(d, np.maximum(src, dst)) and np.array_equal(v, val)
    out = subprocess.run([converter, "--uppertriangular", "--bidirectional", str(tmp_path / "in"), str(tmp_path / "x")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode != 0 and b"Cannot be both" in out.stdout


def test_headerless_unweighted_input_and_float_weights(converter, tmp_path):
    (tmp_path / "in0").write_text("1 2\n2 3\n9 4\n")
    _run(converter, "--inputheader", 0, "--inputedgeweights", 0, "--outputedgeweights", 2, "--nvertices", 12,
         "--edgeweighttype", 2, "--outputformat", 1, tmp_path / "in", tmp_path / "out")
    lines = (tmp_path / "out0").read_text().splitlines()
    assert lines == ["12 12 3", "1 2 1.00000000", "2 3 1.00000000", "9 4 1.00000000"]
    (tmp_path / "w0").write_text("4 4 2\n1 2 0.5\n3 4 2.25\n")
    _run(converter, "--edgeweighttype", 1, "--outputformat", 1, tmp_path / "w", tmp_path / "wo")
    assert (tmp_path / "wo0").read_text().splitlines() == ["4 4 2", "1 2 0.500000000000000", "3 4 2.250000000000000"]


def test_randomize_ids_uses_the_reference_permutation(converter, tmp_path):
    """srand(5); r[i] = rand() % m drawn first; then swap slots i and r[i] in order (edgelist.h:336-366)."""
    from graphmat_amd import mtx
    nv, src, dst, val = _graph(11, nv=64, ne=300)
    mtx.write_edgelist(str(tmp_path / "in0"), nv, nv, src, dst, val, binary=True)
    _run(converter, "--inputformat", 0, "--selfloops", 1, "--duplicatededges", 1, "--randomizeID", tmp_path / "in", tmp_path / "out")
    _, _, s, d, v = _read_out(tmp_path / "out0")
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(5)
    pick = [libc.rand() % nv for _ in range(nv)]
    perm = list(range(nv))
    for i in range(nv):
        perm[i], perm[pick[i]] = perm[pick[i]], perm[i]
    perm = np.array(perm)
    assert sorted(perm) == list(range(nv))
    assert np.array_equal(s, perm[src - 1] + 1) and np.array_equal(d, perm[dst - 1] + 1) and np.array_equal(v, val)


def test_random_weights_stay_in_range(converter, tmp_path):
    from graphmat_amd import mtx
    nv, src, dst, val = _graph(13)
    mtx.write_edgelist(str(tmp_path / "in0"), nv, nv, src, dst, val, binary=True)
    _run(converter, "--inputformat", 0, "--selfloops", 1, "--duplicatededges", 1, "--outputedgeweights", 3, "--r", 16, tmp_path / "in", tmp_path / "out")
    _, _, s, d, v = _read_out(tmp_path / "out0")
    assert np.array_equal(s, src) and v.min() >= 1 and v.max() <= 16 and np.unique(v).size > 4


@pytest.mark.gpu
def test_graph_file_round_trip(converter, tmp_path):
    """--outputformat 2 / --inputformat 2: the GraphMat-bin role (Graph::WriteGraphMatBin / ReadGraphMatBin)."""
    from graphmat_amd import mtx
    nv, src, dst, val = _graph(17, nv=200, ne=3000)
    mtx.write_edgelist(str(tmp_path / "in0"), nv, nv, src, dst, val, binary=True)
    _run(converter, "--inputformat", 0, "--outputformat", 2, "--selfloops", 1, "--duplicatededges", 1, tmp_path / "in", tmp_path / "bin")
    _run(converter, "--inputformat", 2, "--outputformat", 0, "--selfloops", 1, "--duplicatededges", 1, tmp_path / "bin", tmp_path / "back")
    m, n, s, d, v = _read_out(tmp_path / "back0")
    assert (m, n, s.size) == (nv, nv, src.size)
    a = np.stack([src, dst, val.astype(np.int64)], 1)
    b = np.stack([s, d, v.astype(np.int64)], 1)
    assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
