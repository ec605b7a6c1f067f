"""Column tiles (graphmat_hip.h gm_graph_tile): the tiled multiply must give the untiled bits.

A tile holds the edges whose column lies in one contiguous NATIVE range, so folding a row's tiles
one after the other, carrying the running value in y, is the reference's ascending-native-column
fold (include/GMDP/singlenode/spmspv.h:49-81 over the DCSC order of matrices/DCSCTile.h:41-58).
Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphmat_amd import generators as gen


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from graphmat_amd import build
    build.build()
    from graphmat_amd import api
    from oracle import binding as ob
    return api, ob


def f32bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture
def tile_min(env):
    """sets the tiling threshold for one test (rows of more edges are tiled), restores the default"""
    api, _ = env

    def set_(v):
        api._lib.check(api._lib.lib().gm_set_option(b"tile_min_row", v))
    yield set_
    set_(64)


@pytest.mark.parametrize("tiles,threads,minrow", [(2, 1, 64), (5, 3, 64), (8, 1, 200), (64, 2, 64)])
def test_tile_structure(env, tile_min, tiles, threads, minrow):
    """device order = (tile, degree rank); every tile CSR holds exactly the edges of the long rows
    whose column lies in the tile's slice, in the untiled order."""
    api, _ = env
    tile_min(minrow)
    nv, s, d, v = gen.rmat_edges(12, 16, 3, weights="hash")
    g = api.Graph(nv, s, d, v, ref_threads=threads, col_tiles=tiles)
    T = g.col_tiles
    assert 1 < T <= tiles
    don, nod = g.maps_to_host()
    nat = api.native_index(nv, threads * 16)
    sn, dn = nat[s - 1], nat[d - 1]
    deg = np.bincount(sn, minlength=nv) + np.bincount(dn, minlength=nv)
    nlive = int((deg > 0).sum())
    assert (deg[nod[:nlive]] > 0).all() and (deg[nod[nlive:]] == 0).all()
    rp, ci, vv = g.csr_to_host(api.GM_DIR_OUT)
    rowlen = np.diff(rp)
    import torch
    seen = 0
    prev_rows = np.zeros(nv, bool)
    prev_native_max = None
    gathers = []
    # the device order may be cut finer than the tiles (gm_graph_sweep: a tile = k consecutive slices)
    import ctypes as C
    from graphmat_amd import _lib
    sw = _lib.Sweep()
    assert g.L.gm_graph_sweep(g.h, C.byref(sw)) == 0
    cuts = None
    if sw.nslices > 0:
        assert sw.nslices % T == 0 and sw.nslices <= 128
        cuts = np.zeros(sw.nslices + 1, np.int32)
        api.copy_from_device(cuts, sw.slice_base)
        assert cuts[0] == 0 and cuts[-1] == nlive and (np.diff(cuts) >= 0).all()
    for t in range(T):
        c, prev = g.tile(api.GM_DIR_OUT, t)
        lo, hi = c.hot_base, c.hot_base + c.hot_len
        assert lo == (0 if t == 0 else last_hi) and hi <= nlive
        last_hi = hi
        # a slice is a contiguous native range, busiest first (a slice may be empty: they are cut by gathers served,
        # and one hub can outweigh a slice's share on a small graph)
        k = sw.nslices // T if cuts is not None else 1
        bounds = cuts[t * k: (t + 1) * k + 1] if cuts is not None else np.array([lo, hi])
        assert bounds[0] == lo and bounds[-1] == hi
        for a, b in zip(bounds[:-1], bounds[1:]):
            natives = nod[a:b]
            if natives.size:
                if prev_native_max is not None:
                    assert natives.min() > prev_native_max
                prev_native_max = natives.max()
                assert (np.diff(deg[natives]) <= 0).all()
        # expected content: long rows' edges with column in [lo, hi), untiled order
        trp = np.zeros(c.nrows + 1, np.int64)
        tci = np.zeros(max(c.nnz, 1), np.int32)
        tvv = np.zeros(max(c.nnz, 1), np.int32)
        api.copy_from_device(trp, c.rowptr)
        if c.nnz:
            api.copy_from_device(tci[: c.nnz], c.colidx)
            api.copy_from_device(tvv[: c.nnz], c.vals)
        rows_of_edge = np.repeat(np.arange(nv), rowlen)
        keep = (rowlen[rows_of_edge] > minrow) & (ci >= lo) & (ci < hi)
        assert c.nnz == int(keep.sum())
        assert (tci[: c.nnz] == ci[keep]).all() and (tvv[: c.nnz] == vv[keep]).all()
        assert (np.diff(trp) == np.bincount(rows_of_edge[keep], minlength=nv)).all()
        bits = np.zeros((nv + 31) // 32 + 2, np.uint32)
        api.copy_from_device(bits, prev)
        got = ((bits[np.arange(nv) >> 5] >> (np.arange(nv) & 31)) & 1).astype(bool)
        assert (got == prev_rows).all()
        prev_rows |= np.diff(trp) > 0
        seen += c.nnz
        gathers.append(int(((ci >= lo) & (ci < hi)).sum()))  # edges of ALL rows whose column lies in the tile
    assert last_hi == nlive
    assert seen == int(rowlen[rowlen > minrow].sum())
    # tiles serve about equally many gathers: none exceeds its share by more than the heaviest column
    colw = np.bincount(ci, minlength=nv)
    assert max(gathers) <= len(ci) / T + colw.max()
    # the wave rows that stay untiled: more than 64 and at most minrow edges, long ones first
    c = g.csr(api.GM_DIR_OUT)
    assert c.tile_min_row == minrow
    um = np.zeros(c.numid, np.int32)
    api.copy_from_device(um, c.umid_row)
    want = np.nonzero((rowlen > 64) & (rowlen <= minrow))[0]
    assert sorted(um.tolist()) == want.tolist()
    assert (rowlen[um[: c.numid_long]] > 1024).all() and (rowlen[um[c.numid_long:]] <= 1024).all()


@pytest.mark.parametrize("scale,threads,tiles,minrow", [(10, 1, 2, 64), (12, 4, 3, 64), (14, 1, 8, 100), (16, 2, 4, 64),
                                                         (16, 1, 16, 1024), (18, 1, 4, 1024), (14, 2, 4, 0), (17, 1, 8, 0)])
def test_pagerank_tiled_bit_exact(env, tile_min, scale, threads, tiles, minrow):
    api, ob = env
    tile_min(minrow)
    nv, s, d, v = gen.rmat_edges(scale, 16, seed=scale)
    og = ob.OracleGraph(nv, s, d, v, threads)
    g = api.Graph(nv, s, d, v, ref_threads=threads, col_tiles=tiles)
    assert g.col_tiles > 1
    opr, oit, _ = og.pagerank(10)
    for force in (0, 1):
        api._lib.lib().gm_set_option(b"force_ordered", force)
        pr, deg, it = g.pagerank(10)
        assert (deg == og.degree()).all() and it == oit == 10
        assert (f32bits(pr) == f32bits(opr)).all(), "tiled pagerank bits differ (force_ordered=%d)" % force
    api._lib.lib().gm_set_option(b"force_ordered", 0)
    pr, _, it = g.pagerank(-1)
    opr, oit, _ = og.pagerank(-1)
    assert it == oit and (f32bits(pr) == f32bits(opr)).all()


def test_tile_threshold_must_be_zero_or_at_least_the_short_row_limit(env):
    api, _ = env
    L = api._lib.lib()
    assert L.gm_set_option(b"tile_min_row", 7) != 0   # rows of 8..64 edges would be multiplied twice
    assert L.gm_set_option(b"tile_min_row", 0) == 0 and L.gm_set_option(b"tile_min_row", 64) == 0
    assert L.gm_set_option(b"tile_min_row", 1024) == 0


def test_other_programs_on_a_tiled_graph(env, tile_min):
    """BFS / SSSP / SGD do not use the tiles but run on the tiled device order."""
    api, ob = env
    tile_min(64)
    nv, s, d, v = gen.rmat_edges(13, 16, seed=77, weights="hash")
    og = ob.OracleGraph(nv, s, d, v, 2)
    g = api.Graph(nv, s, d, v, ref_threads=2, col_tiles=4)
    depth, parent, it = g.bfs(1)
    od, op, oit, _ = og.bfs(1)
    assert it == oit and (depth == od).all() and (parent == op).all()
    dist, it = g.sssp(1)
    odist, oit = og.sssp(1)
    assert it == oit and (dist == odist).all()


@pytest.mark.parametrize("tiles,minrow", [(3, 64), (8, 200)])
def test_edge_value_updates_reach_the_tiles(env, tile_min, tiles, minrow):
    """The column tiles hold copies of the edge values: gm_graph_set_vals (Graph::applyToAllEdges, host form) and
    gm_graph_sync_tile_vals (after an in-place rewrite on the device, the functor form) must bring them up to date."""
    api, _ = env
    import ctypes as C
    tile_min(minrow)
    nv, s, d, v = gen.rmat_edges(12, 16, 5, weights="hash")
    g = api.Graph(nv, s, d, v, ref_threads=2, col_tiles=tiles)
    assert g.col_tiles > 1
    L = api._lib.lib()
    rp, ci, vv = g.csr_to_host(api.GM_DIR_OUT)
    rowlen = np.diff(rp)
    rows_of_edge = np.repeat(np.arange(nv), rowlen)

    def tiles_hold(expected):
        for t in range(g.col_tiles):
            c, _ = g.tile(api.GM_DIR_OUT, t)
            if not c.nnz:
                continue
            tvv = np.zeros(c.nnz, np.int32)
            api.copy_from_device(tvv, c.vals)
            keep = (rowlen[rows_of_edge] > minrow) & (ci >= c.hot_base) & (ci < c.hot_base + c.hot_len)
            assert c.nnz == int(keep.sum())
            if not (tvv == expected[keep]).all():
                return False
        return True
    assert tiles_hold(vv)
    # host form: new values for every edge of the direction
    new = (vv * 3 + 1).astype(np.int32)
    api._lib.check(L.gm_graph_set_vals(g.h, api.GM_DIR_OUT, new.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(g.csr_to_host(api.GM_DIR_OUT)[2], new) and tiles_hold(new)
    # device form: rewrite the library's array in place, then sync
    newer = (new ^ 0x55).astype(np.int32)
    api.copy_to_device(g.csr(api.GM_DIR_OUT).vals, newer)
    assert not tiles_hold(newer)
    api._lib.check(L.gm_graph_sync_tile_vals(g.h, None))
    assert tiles_hold(newer)


@pytest.mark.parametrize("form", [2])
def test_persistent_wave16_forms_bit_exact(env, tile_min, form):
    """The persistent forms of the 16-rows-per-wave kernel (large LDS hot set loaded once per workgroup,
    gm_set_option("wave16_form")) fold exactly like the one-workgroup-per-64-rows form: tiled and untiled
    PageRank against the oracle, bit for bit."""
    api, ob = env
    tile_min(64)
    L = api._lib.lib()
    nv, s, d, v = gen.rmat_edges(16, 16, seed=5)
    og = ob.OracleGraph(nv, s, d, v, 1)
    opr, oit, _ = og.pagerank(6)
    try:
        api._lib.check(L.gm_set_option(b"wave16_form", 16 + form))  # (+16: also for graphs this small)
        for tiles in (1, 4):
            g = api.Graph(nv, s, d, v, ref_threads=1, col_tiles=tiles)
            pr, deg, it = g.pagerank(6)
            assert it == oit == 6 and (f32bits(pr) == f32bits(opr)).all(), "form %d, %d tiles" % (form, tiles)
            g.close()
    finally:
        L.gm_set_option(b"wave16_form", 2)


@pytest.mark.parametrize("form", [4])
def test_persistent_rowwave_forms_bit_exact(env, tile_min, form):
    """Row-blocks taken by the waves of persistent workgroups that share a large LDS hot set (kernels.hpp:
    k_spmv_rowwave, gm_set_option("rowwave_form")) fold exactly like k_spmv_rowblock: tiled and untiled PageRank
    (fixed count and until convergence) against the oracle, bit for bit; a graph with edge values too."""
    api, ob = env
    tile_min(64)
    L = api._lib.lib()
    try:
        api._lib.check(L.gm_set_option(b"rowwave_form", 16 + form))  # (+16: also for graphs this small)
        for scale, seed, weights in ((16, 5, None), (13, 9, "hash")):
            nv, s, d, v = gen.rmat_edges(scale, 16, seed=seed, weights=weights)
            og = ob.OracleGraph(nv, s, d, v, 2)
            opr, oit, _ = og.pagerank(6)
            opr2, oit2, _ = og.pagerank(-1)
            for tiles in (1, 4):
                g = api.Graph(nv, s, d, v, ref_threads=2, col_tiles=tiles)
                pr, deg, it = g.pagerank(6)
                assert it == oit == 6 and (f32bits(pr) == f32bits(opr)).all(), "form %d, %d tiles" % (form, tiles)
                pr, deg, it = g.pagerank(-1)
                assert it == oit2 and (f32bits(pr) == f32bits(opr2)).all(), "form %d, %d tiles, until convergence" % (form, tiles)
                g.close()
    finally:
        L.gm_set_option(b"rowwave_form", 4)


def _sweep_arrays(api, sw):
    """host copies of a gm_sweep_t's arrays"""
    nvw, T = sw.nsets * 256, sw.nslices
    out = {}
    def get(name, n, dt):
        a = np.zeros(max(int(n), 1), dt)
        ptr = getattr(sw, name)
        if ptr and n:
            api.copy_from_device(a[: int(n)], ptr)
        out[name] = a
    get("scol", sw.nentries, np.uint32); get("gbase", sw.ngroups + 1, np.uint32)
    get("wfirst", nvw * T * 17, np.uint32); get("wrow", nvw * T * 17, np.uint32); get("row_of_slot", nvw * sw.acc_rows, np.int32)
    get("lcol", sw.nedges_long, np.uint32); get("lps", nvw * T * sw.long_slots + 1, np.uint32); get("lrow_of_slot", nvw * sw.long_slots, np.int32)
    get("slice_base", T + 1, np.int32)
    get("gcol", sw.ngiant_edges, np.uint32); get("gdst", sw.ngiant_edges, np.uint32); get("gslice", T + 1, np.uint32)
    if sw.val_bytes:
        get("gval", sw.ngiant_edges, np.uint32)
    if sw.val_bytes:
        get("sval", sw.nentries, np.uint32); get("lval", sw.nedges_long, np.uint32)
        get("src_pos", sw.nentries, np.uint32); get("lsrc_pos", sw.nedges_long, np.uint32)
    return out


def _check_sweep_structure(api, g, sw, keep_values, rng, own=4096):
    """gm_sweep_t covers exactly the rows of more than 64 edges that are not giant; every piece lies inside one slice and holds
    the row's edges of that slice in CSR (= ascending native column) order; first-piece flags, wave ranges and values are right."""
    rp, ci, vv = g.csr_to_host(api.GM_DIR_OUT)
    ln = np.diff(rp)
    c = g.csr(api.GM_DIR_OUT)
    giant = np.zeros(max(c.ngiant, 1), np.int32)
    if c.ngiant:
        api.copy_from_device(giant[: c.ngiant], c.giant_row)
    A = _sweep_arrays(api, sw)
    cuts = A["slice_base"]
    T, NL, ACC = sw.nslices, sw.long_slots, sw.acc_rows
    med = A["row_of_slot"][A["row_of_slot"] >= 0]
    lng = A["lrow_of_slot"][A["lrow_of_slot"] >= 0]
    want = set(np.nonzero(ln > 64)[0].tolist()) - set(giant[: c.ngiant].tolist())
    assert len(med) + len(lng) == sw.nrows == len(want) and set(med.tolist()) | set(lng.tolist()) == want
    assert len(lng) == sw.nrows_long and (ln[lng] > own).all() and (ln[med] <= own).all()
    assert sw.nedges == int(ln[med].sum()) and sw.nedges_long == int(ln[lng].sum())
    gb, wf = A["gbase"].astype(np.int64), A["wfirst"].reshape(-1, 17)
    assert gb[0] == 0 and gb[-1] == sw.nentries and (np.diff(gb) >= 128).all() and (np.diff(gb) % 64 == 0).all()
    # (the short rows' stream groups -- gm_sweep_t.nstream, their own test below -- sit behind a block's medium groups; wfirst / wrow leave them out)
    assert (np.diff(wf, axis=1).astype(np.int64) >= 0).all() and wf[0, 0] == 0 and wf[-1, 16] <= sw.ngroups and (wf[1:, 0] >= wf[:-1, 16]).all()
    if sw.nstream == 0:
        assert wf[-1, 16] == sw.ngroups and (wf[1:, 0] == wf[:-1, 16]).all()
    assert (A["wrow"].astype(np.int64) == gb[A["wfirst"]] // 64).all()
    # every edge exactly once: entries without the pad bit (meta rows and padding carry it); checked per piece below on a sample
    assert int((A["scol"][: sw.nentries] >> 31 == 0).sum()) == sw.nedges + sw.nstream
    def row_part(row, sl):
        whole = ci[rp[row]: rp[row + 1]]
        sel = (whole >= cuts[sl]) & (whole < cuts[sl + 1])
        return whole[sel], (vv[rp[row]: rp[row + 1]][sel] if vv is not None else None), np.nonzero(sel)[0] + rp[row]
    def first_slice(row):
        return int(np.searchsorted(cuts, ci[rp[row]: rp[row + 1]].min(), side="right") - 1)
    blk_of_group = np.searchsorted(wf[:, 16], np.arange(sw.ngroups), side="right")
    for gI in (rng.choice(sw.ngroups, size=min(120, sw.ngroups), replace=False) if sw.ngroups else []):
        b = int(blk_of_group[gI]); vw, sl = b // T, b % T
        width = int(gb[gI + 1] - gb[gI]) // 64 - 1
        grp = A["scol"][gb[gI]: gb[gI + 1]].reshape(width + 1, 64)
        metas, ent = grp[0], grp[1:]
        assert (metas >> 31).all() and (((metas >> 16) & 0x1fff) == width).all()
        if (metas >> 30 & 1).any():  # a stream group: no lane has a slot, full rows of entries up to the block's last edge
            flat = ent.reshape(-1)
            nreal = int((flat >> 31 == 0).sum())
            assert (metas >> 30 & 1).all() and ((metas & 0x7fff) == 0x7fff).all() and 1 <= width <= 8 and (flat[:nreal] >> 31 == 0).all() and nreal > (width - 1) * 64
            continue
        lens = []
        for lane in range(64):
            meta = int(metas[lane]) & 0xffff
            col = ent[:, lane]
            if meta & 0x7fff == 0x7fff:
                assert (col >> 31).all()
                lens.append(0)
                continue
            row = int(A["row_of_slot"][vw * ACC + (meta & 0x7fff)])
            assert row >= 0
            n = int((col >> 31 == 0).sum())
            assert (col[:n] >> 31 == 0).all() and (col[n:] >> 31).all() and ((col[n:] & 0x7fffffff) == (int(cuts[sl]) << 2)).all()
            cols, vals, pos = row_part(row, sl)
            assert n == len(cols) > 0 and ((col[:n] >> 2) == cols).all()
            assert bool(meta & 0x8000) == (first_slice(row) == sl)
            if keep_values:
                assert (A["sval"][gb[gI]: gb[gI + 1]].reshape(width + 1, 64)[1: n + 1, lane] == vals.view(np.uint32)).all()
                assert (A["src_pos"][gb[gI]: gb[gI + 1]].reshape(width + 1, 64)[1: n + 1, lane] == pos).all()
            lens.append(n)
        assert lens[0] == width and all(x >= y for x, y in zip(lens[:-1], lens[1:]))  # longest first, padded to the longest
    # the giant rows' edges by slice: every edge once, at its place in the products stream (gterm_off + position in the row)
    if c.ngiant:
        gto = np.zeros(c.ngiant + 1, np.int64)
        api.copy_from_device(gto, c.gterm_off)
        assert sw.ngiant_edges == int(ln[giant[: c.ngiant]].sum())
        gs = A["gslice"].astype(np.int64)
        assert gs[0] == 0 and gs[-1] == sw.ngiant_edges and (np.diff(gs) >= 0).all()
        want_dst, want_col, want_val = [], [], []
        for gi in range(c.ngiant):
            row = int(giant[gi])
            want_dst.append(gto[gi] + np.arange(ln[row])); want_col.append(ci[rp[row]: rp[row + 1]])
            if keep_values:
                want_val.append(vv[rp[row]: rp[row + 1]])
        want_dst, want_col = np.concatenate(want_dst), np.concatenate(want_col)
        sl_of = np.searchsorted(cuts, want_col, side="right") - 1
        order = np.argsort(sl_of, kind="stable")  # by slice; inside a slice: row, then CSR position
        assert (A["gdst"] == want_dst[order]).all() and ((A["gcol"] >> 2) == want_col[order]).all()
        assert (np.diff(gs) == np.bincount(sl_of, minlength=T)).all()
        if keep_values:
            assert (A["gval"] == np.concatenate(want_val)[order].view(np.uint32)).all()
    lps = A["lps"].astype(np.int64)
    assert lps[0] == 0 and lps[-1] == sw.nedges_long and (np.diff(lps) >= 0).all()
    if sw.nrows_long:
        blocks = lps[:: NL]
        assert int(np.diff(blocks).max()) == sw.max_long_block
        ent = np.nonzero(np.diff(lps) > 0)[0]
        for e in rng.choice(ent, size=min(60, len(ent)), replace=False):
            b, j = int(e) // NL, int(e) % NL; vw, sl = b // T, b % T
            row = int(A["lrow_of_slot"][vw * NL + j])
            cols, vals, pos = row_part(row, sl)
            assert ((A["lcol"][lps[e]: lps[e + 1]] >> 2) == cols).all()
            if keep_values:
                assert (A["lval"][lps[e]: lps[e + 1]] == vals.view(np.uint32)).all() and (A["lsrc_pos"][lps[e]: lps[e + 1]] == pos).all()


@pytest.mark.parametrize("scale,tiles,threads", [(13, 3, 1), (15, 4, 2), (16, 8, 1)])
def test_row_stationary_sweep_bit_exact(env, scale, tiles, threads):
    """The sweep (graphmat_hip.h gm_sweep_t, kernels.hpp k_spmv_sell): its structure covers exactly the rows it takes, every piece
    lies inside one slice in ascending native column order, and PageRank through it -- with and without edge values, several
    launches (sets), the long rows staged in one or several rounds, every placement of the short-row pass -- has the bits of
    the oracle, which are also those of the tile passes (sweep_slices 0)."""
    import ctypes as C
    api, ob = env
    from graphmat_amd import _lib
    L = _lib.lib()
    nv, s, d, v = gen.rmat_edges(scale, 16, 5, weights="hash")
    og = ob.OracleGraph(nv, s, d, None, ref_threads=threads)
    odeg = og.degree()
    opr, _, _ = og.pagerank(6, degree=odeg)
    rng = np.random.default_rng(1)
    seen_sets = [1]
    #        sweep_slices, sweep_form, keep_values, own_wave_row, acc_rows, long_slots
    cases = [(0, 0, False, 4096, 10048, 512), (1, 0, False, 0, 10048, 512), (1, 1, True, 4096, 10048, 512), (1, 2, False, 0, 10048, 512), (1, 8, False, 0, 10048, 512), (1, 9, True, 4096, 10048, 512),
             (1, 0, True, 65, 10048, 512), (1, 0, False, 65, 10048, 512),  # (next to no medium rows: everything staged)
             (16, 0, False, 256, 10048, 512), (16, 4, True, 256, 10048, 512), (24, 5, False, 128, 3, 3), (128, 0, True, 512, 2, 512)]
    try:
        for slices, form, keep, own, accl, longl in cases:
            api._lib.check(L.gm_reset_options())
            for k_, v_ in ((b"sweep_slices", slices), (b"sweep_form", form), (b"sweep_long_row", own), (b"sweep_acc_rows", accl), (b"sweep_long_slots", longl)):
                api._lib.check(L.gm_set_option(k_, v_))
            g = api.Graph(nv, s, d, v if keep else None, ref_threads=threads, keep_values=keep, col_tiles=tiles)
            assert g.col_tiles > 1
            sw = _lib.Sweep()
            assert L.gm_graph_sweep(g.h, C.byref(sw)) == 0
            if slices == 0:
                assert sw.nrows == 0 and sw.nslices == 0
            else:
                assert sw.nrows > 0 and sw.nslices % g.col_tiles == 0 and sw.val_bytes == (4 if keep else 0)
                nmed, nlng = sw.nrows - sw.nrows_long, sw.nrows_long
                assert sw.nsets == max(1, -(-(-(-nmed // 256)) // accl), -(-(-(-nlng // 256)) // longl))
                if 0 < own < 4096 and scale >= 15:
                    assert sw.nrows_long > 0
                seen_sets[0] = max(seen_sets[0], sw.nsets)
                _check_sweep_structure(api, g, sw, keep, rng, own if own else sw.long_row)
            pr, deg, it = g.pagerank(6)
            assert (deg == odeg).all() and it == 6
            assert (f32bits(pr) == f32bits(opr)).all(), "sweep_slices %d sweep_form %d values %d own_wave_row %d" % (slices, form, keep, own)
            g.close()
        if scale >= 15:
            assert seen_sets[0] > 1  # (several launches were exercised)
    finally:
        api._lib.check(L.gm_reset_options())


@pytest.mark.parametrize("keep", [False, True])
def test_sweep_in_768_thread_workgroups_bit_exact(env, keep):
    """gm_set_option("sweep_waves", 12): the blocks' groups dealt over 12 waves and the sweep run by k_spmv_sell_w12 (768-thread workgroups, a smaller
    LDS pool) with the giant rows' gathers (k_giant_gather_sliced), their fold passes and the short rows' kernel BESIDE it on the auxiliary stream -- an
    experiment of round 6 that does not pay (RMAT-26 3.82 against 3.77 ms: the two kernels share the CUs' vector-memory path, their times add) and stays
    an option.  Same bits as the oracle."""
    import ctypes as C
    api, ob = env
    from graphmat_amd import _lib
    L = _lib.lib()
    nv, s, d, v = gen.rmat_edges(16, 16, 5, weights="hash")
    opr, _, _ = ob.OracleGraph(nv, s, d, None, ref_threads=1).pagerank(6)
    try:
        api._lib.check(L.gm_reset_options())
        api._lib.check(L.gm_set_option(b"sweep_waves", 12))
        api._lib.check(L.gm_set_option(b"sweep_long_row", 256))
        g = api.Graph(nv, s, d, v if keep else None, ref_threads=1, keep_values=keep, col_tiles=4)
        sw = _lib.Sweep()
        assert L.gm_graph_sweep(g.h, C.byref(sw)) == 0 and sw.nrows > 0 and sw.waves == 12 and sw.nrows_long > 0
        pr, deg, it = g.pagerank(6)
        assert it == 6 and (f32bits(pr) == f32bits(opr)).all()
        g.close()
    finally:
        api._lib.check(L.gm_reset_options())


@pytest.mark.parametrize("scale,tiles,threads,keep,acc_rows", [(13, 2, 1, False, 0), (15, 3, 2, True, 0), (16, 4, 1, False, 0), (16, 3, 2, True, 0), (16, 3, 1, False, 2)])
def test_short_rows_ride_the_sweep(env, scale, tiles, threads, keep, acc_rows):
    """The rows of 1 .. 64 edges as STREAM groups of the sweep (graphmat_hip.h: gm_sweep_t.nstream; kernels.hpp: k_spmv_sell_stream + k_short_fold):
    the structure lists every short row once, in device order, its bins cover the short rows' edges exactly, every (bin, slice) chunk lies inside
    the products stream, the stream groups are the only difference between wrow and wrow_stream, and sinv is a permutation inside every bin; PageRank through it -- with and without edge values -- has the oracle's bits,
    the path was really taken (note 4 of the graph), and the same graph with the row-block kernel kept (sweep_form bit 7) or without the structure
    (gm_set_option("sweep_stream", 0)) gives the same bits."""
    import ctypes as C
    api, ob = env
    from graphmat_amd import _lib
    L = _lib.lib()
    nv, s, d, v = gen.rmat_edges(scale, 16, 11, weights="hash")
    og = ob.OracleGraph(nv, s, d, v if keep else None, ref_threads=threads)
    opr, oit, _ = og.pagerank(7)
    try:
        api._lib.check(L.gm_reset_options())
        api._lib.check(L.gm_set_option(b"sweep_long_row", 256))
        api._lib.check(L.gm_set_option(b"sweep_form", 256))  # (the path is taken from 2^26 short-row edges on; bit 8 lifts the limit)
        if acc_rows:  # several launches: the stream groups all sit in the first one's blocks
            api._lib.check(L.gm_set_option(b"sweep_acc_rows", acc_rows))
            api._lib.check(L.gm_set_option(b"sweep_long_slots", 1))
        g = api.Graph(nv, s, d, v if keep else None, ref_threads=threads, keep_values=keep, col_tiles=tiles)
        sw = _lib.Sweep()
        assert L.gm_graph_sweep(g.h, C.byref(sw)) == 0 and sw.nrows > 0 and (sw.nsets > 1) == bool(acc_rows)
        assert sw.nstream > 0 and sw.nbins > 0 and sw.bin_cap == 12288 and sw.nstream_slots % 64 == 0 and sw.nstream_slots >= sw.nstream and sw.wrow_stream
        c = g.csr(api.GM_DIR_OUT)
        rp = np.zeros(c.nrows + 1, np.int64)
        api.copy_from_device(rp, c.rowptr)
        lens = np.diff(rp)
        short = np.nonzero((lens >= 1) & (lens <= sw.short_row))[0]
        srow = np.zeros(sw.nshort_rows, np.int32)
        api.copy_from_device(srow, sw.srow)
        soff = np.zeros(sw.nshort_rows + 1, np.uint32)
        api.copy_from_device(soff, sw.soff)
        assert (srow == short).all() and int(soff[-1]) == sw.nstream == int(lens[short].sum()) and (np.diff(soff.astype(np.int64)) == lens[short]).all()
        binrow = np.zeros(sw.nbins + 1, np.uint32)
        api.copy_from_device(binrow, sw.sbin_row)
        assert binrow[0] == 0 and binrow[-1] == sw.nshort_rows and (np.diff(binrow.astype(np.int64)) >= 0).all()
        chunk = np.zeros(sw.nbins * sw.nslices * 2, np.uint32)
        api.copy_from_device(chunk, sw.schunk)
        chunk = chunk.reshape(sw.nbins, sw.nslices, 2).astype(np.int64)
        assert int(chunk[:, :, 1].sum()) == sw.nstream and ((chunk[:, :, 0] + chunk[:, :, 1]) <= sw.nstream_slots).all()
        # the stream groups sit behind the medium groups of the first launch's blocks: wrow (every other kernel form) ends a block in front of
        # them, wrow_stream behind them, and exactly the stream rows lie between
        wr = np.zeros(sw.nsets * 256 * sw.nslices * 17, np.uint32)
        ws = np.zeros_like(wr)
        api.copy_from_device(wr, sw.wrow)
        api.copy_from_device(ws, sw.wrow_stream)
        wr, ws = wr.reshape(-1, 17).astype(np.int64), ws.reshape(-1, 17).astype(np.int64)
        assert (wr[:, 0] == ws[:, 0]).all() and (ws[:, 16] >= wr[:, 16]).all()
        extra_rows = int((ws[:, 16] - wr[:, 16]).sum())  # meta rows + product rows
        assert extra_rows >= sw.nstream_slots // 64 and (ws[256 * sw.nslices:, 16] == wr[256 * sw.nslices:, 16]).all()
        sinv = np.zeros(sw.nstream_slots, np.uint16)
        api.copy_from_device(sinv, sw.sinv)
        for b in range(sw.nbins):
            i0, i1 = int(binrow[b]), int(binrow[b + 1])
            want = int(soff[i1]) - int(soff[i0])
            assert int(chunk[b, :, 1].sum()) == want and int(soff[i0]) >= b * sw.bin_cap and (i1 == i0 or int(soff[i1 - 1]) < (b + 1) * sw.bin_cap)
            pos = np.concatenate([np.arange(p0, p0 + n) for p0, n in chunk[b]]) if want else np.zeros(0, np.int64)
            got = np.sort(sinv[pos].astype(np.int64))
            assert (got == np.arange(int(soff[i0]) - b * sw.bin_cap, int(soff[i0]) - b * sw.bin_cap + want)).all(), "sinv is no permutation inside bin %d" % b
        n4 = C.c_int64(-1)
        pr, _, it = g.pagerank(7)
        assert it == oit == 7 and (f32bits(pr) == f32bits(opr)).all()
        assert L.gm_graph_note_get(g.h, 4, C.byref(n4)) == 0 and n4.value == 7
        api._lib.check(L.gm_set_option(b"sweep_form", 128))  # the row-block kernel for the short rows
        pr2, _, _ = g.pagerank(7)
        assert (f32bits(pr2) == f32bits(opr)).all() and L.gm_graph_note_get(g.h, 4, C.byref(n4)) == 0 and n4.value == 0
        api._lib.check(L.gm_set_option(b"sweep_form", 0))  # (a small graph without bit 8: the row-block kernel too)
        pr2, _, _ = g.pagerank(7)
        assert (f32bits(pr2) == f32bits(opr)).all() and L.gm_graph_note_get(g.h, 4, C.byref(n4)) == 0 and n4.value == 0
        api._lib.check(L.gm_set_option(b"sweep_form", 256))
        pru, itu, _ = og.pagerank(-1)
        pr3, _, it3 = g.pagerank(-1)  # (until convergence: ACTIVE_ONLY-free program, but the iteration count must agree too)
        assert it3 == itu and (f32bits(pr3) == f32bits(pru)).all()
        g.close()
        api._lib.check(L.gm_set_option(b"sweep_stream", 0))
        g = api.Graph(nv, s, d, v if keep else None, ref_threads=threads, keep_values=keep, col_tiles=tiles)
        assert L.gm_graph_sweep(g.h, C.byref(sw)) == 0 and sw.nrows > 0 and sw.nstream == 0 and not sw.sinv
        pr4, _, _ = g.pagerank(7)
        assert (f32bits(pr4) == f32bits(opr)).all()
        g.close()
    finally:
        api._lib.check(L.gm_reset_options())


@pytest.mark.parametrize("scale,tiles,threads", [(13, 3, 1), (15, 3, 2), (16, 4, 1)])
def test_sparse_message_vector_through_the_sweep(env, scale, tiles, threads):
    """ACTIVE_ONLY programs on a swept graph (round 6; kernels.hpp: k_spmv_sell_sparse): the pull steps of SSSP (uint32 messages, int edge values
    carried by the structure, min) go through the sweep with a presence test per entry, the first PRESENT message of a row assigning, and y's
    presence bits produced by the kernel.  Distances and iteration counts must be the oracle's -- from several sources, so that the active sets
    range from a handful of vertices to most of the graph -- and the same with the sweep refused for sparse vectors (sweep_form bit 5); the
    path must really have been taken."""
    import ctypes as C
    api, ob = env
    from graphmat_amd import _lib
    L = _lib.lib()
    nv, s, d, v = gen.rmat_edges(scale, 16, 7, weights="hash")
    og = ob.OracleGraph(nv, s, d, v, ref_threads=threads)
    try:
        api._lib.check(L.gm_reset_options())
        api._lib.check(L.gm_set_option(b"sweep_long_row", 256))
        # (top-down steps off: every iteration is a pull step, whatever the size of its active set)
        api._lib.check(L.gm_set_option(b"debug_flags", 32))
        api._lib.check(L.gm_set_option(b"sweep_form", 64))  # (the form is taken from 2^27 edges on; bit 6 lifts the limit)
        g = api.Graph(nv, s, d, v, ref_threads=threads, col_tiles=tiles)
        sw = _lib.Sweep()
        assert L.gm_graph_sweep(g.h, C.byref(sw)) == 0 and sw.nrows > 0 and sw.val_bytes == 4
        taken = 0
        for src in (1, 2, 77):
            od, oit = og.sssp(src)
            dist, it = g.sssp(src)
            n3 = C.c_int64(0)
            assert L.gm_graph_note_get(g.h, 3, C.byref(n3)) == 0
            taken += n3.value
            assert it == oit and (dist == od).all(), "source %d" % src
            api._lib.check(L.gm_set_option(b"sweep_form", 32))
            dist2, it2 = g.sssp(src)
            api._lib.check(L.gm_set_option(b"sweep_form", 64))
            assert L.gm_graph_note_get(g.h, 3, C.byref(n3)) == 0 and n3.value == 0
            assert it2 == oit and (dist2 == od).all()
        assert taken >= 3
        g.close()
    finally:
        api._lib.check(L.gm_reset_options())


def _blocked_info(g):
    import ctypes as C
    from graphmat_amd import _lib
    bl = _lib.Blocked()
    assert g.L.gm_graph_blocked(g.h, C.byref(bl)) == 0
    return bl


@pytest.mark.parametrize("kind,scale,tiles,threads", [("uniform", 14, 3, 1), ("uniform", 15, 2, 1), ("uniform", 16, 4, 2), ("rmat", 15, 3, 1), ("rmat", 16, 4, 3), ("rmat+sweep", 15, 3, 2), ("rmat+sweep", 16, 4, 1)])
def test_blocked_short_rows_bit_exact(env, kind, scale, tiles, threads):
    """The column-blocked stream of the short rows (graphmat_hip.h gm_blocked_t, kernels.hpp k_spmv_blocked): forced on small graphs --
    a graph without skew (every row short: the whole multiply goes through it), RMAT with the sweep switched off (the rows above
    the short-row limit then take the wave / giant kernels in front of it) and RMAT with the sweep (medium rows swept, giant rows' passes, then the stream) --, its structure covers exactly the rows of 1 .. 64 edges, and
    PageRank through it has the oracle's bits, which are also those of the row-blocks (blocked_rows -1)."""
    api, ob = env
    from graphmat_amd import _lib
    L = _lib.lib()
    if kind == "uniform":
        nv, s, d, v = gen.uniform_out_regular_edges(1 << scale, 16, seed=3)
    else:
        nv, s, d, v = gen.rmat_edges(scale, 16, 7, weights="hash")
    og = ob.OracleGraph(nv, s, d, None, ref_threads=threads)
    odeg = og.degree()
    opr, _, _ = og.pagerank(5, degree=odeg)
    try:
        for mode in (1, -1):
            api._lib.check(L.gm_reset_options())
            api._lib.check(L.gm_set_option(b"blocked_rows", mode))
            if kind == "rmat":  # (debug_flags 16 = no auxiliary stream: the sweep, which needs one, is not taken)
                api._lib.check(L.gm_set_option(b"debug_flags", 16))
            keep = (scale % 2 == 1)  # (the odd scales keep 4-byte edge values: the entries then carry them)
            g = api.Graph(nv, s, d, v if keep else None, ref_threads=threads, keep_values=keep, col_tiles=tiles)
            assert g.col_tiles > 1
            bl = _blocked_info(g)
            rp, ci, vv = g.csr_to_host(api.GM_DIR_OUT)
            rowlen = np.diff(rp)
            short = np.nonzero((rowlen >= 1) & (rowlen <= 64))[0]
            if mode == 1:
                assert bl.nrows == len(short) and bl.nentries == int(rowlen[short].sum()) and bl.nblocks == -(-len(short) // 32768) and bl.nslices >= 2
                row_of = np.zeros(bl.nblocks * 32768, np.int32)
                api.copy_from_device(row_of, bl.row_of)
                # the short rows are dealt over the blocks in runs of 64 (run j -> block j % nblocks)
                i_ = np.arange(bl.nrows); j_ = i_ >> 6
                slot_of = (j_ % bl.nblocks) * 32768 + (j_ // bl.nblocks) * 64 + (i_ & 63)
                assert (row_of[slot_of] == short).all() and int((row_of >= 0).sum()) == bl.nrows
                ecol = np.zeros(bl.nentries, np.uint32); erow = np.zeros(bl.nentries, np.uint16)
                api.copy_from_device(ecol, bl.ecol); api.copy_from_device(erow, bl.erow)
                assert int((erow >> 15).sum()) == bl.nrows  # one "first edge" per row
                assert bl.val_bytes == (4 if keep else 0)
                if keep:
                    ev = np.zeros(bl.nentries, np.int32); ep = np.zeros(bl.nentries, np.uint32)
                    api.copy_from_device(ev, bl.eval); api.copy_from_device(ep, bl.epos)
                    assert (ev == vv[ep]).all() and (ecol == ci[ep].astype(np.uint32)).all()
                woff = np.zeros((bl.nblocks * bl.nslices + 1) * 17, np.uint32)
                api.copy_from_device(woff, bl.woff)
                w = woff[: bl.nblocks * bl.nslices * 17].reshape(-1, 17)
                assert (np.diff(w.astype(np.int64), axis=1) >= 0).all() and w[0, 0] == 0 and w[-1, 16] == bl.nentries and (w[1:, 0] == w[:-1, 16]).all()
                # a row's entries, in stream order, are its CSR columns in CSR order
                for r in np.random.default_rng(2).choice(bl.nrows, size=40, replace=False):
                    b, k = int(slot_of[r]) // 32768, int(slot_of[r]) % 32768
                    seg = slice(int(w[b * bl.nslices, 0]), int(w[(b + 1) * bl.nslices - 1, 16]))
                    mine = np.nonzero((erow[seg] & 0x7fff) == k)[0]
                    row = int(short[r])
                    assert (ecol[seg][mine] == ci[rp[row]: rp[row + 1]].astype(np.uint32)).all()
                    assert (erow[seg][mine][0] >> 15) == 1 and ((erow[seg][mine][1:] >> 15) == 0).all()
            else:
                assert bl.nrows == 0
            pr, deg, it = g.pagerank(5)
            assert (deg == odeg).all() and it == 5
            assert (f32bits(pr) == f32bits(opr)).all(), "%s scale %d blocked_rows %d" % (kind, scale, mode)
            g.close()
    finally:
        api._lib.check(L.gm_reset_options())


def test_blocked_short_rows_several_passes_equal_the_row_blocks(env):
    """More short rows than one pass of 256 workgroups x 32768 holds (uniform 2^24: 512 blocks, two passes): the bits of the
    row-block kernel (blocked_rows -1), which the small cases above tie to the oracle."""
    import torch
    api, _ = env
    from graphmat_amd import _lib
    L = _lib.lib()
    nv, src, dst, _ = api.uniform_on_device(24, 16, 1)
    res = {}
    try:
        for mode in (1, -1):
            api._lib.check(L.gm_reset_options())
            api._lib.check(L.gm_set_option(b"blocked_rows", mode))
            g = api.Graph(nv, src, dst, None, keep_values=False)
            bl = _blocked_info(g)
            assert (bl.nrows > 256 * 32768 and bl.nblocks > 256) if mode == 1 else bl.nrows == 0
            pr, deg, it = g.pagerank(4)
            res[mode] = pr.copy()
            g.close()
    finally:
        api._lib.check(L.gm_reset_options())
    assert (f32bits(res[1]) == f32bits(res[-1])).all()
