"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads
without a GPU and exports every symbol include/graphmat_hip.h declares.  No compute calls."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from graphmat_amd import build
    build.build()
    from graphmat_amd import _lib
    return _lib.lib()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "graphmat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", text))
    names.discard("gm_exchange_fn")
    return sorted(names)


def test_header_and_binding_agree(lib):
    from graphmat_amd import _lib
    decl = declared_functions()
    assert len(decl) >= 30
    assert sorted(_lib.SIGNATURES) == decl


def test_every_declared_symbol_is_exported(lib):
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_host_only_entry_points(lib, golden_dir):
    import ctypes as C
    # id permutation: include/Graph.h:111-150 (V=1024 at 1 thread: P=16, h=64)
    assert lib.gm_vertex_to_native(1, 16, 1024) == 1
    assert lib.gm_vertex_to_native(2, 16, 1024) == 65
    assert lib.gm_vertex_to_native(17, 16, 1024) == 2
    for v in (1, 2, 17, 500, 1024):
        assert lib.gm_native_to_vertex(lib.gm_vertex_to_native(v, 16, 1024), 16, 1024) == v
    assert lib.gm_vertex_to_native(7, 16, 8) == 7  # identity when V < P
    # .mtx reader honours the header count (trailing duplicate record ignored)
    nv, nnz = C.c_int(), C.c_int64()
    s, d, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = lib.gm_mtx_read(os.path.join(golden_dir, "test.bin.mtx").encode(), 4, C.byref(nv), C.byref(nnz),
                         C.byref(s), C.byref(d), C.byref(v))
    assert rc == 0 and nv.value == 8 and nnz.value == 13
    src = (C.c_int32 * 13).from_address(s.value)
    assert list(src)[:3] == [1, 1, 2]
    for p in (s, d, v):
        lib.gm_host_free(p)
    rc = lib.gm_mtx_read(b"/nonexistent/file.mtx", 4, C.byref(nv), C.byref(nnz), C.byref(s), C.byref(d), C.byref(v))
    assert rc != 0 and b"Could not open" in lib.gm_last_error()


def test_package_has_no_oracle_dependency():
    """The product path must never import or link the oracle."""
    pkg = os.path.join(ROOT, "graphmat_amd")
    inc = os.path.join(ROOT, "include")
    for base in (pkg, inc):
        for dirpath, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "gm_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_error_convention_invalid_arguments(lib):
    """C-ABI error convention: status code + gm_last_error text, nothing thrown, no GPU touched
    for argument errors (these run on the CPU-only box)."""
    import ctypes as C
    from graphmat_amd import _lib
    h = C.c_void_p()
    # nvertices <= 0
    d = _lib.GraphDesc(0, 16, 0, 0, 3, 0, 0, 0, 0, 1, 0, 0, 0)
    assert lib.gm_graph_create(C.byref(h), C.byref(d), 0, None, None, None, None) == 1
    assert b"invalid descriptor" in lib.gm_last_error()
    # unaligned shard boundary in the native layout
    d = _lib.GraphDesc(1000, 16, 10, 1000, 3, 0, 0, 0, 0, 1, 0, 0, 0)
    assert lib.gm_graph_create(C.byref(h), C.byref(d), 0, None, None, None, None) == 1
    assert b"multiples of 64" in lib.gm_last_error()
    # unknown layout / bad shard
    d = _lib.GraphDesc(1000, 16, 0, 1000, 3, 0, 0, 0, 7, 1, 0, 0, 0)
    assert lib.gm_graph_create(C.byref(h), C.byref(d), 0, None, None, None, None) == 1
    d = _lib.GraphDesc(1000, 16, 0, 1000, 3, 0, 0, 0, 1, 4, 4, 0, 0)
    assert lib.gm_graph_create(C.byref(h), C.byref(d), 0, None, None, None, None) == 1
    assert b"invalid shard" in lib.gm_last_error()
    # distributed build (edges_local) without the library's communicator, and with the native layout
    d = _lib.GraphDesc(1000, 16, 0, 1000, 3, 0, 0, 0, 1, 2, 0, 0, 0, 0, 1)
    assert lib.gm_graph_create(C.byref(h), C.byref(d), 0, None, None, None, None) == 1
    assert b"collective over the gm_dist communicator" in lib.gm_last_error()
    d = _lib.GraphDesc(1024, 16, 0, 1024, 3, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1)
    assert lib.gm_graph_create(C.byref(h), C.byref(d), 0, None, None, None, None) == 1
    assert b"GM_LAYOUT_DEGREE" in lib.gm_last_error()
    # null handles
    assert lib.gm_graph_desc(None, C.byref(d)) == 1
    assert lib.gm_run_pagerank(None, None, 0.3, 1, None, None) == 1
    assert lib.gm_run_sgd(None, None, 20, 8, 0.0, 0.0, 1, None, None) != 0
    assert lib.gm_set_option(b"no_such_option", 1) == 1
    assert lib.gm_graph_destroy(None) == 0


def test_error_convention_of_the_service_entry_points(lib, tmp_path):
    """The entry points the header layer and the multi-GPU glue use reject bad arguments with a status
    code (no GPU needed), and the edge-list reader reports unreadable / inconsistent files."""
    import ctypes as C
    i32, i64, vp, sz = C.c_int32(), C.c_int64(), C.c_void_p(), C.c_size_t()
    assert lib.gm_graph_split(None, 1, 650, C.byref(i32), C.byref(i32), C.byref(i32)) == 1
    assert lib.gm_graph_note_set(None, 0, 1) == 1 and lib.gm_graph_note_get(None, 0, C.byref(i64)) == 1
    assert lib.gm_graph_workspace_info(None, 0, C.byref(vp), C.byref(sz), C.byref(C.c_int())) == 1
    assert lib.gm_graph_run_resources(None, C.byref(vp), C.byref(vp), C.byref(vp), C.byref(vp)) == 1
    assert lib.gm_graph_workspace(None, 0, 16, C.byref(vp)) == 1
    assert lib.gm_graph_adopt_workspace(None, 1, None, 0) == 1
    m, n = C.c_int(), C.c_int()
    ps, pd, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    args = (C.byref(m), C.byref(n), C.byref(i64), C.byref(ps), C.byref(pd), C.byref(pv))
    assert lib.gm_edgelist_read(str(tmp_path / "nope").encode(), 1, 1, 1, 1, *args) != 0
    assert b"Could not open" in lib.gm_last_error()
    assert lib.gm_edgelist_read(b"x", 1, 1, 1, 99, *args) == 1           # unknown value kind
    assert lib.gm_edgelist_read(b"x", 0, 1, 1, 0x100 + 12, *args) == 1   # opaque values need a binary file
    short = tmp_path / "short.bin"
    short.write_bytes(b"\x05\x00\x00\x00\x05\x00\x00\x00\x03\x00\x00\x00" + b"\x01\x00\x00\x00\x02\x00\x00\x00\x07\x00\x00\x00")
    assert lib.gm_edgelist_read(str(short).encode(), 1, 1, 1, 1, *args) != 0
    assert b"header says 3 edges" in lib.gm_last_error()
    assert lib.gm_edgelist_write(None, 1, 1, 1, 1, 3, 3, 0, None, None, None) == 1
    assert lib.gm_set_option(b"push_edge_permille", 2000) == 1 and lib.gm_set_option(b"push_edge_permille", 50) == 0


def test_product_has_no_test_transport_and_the_stand_in_exports_what_gm_dist_binds():
    """The shared-memory stand-in for librccl lives with the tests (tests/support/), is bound through the same dlopen
    hook as librccl (GRAPHMAT_RCCL_LIBRARY) and exports every entry point gm_dist.hip binds; nothing of it is
    compiled into the product."""
    import ctypes as C
    src = open(os.path.join(ROOT, "graphmat_amd", "csrc", "gm_dist.hip")).read()
    assert "shm_open" not in src and "GRAPHMAT_DIST_TRANSPORT" not in src
    bound = re.findall(r'GM_BIND\(\w+, "(nccl\w+)"\)', src)
    assert len(bound) >= 10
    from tests.support import build as shm_build
    so = C.CDLL(shm_build.build())
    for name in bound:
        assert hasattr(so, name), name


def test_launch_detection_from_the_environment(lib):
    """gm_dist_init_from_env joins a communicator only on an unambiguous multi-rank launch: a lone WORLD_SIZE, a size
    without its rank, or SLURM_NTASKS of an allocation (sbatch without srun) must leave a single process alone --
    these calls return at once instead of waiting for ranks that do not exist (no GPU is touched)."""
    import ctypes as C
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from graphmat_amd import _lib; L = _lib.lib(); "
            "r = C.c_int(-1); n = C.c_int(-1); rc = L.gm_dist_init_from_env(C.byref(r), C.byref(n)); print('RESULT', rc, r.value, n.value)" % ROOT)
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SLURM_NTASKS", "SLURM_PROCID",
                                                            "GRAPHMAT_NRANKS", "GRAPHMAT_RANK", "PMI_SIZE", "PMI_RANK")}
    for extra in ({}, {"WORLD_SIZE": "8"}, {"SLURM_NTASKS": "8", "SLURM_PROCID": "0"}, {"PMI_SIZE": "4"},
                  {"GRAPHMAT_NRANKS": "1", "WORLD_SIZE": "8", "RANK": "3"}):
        out = subprocess.run([sys.executable, "-c", code], env=dict(base, **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
        assert b"RESULT 0 0 1" in out.stdout, (extra, out.stdout[-500:])
    # an invalid rank of a real launch is an error, not a hang
    out = subprocess.run([sys.executable, "-c", code], env=dict(base, GRAPHMAT_NRANKS="2", GRAPHMAT_RANK="5"), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=120)
    assert b"RESULT 1 " in out.stdout, out.stdout[-500:]


def test_engine_options_defaults_overrides_and_reset(lib):
    """gm_engine_options_t: process defaults (gm_set_option), reset (gm_reset_options); no graph and no GPU involved."""
    import ctypes as C
    from graphmat_amd import _lib
    o = _lib.EngineOptions()
    assert lib.gm_graph_engine_options(None, C.byref(o)) == 0
    assert (o.debug_flags, o.wave16_form, o.rowwave_form, o.giant_maps, o.ordered_giant_two_pass, o.fuse_apply_send) == (0, 2, 4, 1, 2, 1)
    assert (o.untiled_pass_plain, o.last_rows_lanes, o.push_edge_permille, o.bits_step_edges, o.sparse_step_edges) == (1, 8, 50, 2 << 20, 1 << 20)
    assert lib.gm_set_option(b"debug_flags", 128) == 0 and lib.gm_set_option(b"wave16_form", 16 + 2) == 0
    assert lib.gm_set_option(b"wave16_form", 7) != 0 and lib.gm_set_option(b"last_rows_lanes", 12) != 0  # out of range: refused
    assert lib.gm_graph_engine_options(None, C.byref(o)) == 0 and (o.debug_flags, o.wave16_form) == (128, 18)
    assert lib.gm_graph_set_option(None, b"debug_flags", 1) != 0  # needs a graph
    assert lib.gm_reset_options() == 0
    assert lib.gm_graph_engine_options(None, C.byref(o)) == 0 and (o.debug_flags, o.wave16_form) == (0, 2)
    assert lib.gm_graph_engine_options(None, None) != 0


def test_engine_options_from_the_environment(lib):
    """GRAPHMAT_OPTIONS="key=value,...": defaults for applications that cannot call gm_set_option (the reference's unchanged
    sources); read when the library first needs the options, bad entries reported and ignored.  (A fresh process: the
    library of this one has long read its environment.)"""
    import subprocess
    import sys
    code = ("import ctypes as C; from graphmat_amd import _lib; L = _lib.lib(); o = _lib.EngineOptions(); "
            "assert L.gm_graph_engine_options(None, C.byref(o)) == 0; print(o.debug_flags, o.wave16_form, o.two_stage_head_permille, o.fuse_apply_send)")
    env = dict(os.environ, GRAPHMAT_OPTIONS="debug_flags=64,fuse_apply_send=0,two_stage_head_permille=800,wave16_form=99,nonsense=3")
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    assert out.stdout.decode().split()[-4:] == ["64", "2", "800", "0"]
    err = out.stderr.decode()
    assert "ignoring 'wave16_form=99'" in err and "ignoring 'nonsense=3'" in err


def test_product_library_has_no_ablation_or_fault_injection_switches(lib):
    """Result-invalidating switches live in separate builds (build/ablation, build/hooks) that tools and the negative
    control load explicitly: the shipped library rejects their keys and does not know the fault-injection variable."""
    assert lib.gm_set_option(b"ablate_cold_from", 1) != 0
    assert lib.gm_set_option(b"ablate_cold_short", 1) != 0
    for bit in (1, 2, 4, 8, 16384):
        assert lib.gm_set_option(b"debug_flags", bit) != 0
    assert lib.gm_set_option(b"debug_flags", 16) == 0 and lib.gm_set_option(b"debug_flags", 0) == 0  # (strategy choices stay)
    assert lib.gm_graph_set_option(None, b"ablate_cold_from", 1) != 0
    from graphmat_amd import _lib
    blob = open(_lib.SO, "rb").read()
    assert b"GRAPHMAT_DEBUG_DROP_WAIT" not in blob
    lib.gm_reset_options()
