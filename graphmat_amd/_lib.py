"""ctypes loader of libgraphmat_hip.so.  Fails loudly when the HIP library is missing:
there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libgraphmat_hip.so")

GM_DIR_OUT = 1
GM_DIR_IN = 2
GM_XCHG_MESSAGES = 0
GM_XCHG_CONVERGED = 1
GM_XCHG_PART = 2
GM_XCHG_WAIT = 3
GM_XCHG_STATE = 4
GM_XCHG_GATHER = 5
GM_XCAP_SPARSE = 1
GM_WS_GATHER = 12
GM_LAYOUT_NATIVE = 0
GM_LAYOUT_DEGREE = 1


class GraphDesc(C.Structure):
    _fields_ = [("nvertices", C.c_int32), ("nparts", C.c_int32), ("row_lo", C.c_int32), ("row_hi", C.c_int32),
                ("directions", C.c_int32), ("val_bytes", C.c_int32), ("ids_on_device", C.c_int32),
                ("ids_are_native", C.c_int32), ("layout", C.c_int32), ("nshards", C.c_int32), ("shard", C.c_int32),
                ("ndevice", C.c_int32), ("xchg_rows", C.c_int32), ("col_tiles", C.c_int32), ("edges_local", C.c_int32)]


class Csr(C.Structure):
    _fields_ = [("nnz", C.c_int64), ("nrows", C.c_int32), ("row_base", C.c_int32), ("ncols", C.c_int32),
                ("val_bytes", C.c_int32), ("rowptr", C.c_void_p), ("colidx", C.c_void_p), ("vals", C.c_void_p), ("rowbits", C.c_void_p),
                ("seg_row", C.c_void_p), ("nseg", C.c_int32), ("blk_seg", C.c_void_p), ("nblk", C.c_int32),
                ("mid_row", C.c_void_p), ("nmid", C.c_int32), ("giant_row", C.c_void_p), ("ngiant", C.c_int32),
                ("gchunk_row", C.c_void_p), ("gchunk_edge", C.c_void_p), ("gterm_off", C.c_void_p),
                ("ngchunk", C.c_int32), ("giant_edges", C.c_int64), ("short_row", C.c_int32), ("nmid_long", C.c_int32),
                ("umid_row", C.c_void_p), ("numid", C.c_int32), ("numid_long", C.c_int32), ("tile_min_row", C.c_int32),
                ("hot_base", C.c_int32), ("hot_len", C.c_int32), ("hot_slices", C.c_int32), ("hot_stride", C.c_int32),
                ("gchunk_state", C.c_void_p), ("edges_blk", C.c_int64), ("edges_wave16", C.c_int64), ("edges_wave", C.c_int64), ("rows_keep_stream", C.c_int32), ("cold_from", C.c_int32)]


class Sweep(C.Structure):
    _fields_ = [("nrows", C.c_int32), ("nrows_long", C.c_int32), ("nsets", C.c_int32), ("nslices", C.c_int32), ("acc_rows", C.c_int32),
                ("long_slots", C.c_int32), ("max_long_block", C.c_int32), ("val_bytes", C.c_int32), ("short_row", C.c_int32), ("long_row", C.c_int32),
                ("nedges", C.c_int64), ("nedges_long", C.c_int64), ("nentries", C.c_int64), ("ngroups", C.c_int64),
                ("scol", C.c_void_p), ("sval", C.c_void_p), ("gbase", C.c_void_p), ("wrow", C.c_void_p), ("wfirst", C.c_void_p),
                ("row_of_slot", C.c_void_p), ("lcol", C.c_void_p), ("lval", C.c_void_p), ("lps", C.c_void_p), ("lrow_of_slot", C.c_void_p),
                ("slice_base", C.c_void_p), ("src_pos", C.c_void_p), ("lsrc_pos", C.c_void_p),
                ("ngiant_edges", C.c_int64), ("gcol", C.c_void_p), ("gval", C.c_void_p), ("gdst", C.c_void_p), ("gslice", C.c_void_p), ("gsrc_pos", C.c_void_p),
                ("nsub", C.c_int32), ("stride", C.c_int32), ("hot_words", C.c_int32), ("waves", C.c_int32),
                ("nstream", C.c_int64), ("nstream_slots", C.c_int64), ("nshort_rows", C.c_int32), ("nbins", C.c_int32), ("bin_cap", C.c_int32),
                ("stream_width", C.c_int32), ("srow", C.c_void_p), ("soff", C.c_void_p), ("sbin_row", C.c_void_p), ("schunk", C.c_void_p), ("sinv", C.c_void_p),
                ("wrow_stream", C.c_void_p)]


class Blocked(C.Structure):
    _fields_ = [("nrows", C.c_int32), ("nblocks", C.c_int32), ("nslices", C.c_int32), ("short_row", C.c_int32), ("nsteps", C.c_int32), ("val_bytes", C.c_int32),
                ("nentries", C.c_int64), ("ecol", C.c_void_p), ("erow", C.c_void_p), ("woff", C.c_void_p), ("row_of", C.c_void_p), ("step_count", C.c_void_p),
                ("eval", C.c_void_p), ("epos", C.c_void_p)]


class RunStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("send_ms", C.c_float), ("spmv_ms", C.c_float), ("apply_ms", C.c_float),
                ("total_ms", C.c_float), ("spmv_launches", C.c_int32), ("rowblock_ms", C.c_float),
                ("wave_ms", C.c_float), ("giant_ms", C.c_float), ("rowblock_launches", C.c_int32),
                ("wave_launches", C.c_int32), ("giant_launches", C.c_int32), ("sparse_exchanges", C.c_int32)]


class EngineOptions(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("debug_flags", "wave16_form", "rowwave_form", "persist_per_cu", "giant_maps", "ordered_giant_two_pass",
                                         "fuse_apply_send", "untiled_pass_plain", "last_rows_lanes", "push_edge_permille", "bits_step_edges",
                                         "sparse_step_edges", "iteration_trace", "ablate_cold_from", "ablate_cold_short", "two_stage_head_permille", "giant_stream", "sweep_form", "blocked_form", "guided_pull")] + [("reserved_", C.c_int32 * 12)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int))

# name -> (restype, argtypes); every symbol include/graphmat_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "gm_last_error": (C.c_char_p, []),
    "gm_version": (C.c_int, []),
    "gm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gm_set_device": (C.c_int, [C.c_int]),
    "gm_vertex_to_native": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "gm_native_to_vertex": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "gm_mtx_read": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(_P),
                              C.POINTER(_P), C.POINTER(_P)]),
    "gm_host_free": (None, [_P]),
    "gm_edgelist_read": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int64), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "gm_edgelist_write": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, _P]),
    "gm_graph_create": (C.c_int, [C.POINTER(_P), C.POINTER(GraphDesc), C.c_int64, _P, _P, _P, _P]),
    "gm_graph_destroy": (C.c_int, [_P]),
    "gm_graph_desc": (C.c_int, [_P, C.POINTER(GraphDesc)]),
    "gm_graph_csr": (C.c_int, [_P, C.c_int, C.POINTER(Csr)]),
    "gm_graph_sweep": (C.c_int, [_P, C.POINTER(Sweep)]),
    "gm_graph_blocked": (C.c_int, [_P, C.POINTER(Blocked)]),
    "gm_graph_rowbits_all": (C.c_int, [_P, C.POINTER(_P)]),
    "gm_graph_tiles": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "gm_graph_tile": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(Csr), C.POINTER(_P)]),
    "gm_graph_csr_to_host": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "gm_graph_relayout_like": (C.c_int, [_P, _P, _P]),
    "gm_graph_maps": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "gm_graph_maps_to_host": (C.c_int, [_P, _P, _P]),
    "gm_graph_set_vals": (C.c_int, [_P, C.c_int, _P]),
    "gm_graph_sync_tile_vals": (C.c_int, [_P, _P]),
    "gm_rmat_generate": (C.c_int, [C.c_int, C.c_uint64, C.c_int64, C.c_int64, _P, _P, _P, C.c_int, _P]),
    "gm_graph_set_exchange": (C.c_int, [_P, EXCHANGE_FN, _P]),
    "gm_graph_set_exchange_caps": (C.c_int, [_P, C.c_int]),
    "gm_graph_exchange_caps": (C.c_int, [_P]),
    "gm_dist_unique_id": (C.c_int, [_P, C.c_size_t]),
    "gm_dist_init": (C.c_int, [C.c_int, C.c_int, _P, C.c_size_t]),
    "gm_dist_finalize": (C.c_int, []),
    "gm_dist_init_from_env": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gm_dist_barrier": (C.c_int, []),
    "gm_dist_allgatherv_host": (C.c_int, [_P, C.c_int64, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "gm_dist_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gm_graph_use_rccl": (C.c_int, [_P]),
    "gm_graph_exchange_is_native": (C.c_int, [_P]),
    "gm_graph_exchange_counters": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "gm_graph_set_run_stream": (C.c_int, [_P, _P]),
    "gm_run_degree": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int), _P]),
    "gm_run_pagerank": (C.c_int, [_P, _P, C.c_float, C.c_int, C.POINTER(C.c_int), _P]),
    "gm_run_bfs": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_int), _P]),
    "gm_run_sssp": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_int), _P]),
    "gm_run_sgd": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), _P]),
    "gm_run_rmse": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "gm_run_sgd_bipartite": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                       C.POINTER(C.c_int), _P]),
    "gm_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "gm_graph_engine_options": (C.c_int, [_P, C.POINTER(EngineOptions)]),
    "gm_graph_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "gm_reset_options": (C.c_int, []),
    "gm_graph_enable_timing": (C.c_int, [_P, C.c_int]),
    "gm_graph_last_stats": (C.c_int, [_P, C.POINTER(RunStats)]),
    "gm_graph_workspace": (C.c_int, [_P, C.c_int, C.c_size_t, C.POINTER(_P)]),
    "gm_graph_adopt_workspace": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "gm_debug_counters": (C.c_int, [C.POINTER(C.c_int64)]),
    "gm_graph_exchange": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, C.POINTER(C.c_int)]),
    "gm_graph_has_exchange": (C.c_int, [_P]),
    "gm_graph_timing_enabled": (C.c_int, [_P]),
    "gm_graph_split": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gm_graph_note_set": (C.c_int, [_P, C.c_int, C.c_int64]),
    "gm_graph_note_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    "gm_graph_workspace_info": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "gm_graph_run_resources": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "gm_graph_record_stats": (C.c_int, [_P, C.POINTER(RunStats)]),
    "gm_reduce_sum_f64": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(C.c_double), _P]),
    "gm_reduce_sum_f32": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(C.c_double), _P]),
    "gm_count_less_u32": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_uint32, C.POINTER(C.c_int64), _P]),
    "gm_popcount_bits": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), _P]),
}

_lib = None


def lib():
    """Load the C-ABI library.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        so = os.environ.get("GRAPHMAT_HIP_LIBRARY") or SO  # (experiments: an ablation build, graphmat_amd/build.py)
        if so != SO:
            print("graphmat_amd: loading %s instead of the product library" % so, flush=True)
        if not os.path.exists(so):
            raise RuntimeError("libgraphmat_hip.so is missing: build it with `python -m graphmat_amd.build` "
                               "(hipcc, gfx950).  graphmat_amd has no CPU fallback.")
        # torch ships its own HIP runtime; import it first so this library binds to the same
        # libamdhip64 instead of bringing a second runtime into the process.
        import torch  # noqa: F401
        L = C.CDLL(so)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class GMError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise GMError("libgraphmat_hip error %d: %s" % (rc, lib().gm_last_error().decode()))
