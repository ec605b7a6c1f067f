/* graphmat_hip.h -- C-ABI of libgraphmat_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for GraphMat's generalized-SpMV hot path.  The
 * reference has no FFI layer (it is a header-only C++ template library), so the
 * seam is cut where the reference's host code meets its compute loops:
 *
 *   reference                                            this library
 *   ---------------------------------------------------  ---------------------------
 *   Graph::vertexToNative / nativeToVertex               gm_vertex_to_native / gm_native_to_vertex
 *     (include/Graph.h:111-150)
 *   load_edgelist, binary .mtx                           gm_mtx_read
 *     (include/GMDP/utils/edgelist.h:242-334)
 *   Graph::ReadEdgelist -> SpMat/DCSCTile ctor,          gm_graph_create (device-side CSR build)
 *     Transpose (include/Graph.h:210-246,
 *     include/GMDP/matrices/DCSCTile.h:241-381)
 *   run_graph_program for the example programs            gm_run_pagerank / gm_run_degree /
 *     (include/GraphMatRuntime.h:93-279 driving           gm_run_bfs / gm_run_sssp / gm_run_sgd /
 *      include/SPMV.h:41-95 ->                            gm_run_rmse  ("fixed menu": the programs
 *      include/GMDP/singlenode/spmspv.h:39-86,            of src/PageRank.cpp, src/BFS.cpp,
 *      spmspv3.h:38-90, intersectreduce.h:39-66)          src/SSSP.cpp, src/SGD.cpp)
 *   MapReduce (include/GMDP/singlenode/reduce.h:51-99)   gm_reduce_*
 *
 * Arbitrary user vertex programs cannot cross a C ABI (their functors must be
 * compiled for the device); for them the C++ headers include/GraphMatRuntime.h,
 * Graph.h, GraphProgram.h instantiate the same kernel templates in the
 * application's translation unit (hipcc --hipstdpar) and use this C-ABI for the
 * graph, memory and utility kernels.
 *
 * Conventions: every function returns 0 on success and a GM_ERR_* code
 * otherwise (gm_last_error() gives text); nothing throws; plain pointers and
 * sizes only.  Pointers named d_* are DEVICE pointers, h_* HOST pointers.  The
 * caller owns every buffer it passes; the library owns what it allocates inside
 * a gm_graph_t.  `stream` is a hipStream_t (NULL = default stream).  One host
 * thread per process drives a given graph.
 *
 * Vertex ids: the API speaks GraphMat's two id spaces: "vertex" ids are the
 * 1-based ids of the .mtx file, "native" ids are the 0-based permuted ids in
 * which all device arrays are laid out (native = gm_vertex_to_native(v)-1).
 */
#ifndef GRAPHMAT_HIP_H_
#define GRAPHMAT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_OK 0
#define GM_ERR_INVALID 1
#define GM_ERR_HIP 2
#define GM_ERR_NOMEM 3
#define GM_ERR_UNSUPPORTED 4
#define GM_ERR_IO 5

typedef void* gm_stream_t; /* hipStream_t */
typedef struct gm_graph gm_graph_t;

const char* gm_last_error(void);
int gm_version(void); /* 100 + the build round whose struct layouts these are (107: gm_sweep_t with nsub / stride / hot_words / waves and the short rows' stream groups, 32-byte gchunk_state records) */
int gm_device_count(int* count);
int gm_set_device(int device);

/* ---- id permutation: include/Graph.h:111-150 ------------------------------------
 * nparts = num_threads * 16 * nranks of the GraphMat layout being reproduced.
 * 1-based in, 1-based out. */
int gm_vertex_to_native(int vertex, int nparts, int len);
int gm_native_to_vertex(int native, int nparts, int len);

/* ---- binary .mtx reader: include/GMDP/utils/edgelist.h:242-334 --------------------
 * header (int32 m,n,nnz) + nnz x (int32 src, int32 dst, val[val_bytes]); the header
 * count governs.  *nv = max(m,n) (Graph::ReadMTX squares the matrix).  Arrays are
 * malloc'ed; release each with gm_host_free. */
int gm_mtx_read(const char* path, int val_bytes, int* nv, int64_t* nnz, int32_t** h_src, int32_t** h_dst,
                void** h_val);
void gm_host_free(void* p);

/* ---- edge-list files in every variant the reference reads and writes
 * (load_edgelist / write_edgelist, include/GMDP/utils/edgelist.h:89-334; the formats
 * src/graph_converter.cpp:161-222 converts between) -----------------------------------
 * binary: optional header of three int32 (m, n, nnz), then records (int32 src, int32 dst
 *         [, value]);  text: optional header line "m n nnz", then lines "src dst [value]".
 * val_kind says how a value is stored/parsed; GM_VAL_RAW(bytes) is an opaque fixed-size
 * value (binary files only).  Reading: without a header m = largest src, n = largest dst and
 * records run to the end of the file; without weights every value is 1 (of val_kind).  With a
 * header the header's count governs, as in gm_mtx_read.  Arrays are malloc'ed (gm_host_free). */
#define GM_VAL_I32 1
#define GM_VAL_U32 2
#define GM_VAL_F32 3
#define GM_VAL_F64 4
#define GM_VAL_RAW(bytes) (0x100 + (bytes))
int gm_edgelist_read(const char* path, int binary, int header, int weights, int val_kind, int* m, int* n,
                     int64_t* nnz, int32_t** h_src, int32_t** h_dst, void** h_val);
/* text values are printed like the reference does: %d, %u, %.8f (float), %.15lf (double) */
int gm_edgelist_write(const char* path, int binary, int header, int weights, int val_kind, int m, int n, int64_t nnz,
                      const int32_t* h_src, const int32_t* h_dst, const void* h_val);

/* ---- graph ---------------------------------------------------------------------- */
#define GM_DIR_OUT 1 /* rows = destinations, cols = sources: GraphMat's AT, used by OUT_EDGES programs */
#define GM_DIR_IN 2  /* rows = sources, cols = destinations: GraphMat's A, used by IN_EDGES programs  */

/* Device order.  All device arrays (vertex state, x, y, presence bits, CSR rows and column
 * ids) are indexed by DEVICE ids.  GM_LAYOUT_NATIVE: device id = native id and the shard owns
 * [row_lo,row_hi) as given.  GM_LAYOUT_DEGREE (what bench.py uses): vertices are ranked by
 * total degree (descending, ties by native id) and rank k becomes device id
 * (k % nshards) * S + k / nshards with S = shard size (multiple of 64): every shard gets an
 * equal slice with its busiest vertices first (their x entries stay cache resident) and an
 * equal share of the edges; the library fills in row_lo/row_hi/ndevice.  In both layouts a
 * row's edges are stored -- and reduced -- in ascending NATIVE column order, so results do
 * not depend on the layout.  gm_graph_maps() gives the two maps. */
#define GM_LAYOUT_NATIVE 0
#define GM_LAYOUT_DEGREE 1

typedef struct {
  int32_t nvertices;     /* global vertex count */
  int32_t nparts;        /* layout parameter of the id permutation (see above)            */
  int32_t row_lo;        /* this shard owns device rows [row_lo,row_hi); multiples of 64   */
  int32_t row_hi;        /*   (input for GM_LAYOUT_NATIVE, output for GM_LAYOUT_DEGREE)    */
  int32_t directions;    /* GM_DIR_OUT | GM_DIR_IN                                         */
  int32_t val_bytes;     /* sizeof(edge value), 0 = drop edge values                       */
  int32_t ids_on_device; /* 1: src/dst/val are device pointers, 0: host pointers           */
  int32_t ids_are_native;/* 1: src/dst are already 0-based native ids (skip permutation)   */
  int32_t layout;        /* GM_LAYOUT_NATIVE or GM_LAYOUT_DEGREE                           */
  int32_t nshards;       /* GM_LAYOUT_DEGREE: number of shards (GPUs), >= 1                */
  int32_t shard;         /* GM_LAYOUT_DEGREE: which shard this graph object holds          */
  int32_t ndevice;       /* output: size of the device id space (>= nvertices)             */
  int32_t xchg_rows;     /* output: only the first xchg_rows entries of every shard's slice of
                            x (and their presence words) are ever read by other shards: with
                            GM_LAYOUT_DEGREE the vertices without any edge sit at the tail of
                            each slice.  Multiple of 64; = slice size for GM_LAYOUT_NATIVE.   */
  int32_t col_tiles;     /* column tiles of the GM_DIR_OUT adjacency (see gm_graph_tile): 0 = library default
                            (gm_set_option("col_tiles"), else environment GRAPHMAT_COL_TILES, else automatic:
                            by the live part of a 4-byte message vector: two tiles from 12 MiB (25 MiB on a sharded graph), three from 50 MiB, one per 40 MiB
                            from 180 MiB on, i.e. none up to RMAT-21, 2 at RMAT-22 .. 24, 3 at RMAT-25 and RMAT-26, 6 at RMAT-27 -- the
                            medium rows are swept over ~64 slices whatever the tile count (gm_graph_sweep), tiles only cut the longer
                            rows; with gm_set_option("sweep_slices", 0): one tile per 17 MiB once the live part reaches 60 MiB; with
                            edge values kept (such an adjacency is not swept): the same from 100 MiB on),
                            1 = none, 2..GM_MAX_TILES = that many.
                            GM_LAYOUT_DEGREE with one shard only (ignored otherwise).  Output: the number of tiles
                            built (1 = none).                                                                 */
  int32_t edges_local;   /* 0: every rank passes the whole edge list (edges of other shards' rows are dropped).
                            1: distributed build -- this rank passes only ITS part of the edge list (any disjoint
                            split; duplicates of an edge keep the order "rank, then position" in which the parts
                            would be concatenated).  Collective over the gm_dist communicator, which must have
                            nshards ranks with this process as rank `shard`; GM_LAYOUT_DEGREE only.  The degree
                            counts are summed over the ranks (all-reduce), every rank derives the same ranking,
                            and each edge travels to the shard that owns its row per direction, so a rank sorts
                            and keeps ~1/nshards of the edges and never holds the whole list.                    */
} gm_graph_desc_t;

/* One direction of the adjacency as laid out in HBM (see DESIGN.md "data layout"). */
typedef struct {
  int64_t nnz;            /* edges in this shard and direction                            */
  int32_t nrows;          /* row_hi - row_lo                                              */
  int32_t row_base;       /* row_lo                                                       */
  int32_t ncols;          /* size of the device id space (x has this many entries)        */
  int32_t val_bytes;
  const int64_t* rowptr;  /* [nrows+1]                                                    */
  const int32_t* colidx;  /* [nnz] DEVICE column ids, stored in ascending NATIVE column order
                             inside a row, duplicates in input order: the reference's
                             reduction order                                               */
  const void* vals;       /* [nnz] edge values or NULL                                    */
  const uint32_t* rowbits;/* presence bits of the non-empty rows: what y's presence vector is
                             after a multiply with every x entry present                    */
  /* work decomposition of the multiply+reduce kernels (include/graphmat/kernels.hpp):  */
  const int32_t* seg_row; /* [nseg+1] boundaries of runs of consecutive rows (local ids); a
                             run is either one row of more than GM_SHORT_ROW edges or up to 256
                             short rows holding < 2*GM_BLOCK_NNZ edges                      */
  int32_t nseg;
  const int32_t* blk_seg; /* [nblk] indices into seg_row of the short-row runs (row-blocks)  */
  int32_t nblk;
  const int32_t* mid_row; /* [nmid] rows with GM_SHORT_ROW < edges <= GM_GIANT_ROW: one wave each */
  int32_t nmid;
  const int32_t* giant_row;/* [ngiant] rows with more than GM_GIANT_ROW edges: one workgroup each */
  int32_t ngiant;
  /* giant rows are multiplied in two passes: a parallel pass over GM_GIANT_CHUNK-edge pieces
   * that writes the per-edge products to a scratch stream, then the ordered fold per row */
  const int32_t* gchunk_row;  /* [ngchunk] index into giant_row of the piece's row           */
  const int64_t* gchunk_edge; /* [ngchunk] first edge (absolute CSR position) of the piece    */
  const int64_t* gterm_off;   /* [ngiant+1] offset of each giant row's products in the scratch */
  int32_t ngchunk;
  int64_t giant_edges;        /* = gterm_off[ngiant]                                          */
  int32_t short_row;          /* rows of up to this many edges are in row-blocks, longer ones in mid/giant_row */
  int32_t nmid_long;          /* mid_row[0, nmid_long) holds every wave row of more than GM_LONG_MID edges (4x that
                                 when there are over 2^20 wave rows) (the list is
                                 in degree-ranked order, so this is a short prefix): one wave each; the rest are folded
                                 16 to a wave */
  const int32_t* umid_row;    /* tiled graphs, whole-graph CSR only: the wave rows that are NOT tiled (more than      */
  int32_t numid;              /*   GM_SHORT_ROW and at most tile_min_row edges), laid out like mid_row: the first    */
  int32_t numid_long;         /*   numid_long entries are the long ones                                              */
  int32_t tile_min_row;       /* rows of more than this many edges are multiplied tile by tile (-1: not tiled)       */
  int32_t hot_base;           /* the columns of this adjacency lie in device ids [hot_base, hot_base + hot_len) and   */
  int32_t hot_len;            /*   the busiest come first: the kernels keep x[hot_base ...] in LDS.  Whole graph:
                                 0 / ncols; a column tile: its slice of the device order                              */
  int32_t hot_slices;         /* sharded graphs (nshards = G > 1): the degree ranking is dealt over the G slices of x, */
  int32_t hot_stride;         /*   so the busiest entries are the first ones of EVERY slice: the hot set is the first
                                 kHot / hot_slices entries of each slice [q * hot_stride, ...), q < hot_slices (1 / 0:
                                 one slice)                                                                           */
  void* gchunk_state;         /* [ngchunk * 8] 32-byte records the giant-row kernels keep between passes, one per 512-product sub-piece
                                 (kernels.hpp: gchunk_state: binade hints and ulp-maps of the exact float replay); scratch owned by the graph */
  int64_t edges_blk;          /* edges of this adjacency by the kernel class that multiplies them: row-blocks (rows of up   */
  int64_t edges_wave16;       /*   to short_row edges), 16-rows-per-wave rows, one-wave-per-row rows (nmid_long), and -- the */
  int64_t edges_wave;         /*   rest of nnz -- giant rows                                                                */
  int32_t rows_keep_stream;   /* a column tile built with row classes fixed per ROW: a row is a one-wave-per-row / giant row in    */
  int32_t cold_from;          /*   every tile or in none, so those kernels and the others never share a y entry within an iteration.
                                 cold_from: ablation builds only (0 otherwise) -- columns >= cold_from are not gathered            */
} gm_csr_t;

#define GM_GIANT_CHUNK 4096 /* edges per piece of the parallel giant-row pass */

#define GM_BLOCK_NNZ 1024   /* a row-block holds < 2*GM_BLOCK_NNZ edges and <= 256 rows     */
#define GM_LONG_MID 1024    /* see gm_csr_t.nmid_long */
#define GM_SHORT_ROW 64     /* rows up to this many edges are folded one lane per row        */
#define GM_GIANT_ROW 32768  /* rows above this get a workgroup of their own; chosen per graph and
                               direction (smallest power of two >= 4096 leaving <= 1024 such rows):
                               32768 is what RMAT-24..27 get, 8192 RMAT-22, 4096 RMAT-20           */

/* src/dst: 1-based vertex ids as in the .mtx (or native ids, see desc).  Edges whose
 * row falls outside the shard are dropped per direction, so every rank passes the full
 * edge list (GM_LAYOUT_DEGREE needs it to rank the vertices) -- unless desc->edges_local is
 * set, in which case the call is a collective and each rank passes its own part.  The input
 * arrays are not modified. */
int gm_graph_create(gm_graph_t** g, const gm_graph_desc_t* desc, int64_t nnz, const int32_t* src,
                    const int32_t* dst, const void* val, gm_stream_t stream);
int gm_graph_destroy(gm_graph_t* g);
int gm_graph_desc(const gm_graph_t* g, gm_graph_desc_t* out);
int gm_graph_csr(const gm_graph_t* g, int direction, gm_csr_t* out);
/* ---- column tiles of a direction's adjacency ------------------------------------------------
 * The gathers of x are what bounds the multiply on large graphs (one 4-byte gather per edge; at
 * RMAT-26 x is 268 MB and 28 % of the gathers miss the 4 MB L2s).  With col_tiles = T > 1 the
 * library cuts the NATIVE id space into T contiguous ranges that serve equally many gathers (a
 * vertex counts as often as it is a column), lays the device order out as (tile, degree rank inside the tile) -- a tile's slice of x
 * is contiguous with its busiest entries first -- and keeps, for the rows of more than
 * GM_SHORT_ROW edges, one CSR per tile (only the edges whose column lies in the tile; same row
 * ids, columns still in ascending native order).  Because the tiles are native ranges, folding a
 * row's tile-0 edges, then its tile-1 edges, ... carrying the running value in y, is exactly the
 * reference's ascending-native-column fold: results do not change, but every pass gathers from
 * an x slice T times smaller (L2-resident hot part, LDS-resident hottest entries).
 * Only rows of more than tile_min_row edges (gm_set_option("tile_min_row"), default GM_TILE_MIN_ROW = the
 * short-row limit: every wave row) are tiled; the shorter rows keep the untiled kernels -- tiling them too only
 * multiplies their bookkeeping (gm_csr_t.umid_row lists the wave rows that stay untiled when the threshold is higher).
 * *d_prev_bits = presence bits of the rows that have an edge in an earlier tile (the rows whose
 * running value y already holds when tile `tile` is multiplied). */
#define GM_MAX_TILES 64
#define GM_TILE_MIN_ROW 64
int gm_graph_tiles(const gm_graph_t* g, int direction, int* ntiles); /* *ntiles = 1: not tiled */
/* ---- the row-stationary sweep (sliced-ELLPACK layout; round 5) ---------------------------------------------------------
 * Every row of the GM_DIR_OUT adjacency with more than GM_SHORT_ROW edges that is not a giant row -- 82 % of RMAT-26's edges
 * -- is multiplied by ONE persistent kernel (kernels.hpp: k_spmv_sell) instead of tile passes.  With
 * gm_set_option("sweep_slices", 1) (default) the device order of a large single-shard graph is cut into nslices contiguous
 * native ranges that serve equally many gathers (busiest vertices first inside a slice; a column tile = k consecutive
 * slices), and the library keeps those rows' edges a second time, laid out for the sweep: workgroup w of 256 owns the rows of
 * length rank r with r % 256 == w, keeps their running values in LDS (long rows: in registers) and walks the slices in
 * ascending order -- every workgroup at about the same slice at any time, so the chip gathers from ~1.3 MB of x at a time
 * (L2-resident), the slice's busiest entries from LDS -- and a row is still folded in ascending native column order: slices are
 * native ranges taken in order, and inside a (row, slice) PIECE the edges keep their CSR order.  `nsets` launches cover all
 * rows (a workgroup holds acc_rows medium + long_slots long rows per launch: rank r -> workgroup r % 256, set (r / 256) % nsets).
 *   MEDIUM rows (at most long_row edges: chosen per graph so that a slice's longest piece does not exceed a wave's share of a
 *   block -- 2048 at RMAT-26, 512 at RMAT-24; gm_set_option("sweep_long_row")): the pieces of a (set, workgroup, slice) BLOCK are sorted by length
 *   (descending) and cut into GROUPS of 64, lane = piece.  Group g occupies scol[gbase[g] .. gbase[g + 1]): first a META row of 64
 *   entries -- GM_SWEEP_PAD | width << 16 | first << 15 | slot: the group's width (rows that follow = its longest piece), whether
 *   this is the lane's row's first piece (its first message is assigned, not reduced: the reference has no additive identity)
 *   and the row's slot in the workgroup's accumulator array (0x7fff: the lane has no piece) -- then `width` rows stored
 *   transposed, scol[gbase[g] + (1 + k) * 64 + lane] = the k-th column of the lane's piece as a BYTE offset into a 4-byte message
 *   vector (column << 2), padded to the width with entries that carry GM_SWEEP_PAD (bit 31; the rest = the slice's first
 *   column): one coalesced 256-byte load hands every lane the next edge of ITS piece, the fold is one instruction per 64 edges,
 *   and the stream describes itself -- the kernel loads nothing else.  sval: the edge values in the same positions (val_bytes =
 *   4) or NULL.  Blocks are stored in (set, workgroup, slice) order, so a block's groups are contiguous; wfirst[((set * 256 + w)
 *   * nslices + s) * 17 + v] = first group of wave v of 16 (entry 16 = end of the block), wrow = the same as positions in scol /
 *   64: contiguous ranges balanced by rows.  row_of_slot[(set * 256 + w) * acc_rows + slot] = local row id (-1: none).
 *   LONG rows (more than long_row edges; at most long_slots per workgroup and set): too few and too uneven for groups.
 *   lcol (column << 2) / lval hold a block's edges in (slot, ascending native column) order, lps[((set * 256 + w) * nslices + s)
 *   * long_slots + j] = first entry of slot j's piece (the next entry ends it; the last entry of all = nedges_long); per
 *   slice all waves gather a block's messages into an LDS stage and the workgroup's last long_slots threads fold one piece each.
 *   lrow_of_slot[(set * 256 + w) * long_slots + j] = local row id (-1: none).  max_long_block = edges of the largest block.
 *   src_pos / lsrc_pos (graphs that keep edge values only): the CSR position every entry came from (0xffffffff: padding), so
 *   that rewritten edge values can be brought over (gm_graph_sync_tile_vals).
 * Single shard, GM_DIR_OUT only.  nrows = 0: the graph has no such structure (nslices and slice_base are valid whenever the
 * device order was sliced). */
typedef struct {
  int32_t nrows;          /* swept rows: medium + long */
  int32_t nrows_long;
  int32_t nsets;          /* launches */
  int32_t nslices;
  int32_t acc_rows;       /* medium-row slots per workgroup and launch (= GM_SWEEP_ACC_ROWS) */
  int32_t long_slots;     /* long-row slots per workgroup and launch (= GM_SWEEP_LONG_SLOTS) */
  int32_t max_long_block; /* most long-row edges of one (set, workgroup, slice) block */
  int32_t val_bytes;      /* 0, or 4: sval / lval hold the edge values */
  int32_t short_row;      /* the structure holds the rows of more than short_row edges ... */
  int32_t long_row;       /* ... that are not giant; those of more than long_row edges are the LONG rows */
  int64_t nedges;         /* edges of the medium rows */
  int64_t nedges_long;
  int64_t nentries;       /* entries of scol (edges + padding) */
  int64_t ngroups;
  const uint32_t* scol;
  const uint32_t* sval;
  const uint32_t* gbase;  /* [ngroups + 1] */
  const uint32_t* wrow;   /* [nsets * 256 * nslices * 17] */
  const uint32_t* wfirst; /* [nsets * 256 * nslices * 17] */
  const int32_t* row_of_slot;
  const uint32_t* lcol;
  const uint32_t* lval;
  const uint32_t* lps;    /* [nsets * 256 * nslices * long_slots + 1] */
  const int32_t* lrow_of_slot;
  const int32_t* slice_base; /* [nslices + 1] first device id of every slice */
  const uint32_t* src_pos;
  const uint32_t* lsrc_pos;
  /* GIANT rows (the whole-graph CSR's giant_row list): they keep their own fold passes (the exact replay / the ordered fold of
     their products stream, gm_csr_t.gterm_off), but the sweep can do the gathers for them, slice by slice with its hot sets:
     gcol[i] (column << 2) / gval[i] = the giant rows' edges sorted by slice (inside a slice: row, then ascending native column),
     gdst[i] = the edge's place in the products stream, gslice[s] = first entry of slice s (gslice[nslices] = ngiant_edges);
     a slice's entries are dealt evenly over the 256 workgroups.  NULL / 0: the giant rows gather for themselves. */
  int64_t ngiant_edges;
  const uint32_t* gcol;
  const uint32_t* gval;
  const uint32_t* gdst;
  const uint32_t* gslice;
  const uint32_t* gsrc_pos;
  /* SHARDED graphs (round 6; nsub = nshards > 1).  Every owner's range [q * stride, (q + 1) * stride) of the device order is laid
     out [slice 0][slice 1] ... (busiest vertices first inside a slice), the j-th busiest vertex of slice t sitting at position
     slice_base[t] + j / nsub of owner (j + t) % nsub: a slice of the message vector is then the SAME sub-range
     [slice_base[t], slice_base[t + 1]) of all nsub owners' ranges (slice_base holds positions inside an owner's range, not device
     ids), the structure is built from the shard's own rows, and the column entries (scol / lcol / gcol) carry the hot-set
     decision made at build time: bit 30 set = the message is in the workgroup's LDS at byte offset (entry & 0x3ffffffc), i.e.
     word q * hq + j for position slice_base[t] + j < hq of owner q, hq = min(slice length, hot_words / nsub); clear = byte offset
     into the message vector (column << 2, columns below 2^28).  Padding = GM_SWEEP_PAD | bit 30.  nsub <= 1: the single-shard
     form described above (hot entries are recognised by their offset). */
  int32_t nsub;
  int32_t stride;     /* rows per owner (= row_hi - row_lo) */
  int32_t hot_words;  /* LDS words the build assumed for a slice's hot entries (the kernel must load at least as many) */
  int32_t waves;      /* waves per workgroup the blocks' groups are dealt over (wrow / wfirst keep 17 entries per block): 16, or 12 / 8 with
                         gm_set_option("sweep_waves"): a smaller sweep workgroup that leaves room on every CU for another kernel */
  /* SHORT rows riding the sweep (round 6, last session; single-shard structures; gm_set_option("sweep_stream", 0) builds none).  The rows of
     1 .. short_row edges -- 15 % of RMAT-26's edges, whose gathers from the whole message vector cost the row-block kernel 35 % of an iteration --
     are listed in device order (srow[i], i < nshort_rows; soff[i] = edges of the rows before i) and cut into BINS of consecutive rows (bin b =
     the rows with soff[i] / bin_cap == b: at most bin_cap + 63 edges; sbin_row[b] = its first row).  Bin b belongs to workgroup b % 256 of the
     FIRST launch, whose (workgroup, slice) blocks get extra STREAM groups behind their medium groups: meta row = GM_SWEEP_PAD | bit 30 | width
     << 16 | 0x7fff in every lane (no lane has a slot), bit 15 of lanes 0 .. 31 spelling the group's first row `first` of the products stream;
     the `width` <= stream_width rows that follow hold the block's short-row edges, 64 per row, in (bin, row, ascending native column) order
     (padding only behind the block's last edge).  wrow_stream = the waves' row ranges WITH those groups (wrow: without -- kernel forms that do
     not store products never see them).  The sweep gathers for the stream rows like for any entry and stores the product of row k, lane l at
     sterms[(first + k) * 64 + l] (nstream_slots entries; kernels.hpp: k_spmv_sell_stream); a block's products are contiguous, so bin b's
     products of slice s are the schunk[(b * nslices + s) * 2 + 1] entries from schunk[(b * nslices + s) * 2] on, and sinv[that position] =
     where the product belongs in the bin's CSR order (soff of its row + index in the row - b * bin_cap).  kernels.hpp: k_short_fold
     folds a bin from there in ascending native column order.  nstream = 0: none built. */
  int64_t nstream;        /* edges of the short rows */
  int64_t nstream_slots;  /* entries of the products stream */
  int32_t nshort_rows;
  int32_t nbins;
  int32_t bin_cap;        /* = GM_STREAM_BIN */
  int32_t stream_width;   /* rows of a full stream group (= GM_STREAM_WIDTH) */
  const int32_t* srow;       /* [nshort_rows] local row ids, ascending */
  const uint32_t* soff;      /* [nshort_rows + 1] */
  const uint32_t* sbin_row;  /* [nbins + 1] */
  const uint32_t* schunk;    /* [nbins * nslices * 2] */
  const uint16_t* sinv;      /* [nstream_slots] */
  const uint32_t* wrow_stream; /* [nsets * 256 * nslices * 17] */
} gm_sweep_t;
#define GM_STREAM_BIN 12288   /* edges of a bin of short rows (k_short_fold keeps a bin's products in LDS) */
#define GM_STREAM_WIDTH 8     /* rows (of 64 entries) of a full stream group */
#define GM_SWEEP_HOT 0x40000000u
#define GM_MAX_SLICES 128
#define GM_SWEEP_ACC_ROWS 10048
#define GM_SWEEP_LONG_SLOTS 512
#define GM_SWEEP_PAD 0x80000000u
#define GM_SWEEP_POOL 30848      /* 4-byte LDS words shared by the slice's hot entries and the long rows' stage */
#define GM_SWEEP_MAX_STAGE 14336 /* largest stage (words): larger blocks are staged in chunks */
#define GM_SWEEP_POOL_SPARSE (GM_SWEEP_POOL - 768) /* the pool of the sparse-x form: 314 + 448 words go to its presence bits */
#define GM_SWEEP_POOL_W12 21504  /* the pool of the 768-thread form (gm_sweep_t.waves = 12): with the accumulators 126 208 bytes of LDS */
#define GM_SWEEP_MAX_STAGE_W12 9216
int gm_graph_sweep(const gm_graph_t* g, gm_sweep_t* out);
int gm_graph_tile(const gm_graph_t* g, int direction, int tile, gm_csr_t* out, const uint32_t** d_prev_bits);
/* ---- the short rows of a graph WITHOUT skew as a column-blocked stream (round 5, last session) -----------------------------
 * When nearly every edge of a large single-shard graph sits in a short row (at most GM_SHORT_ROW edges: no vertex is hot, every
 * gather of the row-block kernel misses -- the shape of the reference's test/generator.h:73-105) the library keeps the short
 * rows' edges a second time, laid out for ONE persistent kernel (kernels.hpp: k_spmv_blocked): the short rows, in device order,
 * are cut into BLOCKS of GM_BLOCKED_ROWS rows whose running values fill a workgroup's LDS; the 256 workgroups take the blocks of a
 * PASS side by side and walk the graph's slices (gm_sweep_t.slice_base: ascending native ranges) together -- a per-XCD counter per
 * (pass, slice) step keeps them within `window` slices of each other (bounded wait), so the slice everybody gathers from is L2
 * resident.  A (block, slice) SEGMENT holds its entries row after row, a row's entries in CSR order (ascending native column):
 * a row folded segment by segment is folded in the reference's order.  Entry i: ecol[i] = device column, erow[i] = row inside the
 * block | 0x8000 for the row's first edge (its message is assigned, SPMV.h:54-59).  woff[(block * nslices + slice) * 17 + w] = where
 * wave w of 16 starts inside the segment (equal shares moved to the next row border; [16] = the segment's end).
 * row_of[block * GM_BLOCKED_ROWS + k] = the row (relative to row_lo) of the block's k-th row.  A graph that keeps 4-byte edge values
 * carries them in the entries' positions (eval, +4 B per edge, and epos for refreshing them); other value widths are not laid out this way.  gm_set_option("blocked_rows", 0 = automatic (>= 90 % of the edges in short rows, >= 48 MiB of live
 * 4-byte messages), 1 = whenever the graph has slices, -1 = never).  nrows = 0: not built.  Prototype and measurements:
 * tools/blocked_bench.hip, profiles/r05_short_rows_blocked_stream_prototype.md. */
typedef struct gm_blocked {
  int32_t nrows;      /* short rows covered (every row of 1 .. short_row edges) */
  int32_t nblocks;    /* ceil(nrows / GM_BLOCKED_ROWS) */
  int32_t nslices;
  int32_t short_row;
  int32_t nsteps;     /* passes * nslices: words per XCD of step_count */
  int32_t val_bytes;  /* 4: the entries carry the edge values (eval), 0: the graph keeps none */
  int64_t nentries;
  const uint32_t* ecol;
  const uint16_t* erow;
  const uint32_t* woff;
  const int32_t* row_of;
  uint32_t* step_count; /* 8 * nsteps words: workgroups of XCD k that have finished step t (cleared by the engine before a launch) */
  const uint32_t* eval; /* val_bytes == 4: entry i's edge value */
  const uint32_t* epos; /* val_bytes == 4: the CSR position entry i came from (gm_graph_sync_tile_vals refreshes eval through it) */
} gm_blocked_t;
#define GM_BLOCKED_ROWS 32768
int gm_graph_blocked(const gm_graph_t* g, gm_blocked_t* out);
/* rowbits of GM_DIR_OUT | rowbits of GM_DIR_IN (graphs built with both directions; ALL_EDGES programs) */
int gm_graph_rowbits_all(const gm_graph_t* g, const uint32_t** d_bits);
/* Rebuild g's adjacency in the device order of `like` (same vertex count and nparts, single
 * shard): afterwards device arrays of the two graphs are interchangeable, which is what
 * Graph::shareVertexProperty (include/Graph.h:300-305 of the reference) needs. */
int gm_graph_relayout_like(gm_graph_t* g, const gm_graph_t* like, gm_stream_t stream);
/* d_dev_of_native[native id] = device id (nvertices entries), d_native_of_dev[device id] =
 * native id or -1 for an unused slot (ndevice entries).  Both NULL for GM_LAYOUT_NATIVE
 * (identity).  Device pointers owned by the graph. */
int gm_graph_maps(const gm_graph_t* g, const int32_t** d_dev_of_native, const int32_t** d_native_of_dev);
/* host copies of the maps (identity is written for GM_LAYOUT_NATIVE); either may be NULL */
int gm_graph_maps_to_host(const gm_graph_t* g, int32_t* h_dev_of_native, int32_t* h_native_of_dev);
/* copy a direction's CSR back to the host, rows and columns in device ids (tests, Graph::getEdgelist) */
int gm_graph_csr_to_host(const gm_graph_t* g, int direction, int64_t* h_rowptr, int32_t* h_colidx, void* h_vals);

/* overwrite a direction's edge values from a host array laid out like gm_graph_csr_to_host's
 * (Graph::applyToAllEdges); the column tiles' copies of the values follow */
int gm_graph_set_vals(gm_graph_t* g, int direction, const void* h_vals);
/* After edge values were rewritten in place on the device (through gm_csr_t.vals of gm_graph_csr): bring the column
 * tiles' copies of the GM_DIR_OUT values up to date.  No-op for untiled graphs and graphs without edge values. */
int gm_graph_sync_tile_vals(gm_graph_t* g, gm_stream_t stream);

/* ---- synthetic RMAT edges generated on the device ---------------------------------
 * Same integer-only definition as graphmat_amd/generators.py:rmat_edges (bit-identical).
 * Writes edges [first_edge, first_edge+count) of the scale/edge-factor/seed stream as
 * 1-based ids; d_val may be NULL.  weights_mode 0: 1, 1: 1 + hash%127. */
int gm_rmat_generate(int scale, uint64_t seed, int64_t first_edge, int64_t count, int32_t* d_src, int32_t* d_dst,
                     int32_t* d_val, int weights_mode, gm_stream_t stream);

/* ---- multi-GPU hook ---------------------------------------------------------------
 * With row-sharded graphs the message vector must be made globally visible between
 * send and multiply, and the convergence flag combined.  The library calls back;
 * the caller implements it with its collective library (bench.py: torch.distributed
 * over RCCL).  kind GM_XCHG_MESSAGES: d_ptr = x values (elt_bytes each, nvertices
 * entries, own slice [row_lo,row_hi) valid), d_bits = presence bit vector
 * (nvertices/32 words, own words valid).  kind GM_XCHG_CONVERGED: h_flag points to a
 * host int (1 = locally converged) to be AND-reduced in place.  Return 0 on success. */
#define GM_XCHG_MESSAGES 0
#define GM_XCHG_CONVERGED 1
/* Overlapped form, used by fixed-count ALL_VERTICES programs when the caller has also adopted a
 * second message buffer (workspace slot 9): the rows are processed in two stages -- first the
 * many rows with few edges, then the few busy rows -- and each stage's part of the NEXT
 * iteration's message vector is exchanged while the other stage computes.
 * GM_XCHG_PART: d_ptr = base of the message buffer to fill (slot 1 or slot 9), h_flag[0] = first
 * row of the part inside every shard's slice, h_flag[1] = number of rows; every shard's rows
 * [h_flag[0], h_flag[0]+h_flag[1]) of its slice must reach all shards.  May return before the data
 * has arrived.  GM_XCHG_WAIT: work enqueued afterwards on the library's stream must see all parts.
 * No presence bits travel (every vertex sends). */
#define GM_XCHG_PART 2
#define GM_XCHG_WAIT 3
/* Sparse exchange for ACTIVE_ONLY programs with a small active set (the reference compresses a segment
 * before sending it when few entries are set, include/GMDP/vectors/DenseSegment.h:532-538,665-700):
 * GM_XCHG_STATE: the convergence flag and the size of the local active set in one step.  In: h_flag[0] =
 *   locally converged (0/1), h_flag[1] = vertices in this shard's next active set.  Out: h_flag[0] = AND
 *   over shards, h_flag[1] = largest such count over the shards, h_flag[2] = their sum (clamped to INT_MAX).
 * GM_XCHG_GATHER: in-place all-gather of equal blocks: d_ptr = a buffer of nshards blocks of
 *   h_flag[0] entries of elt_bytes bytes (workspace slot GM_WS_GATHER), this shard's block filled; afterwards
 *   every shard holds all blocks.  The library packs (device id, message) entries into it and scatters them
 *   into x itself.
 * An exchange implementation announces that it handles these with gm_graph_set_exchange_caps. */
#define GM_XCHG_STATE 4
#define GM_XCHG_GATHER 5
#define GM_XCAP_SPARSE 1 /* GM_XCHG_STATE and GM_XCHG_GATHER are implemented */
#define GM_WS_GATHER 12  /* workspace slot of the gather buffer (adopt it when the collective library must know the buffer) */
typedef int (*gm_exchange_fn)(void* ctx, int kind, void* d_ptr, int64_t elt_bytes, uint32_t* d_bits, int* h_flag);
int gm_graph_set_exchange(gm_graph_t* g, gm_exchange_fn fn, void* ctx);
int gm_graph_set_exchange_caps(gm_graph_t* g, int caps); /* GM_XCAP_*; gm_graph_set_exchange resets them to 0 */
int gm_graph_exchange_caps(const gm_graph_t* g);

/* ---- the same exchange, natively on RCCL (graphmat_amd/csrc/gm_dist.hip) ---------------------------
 * Replaces the reference's MPI transport of the multi-rank SpMSpV (include/GMDP/multinode/spmspv.h:62-116)
 * and its convergence Allreduce (include/GraphMatRuntime.h:226): one process per GPU, shard r of a
 * GM_LAYOUT_DEGREE graph on rank r.  The library implements GM_XCHG_MESSAGES / PART / WAIT / CONVERGED
 * itself with ncclAllGather / ncclAllReduce on HIP streams (the overlapped parts on a side stream,
 * ordered with events): nothing but the convergence flag crosses to the host per iteration.
 *   rank 0:     gm_dist_unique_id(id, 128); hand `id` to the other ranks (launcher's job: MPI_Bcast,
 *               torch.distributed broadcast, a file ...)
 *   every rank: gm_set_device(local gpu); gm_dist_init(rank, nranks, id, 128);
 *               gm_graph_create(... nshards = nranks, shard = rank ...); gm_graph_use_rccl(g);
 * RCCL (librccl.so.1) is loaded by gm_dist_init, not at library load. */
#define GM_DIST_ID_BYTES 128
int gm_dist_unique_id(void* out, size_t bytes);
int gm_dist_init(int rank, int nranks, const void* unique_id, size_t bytes);
int gm_dist_finalize(void);
int gm_dist_info(int* rank, int* nranks); /* *nranks = 0 before gm_dist_init */
/* What an application's MPI_Init can call (include/graphmat/mpi_single.h does): rank / size / local GPU from the
 * launcher's environment -- only from an unambiguous multi-rank launch: GRAPHMAT_NRANKS + GRAPHMAT_RANK, or a launcher's
 * own pair (torchrun WORLD_SIZE + RANK, Open MPI OMPI_COMM_WORLD_SIZE + _RANK, PMI_SIZE + PMI_RANK, srun
 * SLURM_STEP_NUM_TASKS + SLURM_PROCID; a size without its rank, or SLURM_NTASKS of an allocation, is not a launch);
 * GRAPHMAT_NRANKS=1 forces a single process.  The unique id travels through a rendezvous file (GRAPHMAT_RENDEZVOUS, else
 * /tmp/graphmat_rdv_<uid>_<job id of the launch>; node-local by default), created exclusively by rank 0 and accepted
 * only when fresh; setting the communicator up is bounded by GRAPHMAT_INIT_TIMEOUT seconds (default 180): an error,
 * not a hang.  A single process is left alone.  librccl is bound with dlopen at that moment; GRAPHMAT_RCCL_LIBRARY names
 * another library with the same entry points (the test suite's shared-memory stand-in, tests/support/). */
int gm_dist_init_from_env(int* rank, int* nranks);
int gm_dist_barrier(void);
/* all-gather of host buffers of different sizes: *all = malloc'ed concatenation in rank order (gm_host_free),
 * counts[r] = bytes contributed by rank r */
int gm_dist_allgatherv_host(const void* mine, int64_t my_bytes, void** all, int64_t* counts);
int gm_graph_use_rccl(gm_graph_t* g);
int gm_graph_exchange_is_native(const gm_graph_t* g);
/* out[0] exchange calls, out[1] overlapped parts started, out[2] bytes this rank contributed to all-gathers,
 * out[3] sparse (GM_XCHG_GATHER) exchanges */
int gm_graph_exchange_counters(const gm_graph_t* g, int64_t out[4]);
/* the stream the current run enqueues on (the header layer tells the library before it asks for exchanges) */
int gm_graph_set_run_stream(gm_graph_t* g, gm_stream_t stream);
/* Two-stage schedule of a direction's rows: "head" = rows [0, *row_split) holding at least
 * head_permille/1000 of the direction's edges (the degree-ranked device order puts the busy rows
 * first), "tail" = the rest.  On entry *row_split = 0 asks the library to choose, a positive value
 * (shards must agree on one split: the exchanged parts are the same rows of every slice) is
 * used as given.  *row_split is a multiple of 64; *blk_split = first row-block holding
 * a row >= *row_split (a straddling block belongs to the tail), *mid_split = first entry of
 * mid_row >= *row_split.  All giant rows must lie in the head (else GM_ERR_INVALID). */
int gm_graph_split(const gm_graph_t* g, int direction, int head_permille, int32_t* row_split, int32_t* blk_split,
                   int32_t* mid_split);
/* a few integers the header layer may park on a graph between runs (slot in [0, GM_NOTE_SLOTS));
 * get returns GM_ERR_INVALID while a slot has never been set; rebuilding the adjacency
 * (gm_graph_relayout_like) clears them */
#define GM_NOTE_SLOTS 8
int gm_graph_note_set(gm_graph_t* g, int slot, int64_t value);
int gm_graph_note_get(const gm_graph_t* g, int slot, int64_t* value);
/* what a workspace slot currently holds (external = adopted from the caller) */
int gm_graph_workspace_info(const gm_graph_t* g, int slot, void** d_ptr, size_t* bytes, int* external);

/* ---- fixed-menu vertex programs ------------------------------------------------------
 * Vertex state arrays are DEVICE arrays over the shard's rows in device order
 * (entry i = device id row_lo+i; see gm_graph_maps).  iterations <= 0 runs until convergence
 * (GraphMatRuntime.h:254-260); *iters_done (may be NULL) receives the count.
 * d_active: presence bit vector over the shard's rows (bit i&31 of word i>>5),
 * in/out, as Graph::active; NULL = all vertices active on entry. */

/* layout of class PR, src/PageRank.cpp:34-38 */
typedef struct { float pagerank; int32_t degree; } gm_pr_t;
/* Degree program (src/PageRank.cpp:53-79): IN_EDGES, sum of 1s -> vp.degree (out-degree). */
int gm_run_degree(gm_graph_t* g, gm_pr_t* d_vp, int iterations, int* iters_done, gm_stream_t stream);
/* PageRank program (src/PageRank.cpp:81-112): ALL_VERTICES, OUT_EDGES. */
int gm_run_pagerank(gm_graph_t* g, gm_pr_t* d_vp, float alpha, int iterations, int* iters_done, gm_stream_t stream);

/* layout of class BFSD2, src/BFS.cpp:40-45 (natural alignment: 4 bytes padding after depth) */
typedef struct { uint32_t depth; uint32_t pad_; uint64_t parent; uint64_t id; } gm_bfs_t;
/* BFS2 program (src/BFS.cpp:62-99): messages carry vp.id, reduce "a=b" (last writer in
 * native column order wins), do_every_iteration bumps current_depth starting at 1. */
int gm_run_bfs(gm_graph_t* g, gm_bfs_t* d_vp, uint32_t* d_active, int iterations, int* iters_done, gm_stream_t stream);

/* SSSP program (src/SSSP.cpp:64-90): uint32 distances, msg + int edge value, min. */
int gm_run_sssp(gm_graph_t* g, uint32_t* d_dist, uint32_t* d_active, int iterations, int* iters_done,
                gm_stream_t stream);

/* SGD / RMSE programs (src/SGD.cpp:77-156): vertex state is LatentVector<K> =
 * K reals then sqerr, i.e. (K+1) reals per vertex.  real_bytes 8 (reference) or 4. */
int gm_run_sgd(gm_graph_t* g, void* d_latent, int K, int real_bytes, double lambda, double step, int iterations,
               int* iters_done, gm_stream_t stream);
int gm_run_rmse(gm_graph_t* g, void* d_latent, int K, int real_bytes, gm_stream_t stream);
/* SGD on a BIPARTITE ratings graph split by users over the ranks of the gm_dist communicator: only the item side
 * travels (the reference's 2-D scheme also moves the smaller operand, include/GMDP/multinode/spmspv3.h:74-170).
 * Rank k passes a graph built (unsharded: nshards = 1) from the ratings of ITS users only, where the ranks' user sets
 * are contiguous ranges of the users' NATIVE ids, ascending with the rank; every rank holds all vertices' latent
 * vectors (d_latent over g's device order) and keeps the items' current.  d_item_rows[j] = g's device row of item j,
 * the same item order on every rank (nitems entries, device memory).  An item's ordered fold over its users is the
 * concatenation of the ranks' segments, so its running sum travels rank 0 -> 1 -> ... in `blocks` blocks of items (a
 * ring pipeline), the last rank broadcasts the complete sums, and every rank applies them to its copy of the items and
 * its own users' sums to its users: per rank and iteration 2 * nitems * (K + 1) * 4 bytes are received instead of
 * nvertices * K * 4, and the bits are those of one GPU.  Vertices must be users (sources only) or items (destinations
 * only).  K = 128 fp32.  With no communicator (or one rank) this is a single-GPU iteration.  gm_graph_note_get(g, 1)
 * afterwards: bytes this rank received per iteration. */
int gm_run_sgd_bipartite(gm_graph_t* g, void* d_latent, int K, int real_bytes, const int32_t* d_item_rows, int nitems, int blocks,
                         double lambda, double step, int iterations, int* iters_done, gm_stream_t stream);

/* runtime options.  "force_ordered" (0/1): run PageRank with the plain serial long-row fold
 * instead of the exact parallel replay (A/B check; results are bit-identical).  Graph-build experiments (read when a
 * graph is created): "short_row", "giant_row", "rank_by", "rank_cap", "col_tiles", "tile_min_row", "long_mid" (wave rows
 * of more than this many edges get a wave each instead of sharing one 16 to a wave; 0 = GM_LONG_MID rule), "tile_balance"
 * (1: column tiles serve equally many gathers, the default; 0: they hold equally many vertices with edges); "sgd_mfma"
 * (0/1: the dot products of K = 128 fp32 SGD on the matrix cores: an opt-in measurement form whose results deviate from the vector
 * form's -- the reference's -- bits by up to ~1e-5 of the vectors' scale, outside the 1e-6 bar; slower as well).
 * Every field of gm_engine_options_t below is also a key: gm_set_option then sets the PROCESS default, which a graph
 * uses unless gm_graph_set_option gave it a value of its own; the environment variable GRAPHMAT_OPTIONS="key=value,..."
 * sets such defaults for applications that cannot call this themselves (the reference's unchanged sources). */
int gm_set_option(const char* key, int value);

/* ---- engine options: how run_graph_program's iteration loop (include/graphmat/engine.hpp) schedules its kernels.
 * None of them changes a result; they choose between exact strategies (experiments, A/B tests, ablations).  The
 * engine reads them once per run through gm_graph_engine_options: the process defaults (gm_set_option) overlaid with
 * the graph's own values (gm_graph_set_option) -- no state lives in the header layer, so an application binary and
 * the library always agree. */
typedef struct {
  int32_t debug_flags;            /* dev::DBG_* of kernels.hpp: 16 no auxiliary stream, 32 no top-down steps, 64 no grouped wave rows, 128 no two-stage
                                     sharded schedule, 256 no on-demand messages, 512 no 16-rows-per-wave kernel, 1024 no column tiles, 2048 dense
                                     exchanges only, 4096 giant rows start with the head stage, 8192 long wave rows on the main stream; 1-8 and
                                     16384 are in-kernel ablations: -DGRAPHMAT_ABLATION builds only (a product kernel has no such argument).  Default 0 */
  int32_t wave16_form;            /* 16-rows-per-wave kernel: 0 = one workgroup per 64 rows with an 8192-entry LDS hot set; 2 = persistent 1024-thread
                                     workgroups with a 22528-entry hot set where that pays (large unsharded graphs); +16 = on graphs of any size.  Default 2 */
  int32_t rowwave_form;           /* row-blocks: 0 = one workgroup per block (k_spmv_rowblock); 4 = waves of persistent 1024-thread workgroups sharing a
                                     20480-entry LDS hot set where that pays; +16 = on graphs of any size.  Default 4 */
  int32_t persist_per_cu;         /* persistent kernels: workgroups per CU (0 = as many as the LDS allows).  Default 0 */
  int32_t giant_maps;             /* giant rows of float sums: 1 = the exact replay spread over many workgroups (per-piece ulp-maps), 0 = one workgroup
                                     walks the row.  Default 1 */
  int32_t ordered_giant_two_pass; /* giant rows of plain ordered folds: 1 = products by k_giant_terms, then k_giant_fold_ordered; 0 = one wave per row
                                     gathering by itself; 2 = as 1, and when the undeclared reduce_function answers like a float addition (dense x):
                                     the exact replay of the float sum, every 8192-product chunk of it proven with the program's own function
                                     (k_giant_verify_chunks), rows with a disagreeing chunk folded again in order.  Default 2 */
  int32_t fuse_apply_send;        /* 1 = apply of iteration i and send of iteration i+1 in one pass (fixed-count ALL_VERTICES programs that leave
                                     do_every_iteration to the base class).  Default 1 */
  int32_t untiled_pass_plain;     /* tiled graphs: the untiled short-row pass through the plain row-block kernel (1) or the persistent one (0).  Default 1 */
  int32_t last_rows_lanes;        /* a=b programs under a row filter: lanes per row of the grouped wave kernel (8 or 16).  Default 8 */
  int32_t push_edge_permille;     /* top-down steps while the active set owns less than this many thousandths of the edges.  Default 50 */
  int32_t bits_step_edges;        /* an active set too large to list bids from its bitmap while it owns at most this many out-edges.  Default 2097152 */
  int32_t sparse_step_edges;      /* ... and runs entirely on lists while it owns at most this many.  Default 1048576 */
  int32_t iteration_trace;        /* 1 = the reference's __TIMING lines per iteration (host synchronises after every phase).  Default 0, or 1 when the
                                     environment has GRAPHMAT_ITERATION_TRACE=1 */
  int32_t ablate_cold_from;       /* -DGRAPHMAT_ABLATION builds only (profiles/r04_cold_column_ablation.md); 0 */
  int32_t ablate_cold_short;      /* -DGRAPHMAT_ABLATION builds only; 0 */
  int32_t two_stage_head_permille;/* sharded two-stage schedule: the head stage = the first rows holding at least this many thousandths of the
                                     edges (gm_graph_split); the rest -- most of the rows -- is the tail stage whose messages travel while the head
                                     is multiplied.  Default 900: the tail then holds (almost) only short rows, so the 16-rows-per-wave kernel is
                                     not cut into two launches (shard of 8 of RMAT-26, compute only: 1.103 ms against 1.196 with 650 and 1.100 for
                                     the plain loop; profiles/r04_shard_emulation_rmat26.txt) */
  int32_t giant_stream;           /* without effect since round 5 (a third stream for the giant rows' passes of round 4's tiled sweep); kept so that
                                     the fields keep their places */
  int32_t sweep_form;             /* the swept multiply (engine.hpp: multiply_out_swept): bits 0-1 = where the short rows' pass runs: 0 on the main stream
                                     in front of the sweep (default), 1 on the auxiliary stream behind the giant rows' passes (next to the sweep),
                                     2 on the main stream behind the sweep; bit 2 = the long rows staged in rounds of 1024 entries (tests); bit 3 = the
                                     giant rows gather for themselves on the auxiliary stream (k_giant_terms) instead of the sweep gathering for them; bit 4 = the giant rows' gathers in a
                                     kernel of their own behind the sweep (k_giant_gather_sliced: their entries in slice order, on the auxiliary stream next to the short rows); bit 5 = a SPARSE
                                     message vector (ACTIVE_ONLY programs) does not take the sweep (k_spmv_sell_sparse); bit 6 = it does on graphs of any size (default: from 2e8 edges on); bit 7 = the short rows keep the row-block kernel (gm_sweep_t.nstream is not used); bit 8 = they ride the sweep whatever their number (default: from 2^25 short-row edges on) */
  int32_t blocked_form;           /* the column-blocked stream of the short rows (engine.hpp: multiply_out_blocked): bits 0-3 = window -- a workgroup starts a
                                     slice when all workgroups of its XCD have finished the one `window` slices back (default 2; 0 = workgroups not
                                     kept in step); bit 4 = batches of 4 x 64 entries instead of 2 x 64 */
  int32_t guided_pull;            /* ORDERED folds over a sparse x (ACTIVE_ONLY programs that declare nothing, running until convergence, unsharded): while the active
                                     set owns less than 2 % of the edges, the rows it reaches are marked first and only those are folded (engine.hpp: guided pull).
                                     0 = never, 1 = on graphs of at least 2^27 edges (default), 2 = on graphs of any size (tests) */
  int32_t reserved_[12];
} gm_engine_options_t;
/* the options a run on `g` uses (g may be NULL: the process defaults) */
int gm_graph_engine_options(const gm_graph_t* g, gm_engine_options_t* out);
/* an engine option for this graph only (key = a field name of gm_engine_options_t) */
int gm_graph_set_option(gm_graph_t* g, const char* key, int value);
/* every option of gm_set_option back to its documented default (tests call it from a fixture finaliser); graphs keep their own values */
int gm_reset_options(void);

/* ---- timing of the last gm_run_* call on this graph -------------------------------------
 * HIP-event times (ms) summed over iterations; kernel launches counted. */
typedef struct {
  int32_t iterations;
  float send_ms, spmv_ms, apply_ms, total_ms; /* spmv_ms = rowblock_ms + wave_ms + giant_ms */
  int32_t spmv_launches;
  float rowblock_ms, wave_ms, giant_ms;       /* the three multiply+reduce kernel classes, separately; giant_ms is the time of
                                                 the auxiliary stream: the giant-row passes and, in column-tile passes, the
                                                 long wave rows that overlap the other kernels there */
  int32_t rowblock_launches, wave_launches, giant_launches;
  int32_t sparse_exchanges;                   /* iterations whose messages travelled as lists (GM_XCHG_GATHER) */
} gm_run_stats_t;
int gm_graph_enable_timing(gm_graph_t* g, int on);
int gm_graph_last_stats(const gm_graph_t* g, gm_run_stats_t* out);

/* ---- services for the C++ header layer (include/graphmat/engine.hpp) -------------------------
 * Device scratch owned by the graph, grown on demand and reused across runs (the
 * reference allocates x/y per run_graph_program call, GraphMatRuntime.h:110-120).
 * slot in [0, GM_WS_SLOTS). */
#define GM_WS_SLOTS 20
int gm_graph_workspace(gm_graph_t* g, int slot, size_t bytes, void** d_ptr);
/* Let the caller provide a scratch slot (e.g. a torch tensor it also hands to its collective
 * library): slot 1 = message values x (nvertices * elt bytes), slot 2 = x presence bits
 * ((nvertices+31)/32+2 words).  The library uses the buffer while it is large enough and never
 * frees it.  d_ptr = NULL returns the slot to library ownership. */
int gm_graph_adopt_workspace(gm_graph_t* g, int slot, void* d_ptr, size_t bytes);
/* Per-graph run resources, created once with the graph and destroyed with it: an auxiliary
 * non-blocking hipStream_t (the giant-row passes overlap the other multiply kernels on it),
 * two hipEvent_t (fork / join of that stream) and 4096 bytes of pinned host memory (the
 * convergence flag and frontier statistics are copied there every iteration).  Returned as
 * void* so this header stays free of HIP types. */
int gm_graph_run_resources(gm_graph_t* g, void** aux_stream, void** fork_event, void** join_event, void** pinned);
/* counters of the giant-row kernel since the last call (then reset): 16-edge groups
 * out[0] taken by the exact parallel fp32 replay, out[1] folded serially */
int gm_debug_counters(int64_t out[4]);
/* invoke the exchange callback if one is set (no-op returning 0 otherwise) */
int gm_graph_exchange(gm_graph_t* g, int kind, void* d_ptr, int64_t elt_bytes, uint32_t* d_bits, int* h_flag);
int gm_graph_has_exchange(const gm_graph_t* g);
int gm_graph_timing_enabled(const gm_graph_t* g);
int gm_graph_record_stats(gm_graph_t* g, const gm_run_stats_t* st);

/* ---- reductions over device arrays (MapReduce, include/GMDP/singlenode/reduce.h:51-99) --- */
int gm_reduce_sum_f64(const double* d_x, int64_t n, int64_t stride_elems, double* h_out, gm_stream_t stream);
int gm_reduce_sum_f32(const float* d_x, int64_t n, int64_t stride_elems, double* h_out, gm_stream_t stream);
int gm_count_less_u32(const uint32_t* d_x, int64_t n, int64_t stride_elems, uint32_t bound, int64_t* h_out,
                      gm_stream_t stream);
int gm_popcount_bits(const uint32_t* d_bits, int64_t nbits, int64_t* h_out, gm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHMAT_HIP_H_ */
