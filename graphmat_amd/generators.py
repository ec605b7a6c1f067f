"""Synthetic edge lists (host side, numpy).

The small generators restate the shapes used by the reference's tests
(test/generator.h:43-167: identity, upper-triangular, dense, circular chain).
rmat_edges() is this project's own RMAT/Kronecker generator; it is defined with
integer arithmetic only so that the HIP generator (csrc/gm_rmat.hip) produces
bit-identical edges at any scale.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def identity_edges(n):
    i = np.arange(1, n + 1, dtype=np.int32)
    return n, i.copy(), i.copy(), np.ones(n, np.int32)


def upper_triangular_edges(n):
    s, d = np.triu_indices(n, k=1)
    return n, (s + 1).astype(np.int32), (d + 1).astype(np.int32), np.ones(s.size, np.int32)


def dense_edges(n):
    s, d = np.divmod(np.arange(n * n, dtype=np.int64), n)
    return n, (s + 1).astype(np.int32), (d + 1).astype(np.int32), np.ones(n * n, np.int32)


def chain_edges(n):
    i = np.arange(n, dtype=np.int64)
    return n, (i + 1).astype(np.int32), ((i + 1) % n + 1).astype(np.int32), np.ones(n, np.int32)


def uniform_out_regular_edges(n, k, seed=1):
    """Every vertex has exactly k out-edges to k DISTINCT destinations drawn uniformly from all n vertices: the shape of
    the reference's own random test graphs (test/generator.h:73-105, generate_random_edgelist(n, avg_nnz_per_row): per
    source k draws, a draw that repeats one of the row's earlier destinations is redrawn).  No skew at all: the
    opposite end of the spectrum from RMAT for everything that ranks vertices by degree.  1-based ids, values 1."""
    rng = np.random.default_rng(seed)
    k = min(k, n)
    dst = rng.integers(0, n, size=(n, k), dtype=np.int64)
    while True:  # redraw repeated destinations inside a row
        srt = np.sort(dst, axis=1)
        rows = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))[0]
        if rows.size == 0:
            break
        for r in rows:
            seen = set()
            for j in range(k):
                v = int(dst[r, j])
                while v in seen:
                    v = int(rng.integers(0, n))
                seen.add(v)
                dst[r, j] = v
    src = np.repeat(np.arange(1, n + 1, dtype=np.int64), k)
    return n, src.astype(np.int32), (dst.reshape(-1) + 1).astype(np.int32), np.ones(n * k, np.int32)


def splitmix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


# thresholds = floor(p * 2^32) for a=0.57, a+b=0.76, a+b+c=0.95
RMAT_T_A = 2448131358
RMAT_T_AB = 3264175144
RMAT_T_ABC = 4080218931


def rmat_edges(scale, edge_factor=16, seed=1, first_edge=0, num_edges=None, weights="ones"):
    """RMAT a/b/c/d = .57/.19/.19/.05; V = 2^scale, E = edge_factor*V.

    Edge e, level l (l=0 is the most significant id bit) draws the 32-bit value
    r = half (l&1) of splitmix64(splitmix64(seed) + 32*e + (l>>1));
    src_bit = r >= T_AB;  dst_bit = (T_A <= r < T_AB) or (r >= T_ABC).
    Ids are 1-based; duplicates and self loops are kept; no id scrambling.
    weights: "ones" -> 1; "hash" -> 1 + (splitmix64(key ^ e) % 127).
    """
    nv = 1 << scale
    total = edge_factor * nv
    if num_edges is None:
        num_edges = total - first_edge
    e = np.arange(first_edge, first_edge + num_edges, dtype=np.uint64)
    key = splitmix64(np.uint64(seed))
    src = np.zeros(num_edges, np.int64)
    dst = np.zeros(num_edges, np.int64)
    h = None
    for lvl in range(scale):
        if (lvl & 1) == 0:
            with np.errstate(over="ignore"):
                h = splitmix64(key + e * np.uint64(32) + np.uint64(lvl >> 1))
            r = h & np.uint64(0xFFFFFFFF)
        else:
            r = h >> np.uint64(32)
        sb = r >= np.uint64(RMAT_T_AB)
        db = ((r >= np.uint64(RMAT_T_A)) & (r < np.uint64(RMAT_T_AB))) | (r >= np.uint64(RMAT_T_ABC))
        sh = scale - 1 - lvl
        src |= sb.astype(np.int64) << sh
        dst |= db.astype(np.int64) << sh
    if weights == "ones":
        val = np.ones(num_edges, np.int32)
    else:
        val = (1 + (splitmix64(key ^ e) % np.uint64(127))).astype(np.int32)
    return nv, (src + 1).astype(np.int32), (dst + 1).astype(np.int32), val
