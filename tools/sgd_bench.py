#!/usr/bin/env python3
"""SGD / collaborative filtering timing (BASELINE config 5 shape: K=128 fp32 latent vectors).

Synthetic bipartite ratings: users 1..U, items U+1..U+I, `per_user` uniform-random items per
user, ratings uniform 1..5.  One iteration = ALL_EDGES multiply (both directions) + apply.
Reports time per iteration, edge visits/s and achieved GB/s against
  B_alg = 2E*(4+4) + V*K*4*4     and the gather-inclusive 2E*K*4 figure."""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=2_000_000)
    ap.add_argument("--items", type=int, default=200_000)
    ap.add_argument("--per-user", type=int, default=100)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--generic", action="store_true", help="force the generic engine instead of the dedicated kernels")
    ap.add_argument("--lib-option", action="append", default=[], metavar="KEY=VALUE", help="gm_set_option(KEY, VALUE), e.g. sgd_mfma=1")
    ap.add_argument("--compare-mfma", action="store_true", help="also run one iteration with sgd_mfma=1 from the same state and report the largest relative difference")
    args = ap.parse_args()
    from graphmat_amd import api, _lib
    L = _lib.lib()
    for kv in args.lib_option:
        k, v = kv.split("=")
        _lib.check(L.gm_set_option(k.encode(), int(v)))
    if args.generic:
        L.gm_set_option(b"force_ordered", 1)
    dev = torch.device("cuda", 0)
    U, I, K = args.users, args.items, 128
    nv = U + I
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    src = torch.arange(1, U + 1, dtype=torch.int32, device=dev).repeat_interleave(args.per_user)
    dst = (U + 1 + torch.randint(0, I, (src.numel(),), generator=gen, device=dev)).to(torch.int32)
    val = torch.randint(1, 6, (src.numel(),), generator=gen, device=dev).to(torch.int32)
    E = src.numel()
    g = api.Graph(nv, src, dst, val, keep_values=True)
    lat = torch.rand((g.rows, K + 1), generator=gen, device=dev, dtype=torch.float32)
    it = C.c_int(0)
    if args.compare_mfma:
        a, b = lat.clone(), lat.clone()
        _lib.check(L.gm_set_option(b"sgd_mfma", 0))
        _lib.check(L.gm_run_sgd(g.h, a.data_ptr(), K, 4, 0.001, 1e-5, 1, C.byref(it), None))
        _lib.check(L.gm_set_option(b"sgd_mfma", 1))
        _lib.check(L.gm_run_sgd(g.h, b.data_ptr(), K, 4, 0.001, 1e-5, 1, C.byref(it), None))
        torch.cuda.synchronize()
        da = (a[:, :K] - b[:, :K]).abs()
        big = a[:, :K].abs() > 1e-3  # (relative differences of values that cancel to ~0 say nothing)
        rel = (da[big] / a[:, :K].abs()[big]).max().item()
        print("SGD one iteration, matrix-core dot products against the vector form: largest relative difference %.3e over the %d values above 1e-3 in "
              "magnitude, largest absolute difference %.3e, %d of %d values differ in some bit"
              % (rel, int(big.sum()), da.max().item(), int((a[:, :K] != b[:, :K]).sum()), a[:, :K].numel()), flush=True)
        for kv in args.lib_option:
            k, v = kv.split("=")
            _lib.check(L.gm_set_option(k.encode(), int(v)))
        if not any(kv.startswith("sgd_mfma=") for kv in args.lib_option):
            _lib.check(L.gm_set_option(b"sgd_mfma", 0))
        del a, b
    _lib.check(L.gm_run_sgd(g.h, lat.data_ptr(), K, 4, 0.001, 1e-5, 1, C.byref(it), None))  # warm
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(L.gm_run_sgd(g.h, lat.data_ptr(), K, 4, 0.001, 1e-5, args.iters, C.byref(it), None))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.iters
    balg = 2 * E * 8 + nv * K * 4 * 4
    bgather = 2 * E * K * 4
    mode = ("generic" if args.generic else "dedicated") + (" + " + " ".join(args.lib_option) if args.lib_option else "")
    print("SGD K=128 fp32 %s: users=%d items=%d ratings=%d: %.2f ms/iteration, %.2f G edge-visits/s, "
          "algorithmic %.1f GB -> %.0f GB/s (%.1f%% of 8 TB/s); gather-inclusive %.1f GB -> %.0f GB/s; %.2f TFLOP/s of 2E*4K+3KV"
          % (mode, U, I, E, dt * 1e3, 2 * E / dt / 1e9, balg / 1e9, balg / dt / 1e9,
             100 * balg / dt / 8e12, bgather / 1e9, bgather / dt / 1e9, (2 * E * 4 * K + 3 * K * nv) / dt / 1e12), flush=True)

if __name__ == "__main__":
    main()
