#!/usr/bin/env python3
"""profiles/<tag>_scale<S>_pmc_{FETCH,WRITE}_SIZE.md  ->  profiles/pmc_traffic.json
(bytes moved between L2 and the fabric per launch of each PageRank kernel = (FETCH_SIZE + WRITE_SIZE) * 1024,
counters per dispatch as printed by tools/prof_summary.py).  usage: pmc_to_json.py <tag> <scale> [<scale> ...]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


ITERATIONS = 6  # tools/final_profiles.sh runs the PMC passes with --steps 5 --warmup 1


def per_iteration(path, counter):
    """counter sum over all dispatches of a PageRank kernel / iterations of the run (a kernel may be launched
    several times per iteration: once per column tile)"""
    out = {}
    for line in open(path):
        m = re.match(r"\| `(k_\w+)<gm::PageRankP.*` \| %s \| \d+ \| ([^|]+) \| [^|]+ \|" % counter, line)
        if m:
            out[m.group(1)] = out.get(m.group(1), 0.0) + float(m.group(2)) / ITERATIONS
    return out


def main():
    tag, scales = sys.argv[1], sys.argv[2:]
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc["_note"] = ("HBM/fabric bytes per PageRank iteration and kernel (all its launches of the iteration) from rocprofv3 PMC passes of `python bench.py --scale S --steps 5 --warmup 1 "
                    "--no-timing` (one counter set per run, kernel-trace only): (FETCH_SIZE + WRITE_SIZE) * 1024. No x2 correction "
                    "is applied: FETCH_SIZE / TCC_MISS = 63.5-64 B per miss, i.e. these are 64-byte random fetches, not wide "
                    "streaming reads (MI355X_MICROARCH.md HBM section). tools/final_profiles.sh + tools/pmc_to_json.py.")
    sys.path.insert(0, ROOT)
    import bench
    doc["kernels_fingerprint"] = bench.kernels_fingerprint()
    doc["col_tiles"] = {}
    for sc in scales:
        f = per_iteration(os.path.join(ROOT, "profiles", "%s_scale%s_pmc_FETCH_SIZE.md" % (tag, sc)), "FETCH_SIZE")
        w = per_iteration(os.path.join(ROOT, "profiles", "%s_scale%s_pmc_WRITE_SIZE.md" % (tag, sc)), "WRITE_SIZE")
        try:  # how many column tiles the profiled run used (its bench line says)
            doc["col_tiles"]["scale%s" % sc] = json.load(open(os.path.join(ROOT, "profiles", "%s_scale%s_bench.json" % (tag, sc))))["config"]["col_tiles"]
        except Exception:
            doc["col_tiles"]["scale%s" % sc] = None
        doc["scale%s" % sc] = {k + "_bytes_per_iteration": int((f[k] + w.get(k, 0.0)) * 1024) for k in sorted(f)}
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
