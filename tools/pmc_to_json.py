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
        m = re.match(r"\| `(?:GraphMat::dev::)?(k_\w+)(?:<gm::PageRankP.*)?` \| %s \| \d+ \| ([^|]+) \| [^|]+ \|" % counter, line)
        if m and "<" not in line.split("`")[1] and not m.group(1).startswith("k_giant_"):
            m = None  # (kernels without the program in their name: only the giant rows' k_giant_sums / k_giant_maps belong to the iteration)
        if m:
            out[m.group(1)] = out.get(m.group(1), 0.0) + float(m.group(2)) / ITERATIONS
    return out


def calibration(tag):
    """reported bytes / known bytes of tools/pmc_calibrate.hip's three kernels"""
    known = {"k_cal_stream": (512 << 20) * 4, "k_cal_gather": (256 << 20) * 64, "k_cal_write": (256 << 20) * 4}
    out = {}
    for counter, kernels in (("FETCH_SIZE", ("k_cal_stream", "k_cal_gather")), ("WRITE_SIZE", ("k_cal_write",))):
        path = os.path.join(ROOT, "profiles", "%s_pmc_calibration_%s.md" % (tag, counter))
        if not os.path.exists(path):
            continue
        for line in open(path):
            m = re.match(r"\| `(k_cal_\w+)` \| %s \| (\d+) \| ([^|]+) \| [^|]+ \|" % counter, line)
            if m and m.group(1) in kernels:
                per = float(m.group(3)) / int(m.group(2)) * 1024
                out["%s_reported_over_known_%s" % (counter, {"k_cal_stream": "coalesced_stream", "k_cal_gather": "random_4B_gather_at_64B_per_miss",
                                                              "k_cal_write": "coalesced_write"}[m.group(1)])] = round(per / known[m.group(1)], 4)
    return out


def main():
    tag, scales = sys.argv[1], sys.argv[2:]
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc["_note"] = ("HBM/fabric bytes per PageRank iteration and kernel (all its launches of the iteration) from rocprofv3 PMC passes of `python bench.py --scale S --steps 5 --warmup 1 "
                    "--no-timing` (one counter set per run, kernel-trace only): RAW (FETCH_SIZE + WRITE_SIZE) * 1024 per kernel, plus the calibration of the two "
                    "counters on known access patterns (tools/pmc_calibrate.hip, profiles/<tag>_pmc_calibration_*.md): on this gfx950 rocprofv3 FETCH_SIZE reports HALF "
                    "the bytes of a coalesced stream (128-byte requests tallied at 64 B: MI355X_MICROARCH.md, HBM section) and 64 B per random 4-byte gather that misses; "
                    "WRITE_SIZE reports a coalesced write exactly.  bench.py adds the uncounted half of each kernel's coalesced streams (their bytes are known by "
                    "construction) to the raw figure. tools/final_profiles_r5.sh + tools/pmc_to_json.py.")
    doc["calibration"] = calibration(tag)
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
        doc["scale%s" % sc].update({k + "_fetch_bytes_per_iteration": int(f[k] * 1024) for k in sorted(f)})
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
