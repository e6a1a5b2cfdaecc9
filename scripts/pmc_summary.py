#!/usr/bin/env python
"""Summarises the rocprofv3 --pmc passes of scripts/pmc_collect.sh: average FETCH_SIZE / WRITE_SIZE (KB) per kernel and
launch grid.  HBM bytes of a launch = 2*FETCH_SIZE + WRITE_SIZE kilobytes (gfx950: FETCH_SIZE counts 64-byte units where
the tool assumes 32, MI355X_MICROARCH.md HBM section).   python scripts/pmc_summary.py <dir> <tag>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main():
    d, tag = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(d, "%s_%s" % (tag, c), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
                if not name.startswith("k_"):
                    continue
                key = "%s grid=%s" % (name, r["Grid_Size"])
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for key, cs in sorted(acc.items()):
        e = {c: {"n": len(v), "avg_KB": sum(v) / len(v)} for c, v in cs.items()}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_MB_per_launch"] = (2 * e["FETCH_SIZE"]["avg_KB"] + e["WRITE_SIZE"]["avg_KB"]) / 1024.0
            # per launch, in dispatch order (the two passes run the same command: launch i of one is launch i of the other) -- a kernel
            # that is launched with different arguments under one name and grid (K2 at k = 8 and k = 60) is told apart by the caller
            if len(cs["FETCH_SIZE"]) == len(cs["WRITE_SIZE"]):
                e["hbm_MB_launches"] = [round((2 * f_ + w_) / 1024.0, 2) for f_, w_ in zip(cs["FETCH_SIZE"], cs["WRITE_SIZE"])]
        out[key] = e
    # provenance: digests of the kernel sources the counters were collected on (bench.py compares them with the tree it runs
    # in and flags a stale file); the commit hash is added when the file is copied into profiles/ (the GPU box has no .git)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = {}
    for src in ("orth.hip", "spmv.hip", "spmv_tile.hip", "trsv_ml.hip", "gemm.hip"):
        try:
            meta[src.replace(".", "_") + "_digest"] = hashlib.sha256(open(os.path.join(root, "nonlineareigenproblems.jl_amd", "csrc", src), "rb").read()).hexdigest()[:16]
        except OSError:
            pass
    meta["commit"] = os.environ.get("NEP_PROFILE_COMMIT")
    out["_meta"] = meta
    json.dump(out, open(os.path.join(d, "%s_traffic.json" % tag), "w"), indent=1)
    for key, e in out.items():
        if key == "_meta":
            continue
        print("%-60s %s" % (key[:60], {k: (round(v["avg_KB"] / 1024, 2) if isinstance(v, dict) else (round(v, 2) if not isinstance(v, list) else "%d launches" % len(v)))
                                     for k, v in e.items()}))


if __name__ == "__main__":
    main()
