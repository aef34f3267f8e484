#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel (per dispatch).
usage: pmc_summary.py [--json OUT --workload W --source-hash H] DIR [DIR ...]

With --json the HBM traffic per launch of every kernel goes into OUT (the file bench.py looks up as
profiles/rNN/kernel_traffic.json): FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies the 128-B requests of wide
coalesced reads at 64 B: MI355X_MICROARCH.md, HBM) + WRITE_SIZE KiB x 1024, counters from separate passes."""
import argparse
import collections
import csv
import glob
import json

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--json")
ap.add_argument("--workload")
ap.add_argument("--source-hash")
ap.add_argument("--command", default="")
args = ap.parse_args()

means = {}
for d in args.dirs:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per_dispatch.items():
            name = k.replace("(anonymous namespace)::", "").split("(")[0]
            if name.startswith("void "):
                name = name[5:]
            agg[(name, c)].append(v)
        for (k, c), v in sorted(agg.items()):
            if v and max(v) > 0 and not k.startswith("__amd"):
                print(f"{k[:28]:28s} {c:38s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
                means[(k, c)] = (sum(v) / len(v), len(v))

if args.json:
    kernels = sorted({k for k, _ in means})
    traffic, detail = {}, {}
    for k in kernels:
        if (k, "FETCH_SIZE") not in means:
            continue
        fetch = means[(k, "FETCH_SIZE")][0]
        write = means.get((k, "WRITE_SIZE"), (0.0, 0))[0]
        short = k.split("<")[0]
        traffic[short] = fetch * 1024 * 2 + write * 1024
        detail[short] = {"kernel": k, "launches": means[(k, "FETCH_SIZE")][1], "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                         **{c: means[(k, c)][0] for kk, c in means if kk == k and c not in ("FETCH_SIZE", "WRITE_SIZE")}}
    json.dump({"workload": args.workload, "source_hash": args.source_hash, "command": args.command,
               "correction": "traffic = 2 x FETCH_SIZE KiB x 1024 + WRITE_SIZE KiB x 1024 (gfx950: wide coalesced reads are "
                             "tallied at half their bytes; writes are exact) -- MI355X_MICROARCH.md, HBM",
               "traffic_bytes_per_launch": traffic, "counters_per_launch": detail}, open(args.json, "w"), indent=1)
