#!/usr/bin/env python3
"""Condense rocprofv3 output dirs (gpurun_out/prof_{kt,fetch,write,tcc}) into profiles/<tag>_kernel_stats.csv,
profiles/<tag>_pmc.json and profiles/traffic_<workload>_<labeling>.json (read by bench.py).

    python tools/summarize_prof.py <tag> [--workload cfg5 --labeling random] [--src gpurun_out]
"""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def short(k):
    for name in ("hop_fixup_kernel", "hop_kernel", "project_kernel", "relayout_kernel", "pool_max"):
        if name in k:
            return name
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--workload", default="cfg5")
    ap.add_argument("--labeling", default="random")
    ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out"))
    a = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    ks = sorted(glob.glob(os.path.join(a.src, "prof_kt", "*", "*_kernel_stats.csv")) + glob.glob(os.path.join(a.src, "prof_kt", "*_kernel_stats.csv")),
                key=os.path.getmtime, reverse=True)
    if ks:
        rows = list(csv.reader(open(ks[0])))
        keep = [rows[0]] + [r for r in rows[1:] if short(r[0]) or float(r[4]) >= 0.5]
        csv.writer(open(os.path.join(prof, a.tag + "_kernel_stats.csv"), "w")).writerows(keep)
    res = {}
    for d in ("prof_fetch", "prof_write", "prof_tcc"):
        for f in sorted(glob.glob(os.path.join(a.src, d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(a.src, d, "*_counter_collection.csv")),
                        key=os.path.getmtime)[-1:]:
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if k:
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k, v in acc.items():
                for c, vals in v.items():
                    res.setdefault(k, {})[c] = dict(mean=sum(vals) / len(vals), n=len(vals))
    if res:
        json.dump(res, open(os.path.join(prof, a.tag + "_pmc.json"), "w"), indent=1)
        h = res.get("hop_kernel", {})
        if "FETCH_SIZE" in h and "WRITE_SIZE" in h:
            traffic = int((2 * h["FETCH_SIZE"]["mean"] + h["WRITE_SIZE"]["mean"]) * 1024)
            out = dict(hbm_bytes_per_hop_launch=traffic,
                       source="profiles/%s_pmc.json: (2*FETCH_SIZE + WRITE_SIZE)*1024 -- counters are KB, FETCH_SIZE doubled per "
                              "MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B), separate --pmc passes" % a.tag,
                       fetch_size_kb=h["FETCH_SIZE"]["mean"], write_size_kb=h["WRITE_SIZE"]["mean"])
            if "TCC_HIT_sum" in h:
                out["tcc_hit_rate"] = h["TCC_HIT_sum"]["mean"] / h["TCC_REQ_sum"]["mean"]
            json.dump(out, open(os.path.join(prof, "traffic_%s_%s.json" % (a.workload, a.labeling)), "w"), indent=1)
            print(json.dumps(out, indent=1))
    for k, v in res.items():
        print(k, {c: "%.4g" % x["mean"] for c, x in v.items()})


if __name__ == "__main__":
    main()
