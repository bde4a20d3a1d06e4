#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes of bench.py into profiles/traffic_<workload>_<labeling>.json (developer tool).

    python tools/traffic_json.py <out.json> <pmc_dir> [<pmc_dir> ...]

Per kernel: mean FETCH_SIZE / WRITE_SIZE (KB) / TCC_HIT_sum / TCC_MISS_sum per dispatch.  HBM-side bytes of a kernel
= (2*FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KB and gfx950 tallies 128-byte read requests at 64 bytes
(MI355X_MICROARCH.md, HBM).  bench.py reports hop_kernel's figure as roofline.traffic if the recorded source hash still matches the kernel sources."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def short(name):
    for k in ("hop_fixup_kernel", "hop_kernel", "project_x3_stream_kernel", "project_x3v2_kernel", "project_x3_kernel", "project_resident_kernel",
              "project_kernel", "project_narrow_kernel"):
        if k in name:
            return k
    return None


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        e = dict(dispatches=max(len(v) for v in cs.values()))
        if "FETCH_SIZE" in m:
            e["fetch_size_kb"] = m["FETCH_SIZE"]
        if "WRITE_SIZE" in m:
            e["write_size_kb"] = m["WRITE_SIZE"]
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["hbm_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
            e["tcc_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        kernels[k] = e
    from tgcn_amd import _lib
    hop = kernels.get("hop_kernel", {}).get("hbm_bytes_per_launch", 0)
    json.dump(dict(hbm_bytes_per_hop_launch=hop, kernels=kernels, source_hash=_lib.binary_hash(),
                   source="rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, one pass each) of `python3 bench.py --steps 2 --warmup 1 --no-cpu`; "
                          "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B)"), open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
