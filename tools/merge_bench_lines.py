#!/usr/bin/env python3
"""gpurun_out/bench_lines/*.json (tools/collect_bench_lines.sh) -> profiles/<tag>_bench_lines.json.  Developer tool."""
import glob
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "bench_lines", "*.json"))):
    lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if lines:
        out[os.path.basename(f)[:-5]] = json.loads(lines[-1])
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "bench_lines", "train_*.txt"))):
    out[os.path.basename(f)[:-4]] = [l for l in open(f).read().splitlines() if "forward" in l or "HCP-style" in l]
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_bench_lines.json"), "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, v["value"], v["ms_per_step"], (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"))
    else:
        print(k, *v, sep="\n   ")
