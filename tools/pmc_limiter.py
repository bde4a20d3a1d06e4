#!/usr/bin/env python3
"""Name what hop_kernel waits for: rocprofv3 --pmc passes of tools/hop_bench.py (one time step of cfg5 on the compacted operand)
for the request path between the CUs and the fabric -- TCP (vector L1) stalls and miss latency, TA/TD stalls, UTCL1 translation,
TCC (L2) requests / hits / fabric reads and their stalls, SQ wave states.  Developer tool.

    python3 tools/pmc_limiter.py <tag> [--labelings random,degree] [--extra "<hop_bench args>"] [--only-blocks TCP,TA]

The counters that exist are taken from `rocprofv3 -L` on the box (gfx950 names differ between ROCm releases); a pass that does
not fit the block's slots is split in two and run again.  This process never touches the GPU itself: every pass is a child
`rocprofv3 ... -- python3 tools/hop_bench.py ...` (the program itself after `--`).  Output: gpurun_out/<tag>/limiter_<labeling>.json
(mean per hop_kernel dispatch and derived ratios); copy what is to be judged into profiles/."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# block -> (slots to try per pass, candidate counters in order of interest)
CANDIDATES = collections.OrderedDict([
    ("TCP", (4, ["TCP_PENDING_STALL_CYCLES", "TCP_READ_TAGCONFLICT_STALL_CYCLES", "TCP_TCC_READ_REQ_LATENCY", "TCP_TCC_READ_REQ",
                 "TCP_UTCL1_REQUEST", "TCP_UTCL1_TRANSLATION_MISS", "TCP_GATE_EN1", "TCP_GATE_EN2",
                 "TCP_TCP_LATENCY", "TCP_TA_TCP_STATE_READ", "TCP_TCR_TCP_STALL_CYCLES", "TCP_TA_DATA_STALL_CYCLES",
                 "TCP_TD_TCP_STALL_CYCLES", "TCP_TOTAL_CACHE_ACCESSES", "TCP_UTCL1_TRANSLATION_HIT", "TCP_UTCL1_PERMISSION_MISS",
                 "TCP_TOTAL_ACCESSES", "TCP_TOTAL_READ", "TCP_VOLATILE", "TCP_TCC_NC_READ_REQ"])),
    ("TA", (2, ["TA_ADDR_STALLED_BY_TC_CYCLES", "TA_ADDR_STALLED_BY_TD_CYCLES", "TA_DATA_STALLED_BY_TC_CYCLES", "TA_TA_BUSY",
                "TA_BUSY", "TA_FLAT_READ_WAVEFRONTS", "TA_BUFFER_READ_WAVEFRONTS", "TA_TOTAL_WAVEFRONTS"])),
    ("TD", (2, ["TD_TC_STALL", "TD_TD_BUSY", "TD_LOAD_WAVEFRONT", "TD_SPI_STALL"])),
    ("TCC", (4, ["TCC_REQ", "TCC_HIT", "TCC_MISS", "TCC_EA0_RDREQ", "TCC_EA0_RDREQ_LEVEL", "TCC_TAG_STALL", "TCC_EA0_RDREQ_32B", "TCC_EA0_RDREQ_DRAM",
                 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL", "TCC_EA0_RDREQ_GMI_CREDIT_STALL", "TCC_BUSY", "TCC_CYCLE",
                 "TCC_EA0_RDREQ_IO_CREDIT_STALL", "TCC_READ", "TCC_EA0_RD_UNCACHED_32B", "TCC_NC_REQ", "TCC_UC_REQ", "TCC_CC_REQ", "TCC_RW_REQ",
                 "TCC_PROBE", "TCC_STREAMING_REQ", "TCC_EA0_WRREQ", "TCC_EA0_WRREQ_STALL", "TCC_TOO_MANY_EA_WRREQS_STALL",
                 "TCC_EA0_ATOMIC", "TCC_NORMAL_WRITEBACK", "TCC_NORMAL_EVICT", "TCC_BUBBLE", "TCC_EA0_WRREQ_LEVEL", "TCC_EA0_ATOMIC_LEVEL"])),
    ("SQ", (8, ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM",
                "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR", "SQ_INST_CYCLES_VMEM_RD",
                "SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_FLAT",
                "SQ_BUSY_CU_CYCLES", "SQ_CYCLES"])),
    ("GRBM", (2, ["GRBM_GUI_ACTIVE", "GRBM_COUNT"])),
    # matrix pipe and LDS of the MFMA kernels (projection): --only-blocks SQX,SQ,GRBM
    ("SQX", (8, ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
                 "SQ_LDS_ADDR_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU2", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL", "SQ_LDS_UNALIGNED_STALL",
                 "SQ_INSTS_VALU_MFMA_BF16", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA"])),
])


def available(out_dir):
    path = os.path.join(out_dir, "counters_avail.txt")
    if not os.path.exists(path):
        with open(path, "w") as f:
            subprocess.run(["rocprofv3", "-L"], stdout=f, stderr=subprocess.STDOUT, timeout=300)
    text = open(path, errors="replace").read()
    return set(re.findall(r"\b[A-Z][A-Za-z0-9]*_[A-Za-z0-9_]+\b", text))


def resolve(name, names):
    """The summed form when the list has one (`X_sum`: over the block's instances), else the plain counter."""
    for cand in (name + "_sum", name):
        if cand in names:
            return cand
    return None


SCRIPT = os.path.join(ROOT, "tools", "hop_bench.py")      # the program profiled (--script)
KERNELS = ["hop_fixup_kernel", "hop_kernel"]               # substrings of the kernel names folded (--kernels; first match wins)


def run_pass(counters, out_dir, tag, bench_args, limit):
    d = os.path.join(out_dir, tag)
    cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, SCRIPT] + bench_args
    t0 = time.time()
    with open(d + ".log", "w") as log:
        log.write(" ".join(cmd) + "\n")
        log.flush()
        try:
            rc = subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT, timeout=limit, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp")).returncode
        except subprocess.TimeoutExpired:
            print("pass %s was killed after %d s: stopping (no further GPU work)" % (tag, limit), flush=True)
            sys.exit(3)
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((name for name in KERNELS if name in r["Kernel_Name"]), None)
            if k:
                rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    shutil.rmtree(d, ignore_errors=True)        # raw rocprofv3 output: tens of MB per pass, gpurun merges at most 64 MiB back
    print("pass %s: rc=%d %.0f s, counters %s -> %s" % (tag, rc, time.time() - t0, counters, {k: sorted(v) for k, v in rows.items()}), flush=True)
    return rc, rows


def collect(counters, out_dir, tag, bench_args, limit, acc, failed):
    rc, rows = run_pass(counters, out_dir, tag, bench_args, limit)
    got = set(rows.get(KERNELS[-1], {}))
    if got:
        for k, cs in rows.items():
            for c, v in cs.items():
                acc[k][c] = (sum(v) / len(v), len(v))
    missing = [c for c in counters if c not in got]
    if missing and len(counters) > 1 and not got:
        h = len(counters) // 2
        collect(counters[:h], out_dir, tag + "a", bench_args, limit, acc, failed)
        collect(counters[h:], out_dir, tag + "b", bench_args, limit, acc, failed)
    elif missing:
        failed.extend(missing)


def derived(m):
    d = {}

    def g(*names):
        for n in names:
            for s in (n + "_sum", n):
                if s in m:
                    return m[s][0]
        return None
    lat, req = g("TCP_TCC_READ_REQ_LATENCY"), g("TCP_TCC_READ_REQ")
    if lat and req:
        d["tcp_to_tcc_read_latency_cycles"] = lat / req
    tl, ta = g("TCP_TCP_LATENCY"), g("TCP_TA_TCP_STATE_READ")
    if tl and ta:
        d["tcp_latency_cycles_per_ta_read"] = tl / ta
    for stall in ("TCP_PENDING_STALL_CYCLES", "TCP_READ_TAGCONFLICT_STALL_CYCLES", "TCP_TCR_TCP_STALL_CYCLES", "TCP_TA_DATA_STALL_CYCLES", "TCP_TD_TCP_STALL_CYCLES"):
        s, gate = g(stall), g("TCP_GATE_EN2", "TCP_GATE_EN1")
        if s is not None and gate:
            d[stall.lower() + "_share_of_tcp_active"] = s / gate
    um, ur = g("TCP_UTCL1_TRANSLATION_MISS"), g("TCP_UTCL1_REQUEST")
    if um is not None and ur:
        d["utcl1_miss_rate"] = um / ur
    h, mi = g("TCC_HIT"), g("TCC_MISS")
    if h is not None and mi is not None and h + mi:
        d["tcc_hit_rate"] = h / (h + mi)
    lvl, rd = g("TCC_EA0_RDREQ_LEVEL"), g("TCC_EA0_RDREQ")
    if lvl and rd:
        d["tcc_ea_read_latency_cycles"] = lvl / rd
    wa, wi, ac, wc = g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_ANY"), g("SQ_WAVE_CYCLES")
    if wc:
        for n, v in (("sq_wait_any", wa), ("sq_wait_inst_any", wi), ("sq_active_inst_any", ac)):
            if v is not None:
                d[n + "_share_of_wave_cycles"] = v / wc
    lv, nv = g("SQ_INST_LEVEL_VMEM"), g("SQ_INSTS_VMEM", "SQ_INSTS_VMEM_RD")
    if lv and nv:
        d["sq_vmem_latency_cycles"] = lv / nv
    for stall in ("TA_ADDR_STALLED_BY_TC_CYCLES", "TA_ADDR_STALLED_BY_TD_CYCLES", "TA_DATA_STALLED_BY_TC_CYCLES"):
        s, busy = g(stall), g("TA_TA_BUSY", "TA_BUSY")
        if s is not None and busy:
            d[stall.lower() + "_share_of_ta_busy"] = s / busy
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--labelings", default="random,degree")
    ap.add_argument("--extra", default="", help="further arguments for tools/hop_bench.py")
    ap.add_argument("--only-blocks", default="")
    ap.add_argument("--limit", type=int, default=300, help="seconds per pass")
    ap.add_argument("--max-passes", type=int, default=99, help="stop after this many combined passes per labeling")
    ap.add_argument("--script", default=None, help="program to profile instead of tools/hop_bench.py (e.g. bench.py); --extra then holds ALL its arguments")
    ap.add_argument("--kernels", default=None, help="comma list of kernel-name substrings to fold (first match wins; the LAST one is the kernel of interest)")
    args = ap.parse_args()
    global SCRIPT, KERNELS
    if args.script:
        SCRIPT = os.path.join(ROOT, args.script)
    if args.kernels:
        KERNELS = args.kernels.split(",")
    out_dir = os.path.join(ROOT, "gpurun_out", args.tag)
    os.makedirs(out_dir, exist_ok=True)
    names = available(out_dir)
    print("%d counter names on this box" % len(names), flush=True)
    blocks = [b for b in CANDIDATES if not args.only_blocks or b in args.only_blocks.split(",")]
    for lab in args.labelings.split(","):
        bench_args = (["--compact", "--variants", "0", "--rounds", "2", "--labeling", lab] if not args.script else ["--labeling", lab]) + args.extra.split()
        acc = collections.defaultdict(dict)
        failed, absent = [], []
        groups = {}
        for b in blocks:
            slots, cands = CANDIDATES[b]
            have = []
            for c in cands:
                r = resolve(c, names)
                (have if r else absent).append(r or c)
            groups[b] = [have[i:i + slots] for i in range(0, len(have), slots)]
        # the blocks have their own counter slots: pass i carries group i of every block; a pass that collects nothing
        # is repeated block by block (and a block's group that still fails is halved)
        for i in range(max([len(g) for g in groups.values()] + [0])):
            if i >= args.max_passes:
                break
            parts = [(b, groups[b][i]) for b in blocks if i < len(groups[b])]
            combined = [c for _, g in parts for c in g]
            rc, rows = run_pass(combined, out_dir, "%s_p%d" % (lab, i), bench_args, args.limit)
            got = set(rows.get(KERNELS[-1], {}))
            for k, cs in rows.items():
                for c, v in cs.items():
                    acc[k][c] = (sum(v) / len(v), len(v))
            if not got and len(parts) > 1:
                for b, g in parts:
                    collect(g, out_dir, "%s_p%d_%s" % (lab, i, b.lower()), bench_args, args.limit, acc, failed)
            else:
                failed.extend(c for c in combined if c not in got)
        res = dict(labeling=lab, command="rocprofv3 --pmc <counters> -- python3 %s " % os.path.relpath(SCRIPT, ROOT) + " ".join(bench_args),
                   note="mean per dispatch over the launches of each kernel in the run (summed over the block's instances where the name ends in _sum); "
                        "kernels run serialised under the profiler",
                   counters_absent_on_this_box=absent, counters_that_did_not_collect=failed,
                   derived=derived(acc.get(KERNELS[-1], {})), derived_for=KERNELS[-1])
        for kname in KERNELS:
            res[kname] = {c: v[0] for c, v in sorted(acc.get(kname, {}).items())}
            res[kname + "_dispatches"] = {c: v[1] for c, v in sorted(acc.get(kname, {}).items())}
        path = os.path.join(out_dir, "limiter_%s.json" % lab)
        json.dump(res, open(path, "w"), indent=1)
        print(open(path).read(), flush=True)


if __name__ == "__main__":
    main()
