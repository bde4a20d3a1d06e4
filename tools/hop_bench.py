#!/usr/bin/env python3
"""A/B micro-benchmark of hop_kernel variants on one time step of the cfg5 workload (interleaved rounds in ONE
process, hipEvent timing on the launch stream).  Developer tool; not part of the product path.

    python tools/hop_bench.py [--variants 0,1,2] [--rounds 5] [--labeling random] [--n N --nnz NNZ] [--long 256]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,1,2,3")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--labeling", default="random")
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--nnz", type=int, default=160_000_000)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--graph", default="rmat")
    ap.add_argument("--row-thresh", type=int, default=None)
    ap.add_argument("--seg-len", type=int, default=None)
    ap.add_argument("--seg-key", default="first")
    ap.add_argument("--only", default="all", choices=["all", "long", "short"], help="keep only the rows above / up to the row threshold (path analysis)")
    ap.add_argument("--split", type=int, default=1, help="process the C channels in this many column slices")
    ap.add_argument("--sweep", default="1", help="comma list of 0/1: long rows on the sweep schedule (1) or as column-ordered segments (0)")
    ap.add_argument("--sweep-loads", default="8", help="comma list: row loads in flight per lane of the sweep kernel (4, 8, 16)")
    ap.add_argument("--sweep-panel-kb", type=int, default=None)
    ap.add_argument("--sweep-hot-panels", type=int, default=None)
    ap.add_argument("--sweep-barriers", type=int, default=None)
    ap.add_argument("--sweep-thresh", type=int, default=None, help="rows above this many entries go on the sweep schedule (ROW_THRESH)")
    args = ap.parse_args()
    from tools import synth
    from tgcn_amd import _lib, graph, functional as F
    if args.row_thresh:
        graph.ROW_THRESH = args.row_thresh
    if args.seg_len:
        graph.SEG_LEN = args.seg_len
    graph.SEG_KEY = args.seg_key
    dev = torch.device("cuda:0")
    if args.graph == "mesh":
        args.n, row, col, val = synth.sheet_mesh(300, device=dev)
    else:
        _, row, col, val = synth.rmat(args.n, args.nnz, labeling=args.labeling, device=dev)
    if args.only != "all":
        deg = torch.bincount(row, minlength=args.n)
        thr = args.row_thresh or graph.ROW_THRESH
        keep = (deg[row] > thr) if args.only == "long" else (deg[row] <= thr)
        row, col, val = row[keep], col[keep], val[keep]
        print("kept %d entries (%s rows)" % (row.numel(), args.only), flush=True)
    if args.sweep_barriers is not None:
        graph.SWEEP_BARRIER_PANELS = args.sweep_barriers
    if args.sweep_hot_panels is not None:
        graph.SWEEP_HOT_PANELS = args.sweep_hot_panels
    if args.sweep_thresh:
        graph.ROW_THRESH = args.sweep_thresh
    if args.sweep_panel_kb:
        graph.SWEEP_PANEL_BYTES = args.sweep_panel_kb << 10
    ops = {}
    for sw in [int(v) for v in args.sweep.split(",")]:
        graph.SWEEP = bool(sw)
        import time
        t0 = time.time()
        op = ops[sw] = graph.GraphOperand.from_coo(args.n, row, col, val, dev)
        s = op.schedule_for(args.C // args.split)
        torch.cuda.synchronize()
        print("sweep=%d: n=%d nnz=%d blocks=%d segments=%d long rows=%d huge=%d partial slots=%d (T=%d S=%d) built in %.1f s" % (sw, op.n, op.nnz, s.nblk, s.nseg, s.nlong, s.nhuge, s.npartial, s.row_thresh, s.seg_len, time.time() - t0), flush=True)
        if s.sweep is not None:
            w = s.sweep
            print("   sweep schedule: %d rows, %d entries, %d rounds x %d workgroups, panel %d rows" % (w.n_rows, w.n_entries, w.rounds, w.nwg, w.panel_rows), flush=True)
    del row, col, val
    x = torch.randn(1, op.n, args.C, device=dev)
    y = torch.empty_like(x)
    ref = None
    variants = [(int(v), sw, sl) for v in args.variants.split(",") for sw in ops for sl in ([int(u) for u in args.sweep_loads.split(",")] if sw else [8])]
    times = {v: [] for v in variants}
    fix = {v: [] for v in variants}
    sweep_ms = {v: [] for v in variants}
    L = _lib.lib()
    for r in range(args.rounds + 1):
        for v in variants:
            _lib.check(L.tgcn_set_tuning(b"hop_variant", v[0]))
            _lib.check(L.tgcn_set_tuning(b"sweep_loads", v[2]))
            op = ops[v[1]]
            _lib.profile_start(16)
            cs = args.C // args.split
            for sp in range(args.split):
                F.csr_hop(op, x[:, :, sp * cs:(sp + 1) * cs], out=y[:, :, sp * cs:(sp + 1) * cs])
            prof = _lib.profile_stop(16)
            if r == 0:
                if ref is None:
                    ref = y.clone()
                else:
                    err = float((ref - y).abs().max() / ref.abs().max())
                    assert err <= 1e-5, "variant %s changed the result: %g" % (v, err)
                continue
            times[v].append(sum(ms for k, ms in prof if k in (0, 7)))
            sweep_ms[v].append(sum(ms for k, ms in prof if k == 7))
            fix[v].append(sum(ms for k, ms in prof if k == 1))
    _lib.check(L.tgcn_set_tuning(b"hop_variant", 0))
    _lib.check(L.tgcn_set_tuning(b"sweep_loads", 8))
    alg = (8 * op.nnz + 4 * (op.n + 1)) / (16 if args.graph == 'rmat' else 1) + 8 * op.n * args.C
    for v in variants:
        t = np.array(times[v])
        print("hop_variant %d sweep %d loads %d: median %.3f ms  min %.3f ms (sweep kernel %.3f ms)  -> %.0f GB/s algorithmic (cfg5 accounting) = %.3f of 8 TB/s; fixup %.3f ms" % (v[0], v[1], v[2], np.median(t), t.min(), np.median(sweep_ms[v]), alg / np.median(t) / 1e6, alg / np.median(t) / 1e6 / 8000, np.median(fix[v])))


if __name__ == "__main__":
    main()
