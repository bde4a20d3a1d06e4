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
    ap.add_argument("--fold-cols", type=int, default=None, help="replace every column c by c %% H spread over the table: the same row structure with all gathers inside an H-row hot set (upper bound of any locality scheme)")
    ap.add_argument("--block-cost", type=int, default=None, help="graph.BLOCK_ROWS_MAX: entry-equivalents per lane group and row block (default 256)")
    ap.add_argument("--remap", default="1", help="comma list of hop_xcd_remap values, each crossed with the others")
    ap.add_argument("--seg-modes", default="1", help="comma list of tgcn_csr_sched.seg_mode values (0 lane-group segments, 1 wave segments)")
    ap.add_argument("--seg-remaps", default="0", help="comma list of hop_seg_remap values")
    ap.add_argument("--mix", default="0", help="comma list of hop_mix values (row blocks dealt among the segment blocks)")
    ap.add_argument("--compact", action="store_true", help="time the hop on the compacted operand (graph.CompactPlan.rest: only rows with entries)")
    ap.add_argument("--drop-core", type=int, default=None, help="remove the entries whose row AND column are among the H vertices of largest degree (what a hub-core kernel would take over)")
    ap.add_argument("--no-wave-rows", action="store_true", help="graph.WAVE_ROWS = False: medium rows as lane-group segments too")
    ap.add_argument("--lds-pads", default="0", help="comma list of hop_lds_pad values (KB): occupancy limiter, each crossed with --variants")
    ap.add_argument("--cold-last", type=int, default=None, help="reorder the entries inside every row: entries whose column is among the RANK most referenced "
                    "columns first (in column order), the others behind them (in column order) -- a gather instruction then carries lines of one latency class")
    ap.add_argument("--xcd-classes", default=None, choices=["low", "shift", "hash"], help="XCD-specialised column classes WITHOUT extra segments: the entries of every row of more "
                    "than 128 entries are regrouped by class(col) (then column), and the lane-group segments of class c are dealt to the workgroups that run on XCD c (block id %% 8), "
                    "so that the eight 4-MB L2s hold disjoint column sets.  class = col %% 8 (low: round 3's experiment, fixes address bits 8-10), (col >> 4) %% 8 (shift) or a multiplicative hash")
    ap.add_argument("--reorder", default=None, help="GraphOperand.reordered(KIND) before anything else (degree | degree_sorted | rcm)")
    ap.add_argument("--cold-nt", action="store_true", help="with --cold-last: variant 5 runs on a copy of the entries whose COLD columns carry bit 31 "
                    "(hop_kernel gathers them with the non-temporal hint); every other variant runs on the plain entries")
    args = ap.parse_args()
    assert not args.cold_nt or args.cold_last is not None, "--cold-nt needs --cold-last RANK"
    from tools import synth
    from tgcn_amd import _lib, graph, functional as F
    if args.row_thresh:
        graph.ROW_THRESH = args.row_thresh
    if args.seg_len:
        graph.SEG_LEN = args.seg_len
    graph.SEG_KEY = args.seg_key
    if args.block_cost:
        graph.BLOCK_ROWS_MAX = args.block_cost
    graph.WAVE_ROWS = not args.no_wave_rows
    dev = torch.device("cuda:0")
    if args.graph == "mesh":
        args.n, row, col, val = synth.sheet_mesh(300, device=dev)
    elif args.graph == "banded":
        _, row, col, val = synth.banded(args.n, args.nnz, device=dev)
    else:
        _, row, col, val = synth.rmat(args.n, args.nnz, labeling=args.labeling, device=dev)
    if args.fold_cols:
        col = (col % args.fold_cols) * (args.n // args.fold_cols)
    if args.drop_core:
        deg = torch.bincount(row, minlength=args.n)
        hub = torch.zeros(args.n, dtype=torch.bool, device=dev)
        hub[torch.argsort(deg, descending=True)[: args.drop_core]] = True
        keep = ~(hub[row] & hub[col])
        print("hub core %d x %d holds %d of %d entries" % (args.drop_core, args.drop_core, int((~keep).sum()), row.numel()), flush=True)
        row, col, val = row[keep], col[keep], val[keep]
    if args.only != "all":
        deg = torch.bincount(row, minlength=args.n)
        thr = args.row_thresh or graph.ROW_THRESH
        keep = (deg[row] > thr) if args.only == "long" else (deg[row] <= thr)
        row, col, val = row[keep], col[keep], val[keep]
        print("kept %d entries (%s rows)" % (row.numel(), args.only), flush=True)
    op = graph.GraphOperand.from_coo(args.n, row, col, val, dev)
    del row, col, val
    if args.reorder:
        op = op.reordered(args.reorder)
        print("reordered(%r)" % args.reorder, flush=True)
    if args.compact:
        plan = op.compact_plan()
        print("compact: %d rows with entries, %d empty" % (plan.n_c, plan.n_empty), flush=True)
        op = plan.rest
    if args.cold_last is not None:
        # popularity of a column = stored entries that point at it; stable sort of the entries by (row, is_cold) keeps the column order inside both classes
        cols = op.edges[: op.nnz, 0].long()
        pop = torch.bincount(cols, minlength=op.n_cols)
        rank = torch.empty_like(pop)
        rank[torch.argsort(pop, descending=True, stable=True)] = torch.arange(op.n_cols, device=dev)
        cold = (rank[cols] >= args.cold_last)
        counts = (op.rowptr[1:] - op.rowptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(op.n, device=dev), counts)
        order = torch.argsort(rows * 2 + cold.long(), stable=True)
        print("cold-last: %d of %d entries point at a column outside the %d most referenced" % (int(cold.sum()), op.nnz, args.cold_last), flush=True)
        edges2 = op.edges[: op.nnz][order].contiguous()
        cold_sorted = cold[order]
        del cols, pop, rank, cold, rows, order
        op = graph.GraphOperand._from_packed(op.n, op.rowptr, edges2, op.nnz, n_cols=op.n_cols)
        if args.cold_nt:
            flagged = edges2.clone()
            flagged[cold_sorted, 0] |= -2147483648          # bit 31 of the column: "cold" (kNtColdGather in csrc/hop.h masks it off)
        del cold_sorted
    def col_class(c):
        if args.xcd_classes == "low":
            return c % 8
        if args.xcd_classes == "shift":
            return (c >> 4) % 8
        return ((c * 2654435761) >> 13) % 8
    op_plain = op
    if args.xcd_classes:
        cols = op.edges[: op.nnz, 0].long()
        counts = (op.rowptr[1:] - op.rowptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(op.n, device=dev), counts)
        is_long = (counts > 128)[rows]
        # the class order inside a row starts at class (row % 8): the segment that straddles two classes goes to the class of its first entry, and
        # without the rotation class 0 would collect one extra segment per row (17 % more work for XCD 0)
        key = rows * 8 + torch.where(is_long, (col_class(cols) - rows) % 8, torch.zeros_like(cols))
        order = torch.argsort(key, stable=True)             # entries are sorted by column inside a row: stable keeps that inside every class
        edges2 = op.edges[: op.nnz][order].contiguous()
        print("xcd classes (%s): %d of %d entries sit in rows of more than 128 entries" % (args.xcd_classes, int(is_long.sum()), op.nnz), flush=True)
        del cols, rows, is_long, key, order
        op = graph.GraphOperand._from_packed(op.n, op.rowptr, edges2, op.nnz, n_cols=op.n_cols)
    lanes = _lib.lib().tgcn_hop_lanes_per_row(args.C // args.split, 1)
    scheds = {m: graph.Schedule(op.rowptr, op.n, lanes, edges=op.edges, seg_mode=m, n_cols=op.n_cols) for m in sorted(set(int(m) for m in args.seg_modes.split(",")))}
    for m, sm in scheds.items():
        print("seg_mode %d: blocks=%d segments=%d (whole-row wave segments %d) long rows=%d huge=%d partial slots=%d seg_len=%d" % (m, sm.nblk, sm.nseg, getattr(sm, "nwseg", 0), sm.nlong, sm.nhuge, sm.npartial, sm.seg_len), flush=True)
    if args.xcd_classes:
        # deal the lane-group segments (behind the whole-row wave segments) to the XCDs by the class of their first column: segment block j runs as
        # workgroup nblk + nwblk + j, i.e. on XCD (nblk + nwblk + j) % 8; class lists keep the builder's order (by first column) and are padded
        # with empty segments that write zeros into one dummy partial slot
        assert set(scheds) == {0} and lanes == 16, "--xcd-classes: --seg-modes 0, 16-lane groups"
        sm = scheds[0]
        nw, ns, gpb = sm.nwseg, sm.nseg, 256 // lanes
        seg_row, e0, e1, slot = [t[nw:ns].long() for t in (sm.seg_row, sm.seg_e0, sm.seg_e1, sm.seg_slot)]
        cls = col_class(op.edges[e0, 0].long())
        base = (sm.nblk + (nw + 3) // 4) % 8
        cnt = torch.bincount(cls, minlength=8)
        T = int(((cnt + gpb - 1) // gpb).max().item())
        total = T * 8 * gpb
        new = [torch.zeros(total, dtype=torch.int64, device=dev) for _ in range(3)] + [torch.full((total,), sm.npartial, dtype=torch.int64, device=dev)]
        for c in range(8):
            idx = (cls == c).nonzero().flatten()
            k = torch.arange(idx.numel(), device=dev)
            pos = ((k // gpb) * 8 + (c - base) % 8) * gpb + k % gpb
            for dst, src in zip(new, (seg_row, e0, e1, slot)):
                dst[pos] = src[idx]
        cat = lambda a, b: torch.cat([a[:nw], b.to(torch.int32)]).contiguous()
        sm.seg_row, sm.seg_e0, sm.seg_e1, sm.seg_slot = cat(sm.seg_row, new[0]), cat(sm.seg_e0, new[1]), cat(sm.seg_e1, new[2]), cat(sm.seg_slot, new[3])
        sm.nseg, sm.npartial = nw + total, sm.npartial + 1
        sm.struct = _lib.SchedStruct(sm.lanes_per_row, sm.row_thresh, sm.nblk, sm.nseg, sm.nlong, sm.nhuge, sm.npartial, sm.seg_mode, sm.row_mix, sm.nwseg,
                                     sm.blk_row.data_ptr(), sm.seg_row.data_ptr(), sm.seg_e0.data_ptr(), sm.seg_e1.data_ptr(), sm.seg_slot.data_ptr(),
                                     sm.long_row.data_ptr(), sm.long_slot.data_ptr())
        op._sched[lanes] = sm
        print("xcd classes: %d lane-group segments per class %s -> %d blocks per class, %d padding segments" % (ns - nw, cnt.tolist(), T, total - (ns - nw)), flush=True)
    s = op.schedule_for(args.C // args.split)
    print("n=%d nnz=%d blocks=%d segments=%d long rows=%d huge=%d partial slots=%d (T=%d S=%d)" % (op.n, op.nnz, s.nblk, s.nseg, s.nlong, s.nhuge, s.npartial, s.row_thresh, s.seg_len), flush=True)
    x = torch.randn(1, op.n_cols, args.C, device=dev)
    y = torch.empty(1, op.n, args.C, device=dev)
    ref = None
    variants = [(int(v), int(pd), int(rm), int(sm), int(sr), int(mx)) for v in args.variants.split(",") for pd in args.lds_pads.split(",") for rm in args.remap.split(",")
                for sm in args.seg_modes.split(",") for sr in args.seg_remaps.split(",") for mx in args.mix.split(",")]
    times = {v: [] for v in variants}
    fix = {v: [] for v in variants}
    L = _lib.lib()
    for r in range(args.rounds + 1):
        for v in variants:
            _lib.check(L.tgcn_set_tuning(b"hop_variant", v[0]))
            _lib.check(L.tgcn_set_tuning(b"hop_lds_pad", v[1] * 1024))
            _lib.check(L.tgcn_set_tuning(b"hop_xcd_remap", v[2]))
            _lib.check(L.tgcn_set_tuning(b"hop_seg_remap", v[4]))
            _lib.check(L.tgcn_set_tuning(b"hop_mix", v[5]))
            op._sched[lanes] = scheds[v[3]]
            if args.cold_nt:      # ONLY variant 5 understands flagged columns: every other kernel would read out of bounds
                op.struct.edges = flagged.data_ptr() if v[0] == 5 else op.edges.data_ptr()
            _lib.profile_start(16)
            cs = args.C // args.split
            for sp in range(args.split):
                F.csr_hop(op, x[:, :, sp * cs:(sp + 1) * cs], out=y[:, :, sp * cs:(sp + 1) * cs])
            prof = _lib.profile_stop(16)
            if r == 0:
                if ref is None:
                    ref = y.clone()
                else:
                    assert torch.allclose(ref, y, rtol=1e-4, atol=1e-5), "variant %s changed the result" % (v,)
                continue
            times[v].append(sum(ms for k, ms in prof if k == 0))
            fix[v].append(sum(ms for k, ms in prof if k == 1))
    _lib.check(L.tgcn_set_tuning(b"hop_variant", 0))
    _lib.check(L.tgcn_set_tuning(b"hop_lds_pad", 0))
    _lib.check(L.tgcn_set_tuning(b"hop_xcd_remap", 1))
    _lib.check(L.tgcn_set_tuning(b"hop_seg_remap", 0))
    _lib.check(L.tgcn_set_tuning(b"hop_mix", 0))
    if args.xcd_classes:
        yp = F.csr_hop(op_plain, x)
        print("class-regrouped schedule vs plain: max rel err %.2e" % float((y - yp).abs().max() / yp.abs().max()), flush=True)
    alg = (8 * op.nnz + 4 * (op.n + 1)) / (16 if args.graph in ('rmat', 'banded') else 1) + 8 * op.n * args.C
    for v in variants:
        t = np.array(times[v])
        print("variant %s: median %.3f ms  min %.3f ms   -> %.0f GB/s algorithmic (cfg5 accounting); fixup %.3f ms" % (v, np.median(t), t.min(), alg / np.median(t) / 1e6, np.median(fix[v])))


if __name__ == "__main__":
    main()
