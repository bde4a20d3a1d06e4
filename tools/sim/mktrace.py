#!/usr/bin/env python3
"""Developer tool: build access traces of hop-kernel schedules for tools/sim/l2sim.c (see tools/sim/README.md).

    python tools/sim/mktrace.py <rowptr.npy> <col.npy> <schedule> <out.bin> [key=value ...]
"""
import sys
import numpy as np


def write_trace(path, wg_stream_ptr, stream_ptr, cols, slots, quantum, ways=16, sets=1024, rounds=0):
    hdr = np.array([len(wg_stream_ptr) - 1, len(stream_ptr) - 1, len(cols), slots, quantum, ways, sets, rounds], dtype=np.int64)
    with open(path, "wb") as f:
        hdr.tofile(f)
        np.asarray(wg_stream_ptr, dtype=np.int64).tofile(f)
        np.asarray(stream_ptr, dtype=np.int64).tofile(f)
        np.asarray(cols, dtype=np.int32).tofile(f)
    print("trace: %d workgroups, %d streams, %d accesses" % (hdr[0], hdr[1], hdr[2]))


def expand_rows(rowptr, rows):
    """entry indices of the given rows, concatenated in the given row order"""
    deg = (rowptr[rows + 1] - rowptr[rows]).astype(np.int64)
    total = int(deg.sum())
    starts = np.repeat(rowptr[rows].astype(np.int64) - np.concatenate([[0], np.cumsum(deg)[:-1]]), deg)
    return starts + np.arange(total, dtype=np.int64), deg


def items_of_rows(rowptr, col, rows, ymark_base):
    """per row: its columns then one negative 'output line' marker"""
    e, deg = expand_rows(rowptr, rows)
    n_items = deg + 1
    tot = int(n_items.sum())
    out = np.empty(tot, dtype=np.int32)
    ends = np.cumsum(n_items) - 1                      # marker positions
    mask = np.ones(tot, dtype=bool)
    mask[ends] = False
    out[mask] = col[e]
    out[ends] = -(ymark_base + rows.astype(np.int64) + 1).astype(np.int32)
    return out, n_items


def sched_current(rowptr, col, row_thresh=32, seg_len=32, gpb=16, **kw):
    """round-1 schedule: nnz-balanced row blocks (XCD-contiguous) for short rows + column-ordered 32-entry segments."""
    n = len(rowptr) - 1
    deg = np.diff(rowptr)
    is_seg = deg > row_thresh
    cost = np.where(is_seg, 0, deg) + 4
    cum = np.cumsum(cost)
    total = int(cum[-1])
    target = max(gpb * 16, min(gpb * 256, -(-total // 2048)))
    nblk = max(1, -(-total // target))
    marks = np.arange(1, nblk) * target
    inner = np.minimum(np.searchsorted(cum, marks) + 1, n)
    blk_row = np.concatenate([[0], inner, [n]])
    # dispatch id -> block (xcd_remap): block = base(xcd) + bid>>3
    bid = np.arange(nblk)
    q, r = nblk >> 3, nblk & 7
    xcd = bid & 7
    base = np.where(xcd < r, xcd * (q + 1), r * (q + 1) + (xcd - r) * q)
    block_of_bid = base + (bid >> 3)
    disp_of_block = np.empty(nblk, dtype=np.int64)
    disp_of_block[block_of_bid] = bid
    rows = np.nonzero(~is_seg)[0]
    b = np.searchsorted(blk_row, rows, side="right") - 1
    g = (rows - blk_row[b]) % gpb
    stream = disp_of_block[b] * gpb + g
    order = np.argsort(stream, kind="stable")
    rows_o = rows[order]
    items, n_items = items_of_rows(rowptr, col, rows_o, ymark_base=0)
    per_stream = np.bincount(stream[order], weights=n_items, minlength=nblk * gpb).astype(np.int64)
    # segments
    seg_rows = np.nonzero(is_seg)[0]
    nsegs = (deg[seg_rows] + seg_len - 1) // seg_len
    row_of = np.repeat(seg_rows, nsegs)
    first = np.cumsum(nsegs) - nsegs
    within = np.arange(int(nsegs.sum())) - np.repeat(first, nsegs)
    e0 = rowptr[row_of] + within * seg_len
    e1 = np.minimum(e0 + seg_len, rowptr[row_of + 1])
    perm = np.argsort(col[e0], kind="stable")
    e0, e1, row_of = e0[perm], e1[perm], row_of[perm]
    slen = (e1 - e0).astype(np.int64)
    tot = int(slen.sum())
    st = np.repeat(e0 - np.concatenate([[0], np.cumsum(slen)[:-1]]), slen) + np.arange(tot)
    seg_items = np.empty(tot + len(slen), dtype=np.int32)
    n_it = slen + 1
    ends = np.cumsum(n_it) - 1
    mask = np.ones(tot + len(slen), dtype=bool)
    mask[ends] = False
    seg_items[mask] = col[st]
    seg_items[ends] = -(n + 1 + np.arange(len(slen))).astype(np.int32)      # partial-sum / output line of the segment
    nseg = len(slen)
    nseg_wg = -(-nseg // gpb)
    per_stream2 = np.zeros(nseg_wg * gpb, dtype=np.int64)
    per_stream2[:nseg] = n_it
    cols = np.concatenate([items, seg_items])
    stream_ptr = np.concatenate([[0], np.cumsum(np.concatenate([per_stream, per_stream2]))])
    nwg = nblk + nseg_wg
    wg_stream_ptr = np.arange(nwg + 1) * gpb
    print("current: %d row blocks, %d segments in %d workgroups" % (nblk, nseg, nseg_wg))
    return wg_stream_ptr, stream_ptr, cols


SCHEDULES = {"current": sched_current}

if __name__ == "__main__":
    rowptr = np.load(sys.argv[1]).astype(np.int64)
    col = np.load(sys.argv[2])
    name, out = sys.argv[3], sys.argv[4]
    kw = {}
    for a in sys.argv[5:]:
        k, v = a.split("=")
        kw[k] = int(v)
    slots = kw.pop("slots", 256)
    quantum = kw.pop("quantum", 4)
    rounds = kw.pop("rounds", 0)
    import importlib
    fn = SCHEDULES.get(name)
    if fn is None:
        mod = importlib.import_module("sched_" + name)
        fn = mod.build
    w, s, c = fn(rowptr, col, **kw)
    write_trace(out, w, s, c, slots, quantum, rounds=rounds)
