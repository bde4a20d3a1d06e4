"""short rows (deg <= T) processed in the order of their hottest (lowest-rank) column instead of row order"""
import numpy as np
from mktrace import items_of_rows
from sched_sweep import relabel_by_degree


def build(rowptr, col, T=32, relabel=1, key=0, gpb=16, **kw):
    if relabel:
        rowptr, col = relabel_by_degree(rowptr, col)
    n = len(rowptr) - 1
    deg = np.diff(rowptr)
    rows = np.nonzero((deg <= T) & (deg > 0))[0]
    if key == 0:
        k = col[rowptr[rows]]                      # first (lowest-rank) column
    elif key == 1:
        k = col[rowptr[rows + 1] - 1]              # last column
    else:
        k = col[(rowptr[rows] + rowptr[rows + 1]) // 2]
    rows = rows[np.argsort(k, kind="stable")]
    iso = np.nonzero(deg == 0)[0]
    rows = np.concatenate([rows, iso])
    cost = deg[rows] + 4
    cum = np.cumsum(cost)
    target = 4096
    nblk = -(-int(cum[-1]) // target)
    b = np.minimum(cum // target, nblk - 1)
    first = np.searchsorted(b, np.arange(nblk))
    g = (np.arange(len(rows)) - first[b]) % gpb
    # XCD-contiguous ranges of blocks
    bid = np.arange(nblk)
    q, r = nblk >> 3, nblk & 7
    xcd = bid & 7
    base = np.where(xcd < r, xcd * (q + 1), r * (q + 1) + (xcd - r) * q)
    block_of_bid = base + (bid >> 3)
    if kw.get("noremap", 0):
        block_of_bid = bid
    disp = np.empty(nblk, dtype=np.int64)
    disp[block_of_bid] = bid
    stream = disp[b] * gpb + g
    o = np.argsort(stream, kind="stable")
    items, n_items = items_of_rows(rowptr, col, rows[o], 0)
    per_stream = np.bincount(stream[o], weights=n_items, minlength=nblk * gpb).astype(np.int64)
    return np.arange(nblk + 1) * gpb, np.concatenate([[0], np.cumsum(per_stream)]), items
