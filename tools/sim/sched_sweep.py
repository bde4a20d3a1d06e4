"""Row-block-stationary, column-ordered sweep for long rows (accumulators on chip), degree-sorted internal labels."""
import numpy as np
from mktrace import items_of_rows, expand_rows


def relabel_by_degree(rowptr, col):
    n = len(rowptr) - 1
    deg = np.diff(rowptr)
    order = np.argsort(-deg, kind="stable")
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(n)
    e, d = expand_rows(rowptr, order)
    c2 = rank[col[e]].astype(np.int32)
    rp2 = np.concatenate([[0], np.cumsum(d)])
    # sort columns inside each row
    rowid = np.repeat(np.arange(n), d)
    o = np.lexsort((c2, rowid))
    return rp2, c2[o]


def build(rowptr, col, T=32, RB=512, E=65536, G=64, P=8192, part="long", relabel=1, hot=0, **kw):
    if relabel:
        rowptr, col = relabel_by_degree(rowptr, col)
    n = len(rowptr) - 1
    deg = np.diff(rowptr)
    long_rows = np.nonzero(deg > T)[0]
    if part == "short":
        # short rows in row order, 16 lane groups per 256-thread workgroup, nnz-balanced blocks, XCD-contiguous ranges
        gpb = 16
        is_seg = deg > T
        cost = np.where(is_seg, 0, deg) + 4
        cum = np.cumsum(cost)
        total = int(cum[-1])
        target = 4096
        nblk = max(1, -(-total // target))
        marks = np.arange(1, nblk) * target
        inner = np.minimum(np.searchsorted(cum, marks) + 1, n)
        blk_row = np.concatenate([[0], inner, [n]])
        bid = np.arange(nblk)
        q, r = nblk >> 3, nblk & 7
        xcd = bid & 7
        base = np.where(xcd < r, xcd * (q + 1), r * (q + 1) + (xcd - r) * q)
        block_of_bid = base + (bid >> 3)
        if kw.get("noremap", 0):
            block_of_bid = bid
        disp_of_block = np.empty(nblk, dtype=np.int64)
        disp_of_block[block_of_bid] = bid
        rows = np.nonzero(~is_seg)[0]
        b = np.searchsorted(blk_row, rows, side="right") - 1
        g = (rows - blk_row[b]) % gpb
        stream = disp_of_block[b] * gpb + g
        order = np.argsort(stream, kind="stable")
        items, n_items = items_of_rows(rowptr, col, rows[order], 0)
        per_stream = np.bincount(stream[order], weights=n_items, minlength=nblk * gpb).astype(np.int64)
        return np.arange(nblk + 1) * gpb, np.concatenate([[0], np.cumsum(per_stream)]), items
    # ---- long rows: blocks of <= RB rows and ~E entries (rows longer than E are cut into pieces of E entries)
    e, d = expand_rows(rowptr, long_rows)
    rowid = np.repeat(long_rows, d)
    c = col[e].astype(np.int64)
    # piece index for very long rows
    pos = np.arange(len(e)) - np.repeat(np.cumsum(d) - d, d)
    piece = pos // E
    # unit = (row, piece); greedy blocks over units in order
    unit_change = np.ones(len(e), dtype=bool)
    unit_change[1:] = (rowid[1:] != rowid[:-1]) | (piece[1:] != piece[:-1])
    unit_id = np.cumsum(unit_change) - 1
    nunits = int(unit_id[-1]) + 1
    ulen = np.bincount(unit_id)
    blk = np.zeros(nunits, dtype=np.int64)
    b = 0
    rows_in = 0
    ent_in = 0
    NW = kw.get("strided", 0)
    if NW:
        # unit u -> round u // (NW*RB), workgroup u % NW of that round: every workgroup gets the same degree mix
        blk = (np.arange(nunits) // (NW * RB)) * NW + (np.arange(nunits) % NW)
        # re-sort units by block so that blocks are contiguous
        uo = np.argsort(blk, kind="stable")
        newid = np.empty(nunits, dtype=np.int64); newid[uo] = np.arange(nunits)
        unit_id = newid[unit_id]
        eo = np.argsort(unit_id, kind="stable")
        e, rowid, c, unit_id = e[eo], rowid[eo], c[eo], unit_id[eo]
        blk = blk[uo]
        ulen = ulen[uo]
        b = int(blk[-1])
    for u in range(0 if NW else nunits):          # a few hundred thousand iterations
        if rows_in >= RB or (ent_in + ulen[u] > E and rows_in > 0):
            b += 1
            rows_in = 0
            ent_in = 0
        blk[u] = b
        rows_in += 1
        ent_in += ulen[u]
    nb = b + 1
    first_unit = np.concatenate([[0], np.nonzero(np.diff(blk))[0] + 1])
    uslot = np.arange(nunits) - first_unit[blk]            # unit's index inside its block
    grp = uslot % G
    eb = blk[unit_id]
    eg = grp[unit_id]
    panel = c // P if not hot else np.where(c < hot, c // P, (hot // P) + 0 * c)
    stream = eb * G + eg
    if kw.get("chunked", 1):
        # entries of a block in (panel, row, col) order, dealt to the lane groups in chunks of 16
        key2 = np.where(c < hot, panel, hot // P + 1) if hot else panel
        o = np.lexsort((c, unit_id, key2, eb))
        posb = np.arange(len(e)) - np.repeat(np.cumsum(np.bincount(eb, minlength=nb)) - np.bincount(eb, minlength=nb), np.bincount(eb, minlength=nb))
        g2 = (posb // 16) % G
        stream_o = eb[o] * G + g2
        o2 = np.argsort(stream_o, kind="stable")
        cols = c[o][o2].astype(np.int32)
        per_stream = np.bincount(stream_o, minlength=nb * G)
        nbar = kw.get("barriers", 0)          # barrier after each of the first `nbar` panels
        if nbar:
            pan = (cols.astype(np.int64) // P)
            so = stream_o[o2]
            ns = nb * G
            m_stream = np.repeat(np.arange(ns), nbar)
            m_pan = np.tile(np.arange(nbar), ns)
            all_stream = np.concatenate([so, m_stream])
            all_pan = np.concatenate([pan, m_pan])
            all_ism = np.concatenate([np.zeros(len(so), dtype=np.int8), np.ones(len(m_stream), dtype=np.int8)])
            all_val = np.concatenate([cols, np.full(len(m_stream), -2**31, dtype=np.int64).astype(np.int32)])
            seq = np.concatenate([np.arange(len(so)), np.zeros(len(m_stream), dtype=np.int64)])
            oo = np.lexsort((seq, all_ism, all_pan, all_stream))
            cols = all_val[oo]
            per_stream = per_stream + nbar
        print("sweep(chunked): %d long rows, %d entries, %d blocks (RB=%d E=%d G=%d P=%d)" % (len(long_rows), len(e), nb, RB, E, G, P))
        return np.arange(nb + 1) * G, np.concatenate([[0], np.cumsum(per_stream)]), cols
    if hot:
        # hot part swept in panel order, cold remainder in row order
        key2 = np.where(c < hot, panel, hot // P + 1)
        o = np.lexsort((c, unit_id, key2, stream))
    else:
        o = np.lexsort((c, unit_id, panel, stream))
    cols = c[o].astype(np.int32)
    per_stream = np.bincount(stream, minlength=nb * G)
    print("sweep: %d long rows, %d entries, %d units, %d blocks (RB=%d E=%d G=%d P=%d)" % (len(long_rows), len(e), nunits, nb, RB, E, G, P))
    return np.arange(nb + 1) * G, np.concatenate([[0], np.cumsum(per_stream)]), cols
