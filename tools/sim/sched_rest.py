"""rows with deg <= T of the degree-relabelled graph on the round-1 schedule (row blocks + column-ordered segments)"""
import numpy as np
from mktrace import sched_current
from sched_sweep import relabel_by_degree


def build(rowptr, col, T=190, relabel=1, **kw):
    if relabel:
        rowptr, col = relabel_by_degree(rowptr, col)
    deg = np.diff(rowptr)
    keep_row = deg <= T
    keep = np.repeat(keep_row, deg)
    d2 = np.where(keep_row, deg, 0)
    rp2 = np.concatenate([[0], np.cumsum(d2)])
    return sched_current(rp2, col[keep], **kw)
