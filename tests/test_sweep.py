"""Sweep schedule of the long rows (tgcn_csr_sched ABI v3, csrc/hop.h hop_sweep_kernel): the host-side builder is checked
on the CPU by replaying its streams in numpy; the kernel is checked against the oracle with the schedule forced onto small
operands (shipped: only operands with >= 8 M entries in long rows take it) and at full width through linearity / adjoint
identities."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import cheb_oracle as O

TOL = 1e-5


def _hub_graph(n, rng, hubs):
    deg = rng.integers(0, 20, n)
    deg[rng.integers(0, n, n // 10)] = 0                       # isolated vertices
    for h, d in hubs:
        deg[h] = d
    row = np.repeat(np.arange(n), deg)
    col = rng.integers(0, n, row.shape[0])
    col[rng.random(row.shape[0]) < 0.3] //= 50                  # popular columns
    val = (rng.standard_normal(row.shape[0]) / 4).astype(np.float32)
    return row, col, val


def _replay(sw, X):
    """what hop_sweep_kernel computes, in numpy: per lane group one stream, sums per row slot"""
    G, SL = sw.groups, sw.slots
    ent = sw.ent.cpu().numpy()
    vals = sw.ent[:, 1].contiguous().view(torch.float32).cpu().numpy()
    gptr, srow, slot = (t.cpu().numpy() for t in (sw.gptr, sw.slot_row, sw.slot))
    out = {}
    seen = 0
    for wg in range(sw.rounds * sw.nwg):
        acc = np.zeros((SL, X.shape[1]))
        for g in range(G):
            for e in range(gptr[wg * G + g], gptr[wg * G + g + 1]):
                acc[slot[e]] += vals[e] * X[ent[e, 0]]
                seen += 1
        for sl in range(SL):
            r = srow[wg * SL + sl]
            if r >= 0:
                assert r not in out
                out[int(r)] = acc[sl]
    return out, seen


@pytest.mark.parametrize("lanes,nwg", [(16, 4), (4, 2), (64, 3)])
def test_sweep_builder_replay_cpu(lanes, nwg):
    from tgcn_amd import graph
    rng = np.random.default_rng(lanes)
    n = 2500
    row, col, val = _hub_graph(n, rng, hubs=((3, 2400), (77, 900), (1500, 5000), (2499, 333)) + tuple((int(h), 40 + int(h) % 300) for h in rng.integers(0, n, 60)))
    op = graph.GraphOperand.from_coo(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val))
    sw = graph.SweepSchedule.build(op.rowptr, op.edges, n, n, lanes, 32, nwg=nwg, panel_rows=97, force=True)
    assert sw is not None and sw.groups == 1024 // lanes and sw.slots == 8 * sw.groups
    deg = np.bincount(row, minlength=n)
    long_rows = np.nonzero(deg > 32)[0]
    X = rng.standard_normal((n, 3))
    got, seen = _replay(sw, X)
    assert seen == int(deg[long_rows].sum()) == sw.n_entries
    assert sorted(got) == long_rows.tolist()                   # every long row is written exactly once, no short row
    ref = O.coo_to_csr(row, col, val.astype(np.float64), n) @ X
    for r in long_rows:
        assert np.abs(got[r] - ref[r]).max() <= 1e-9 * max(1.0, np.abs(ref[r]).max())
    # lane groups of a workgroup walk the panels together: chunk c of the workgroup's (panel, slot, popularity) order goes to
    # lane group c mod G, so inside a stream the (panel, slot) key never decreases and streams differ by < 1 chunk in length
    cnt = np.bincount(col, minlength=n)
    rank = np.empty(n, dtype=np.int64)
    rank[np.argsort(-cnt, kind="stable")] = np.arange(n)
    ent, slot, gptr = sw.ent.numpy(), sw.slot.numpy().astype(np.int64), sw.gptr.numpy()
    G = sw.groups
    for wg in range(sw.rounds * sw.nwg):
        lens = np.diff(gptr[wg * G: (wg + 1) * G + 1])
        assert lens.max() - lens.min() <= lanes
        for g in range(0, G, 5):
            sl = slice(gptr[wg * G + g], gptr[wg * G + g + 1])
            key = np.minimum(rank[ent[sl, 0]] // 97, graph.SWEEP_HOT_PANELS) * sw.slots + slot[sl]
            assert np.all(np.diff(key) >= 0)
            if sw.nbar:                                    # where each of the first nbar panels ends inside the stream
                pan = np.minimum(rank[ent[sl, 0]] // 97, graph.SWEEP_HOT_PANELS)
                want = gptr[wg * G + g] + np.searchsorted(pan, np.arange(sw.nbar), side="right")
                assert np.array_equal(sw.pptr.numpy()[wg * G + g], want)


def test_sweep_not_built_for_small_operands():
    from tgcn_amd import graph
    rng = np.random.default_rng(1)
    row, col, val = _hub_graph(500, rng, hubs=((3, 400),))
    op = graph.GraphOperand.from_coo(500, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val))
    assert graph.SweepSchedule.build(op.rowptr, op.edges, 500, 500, 16, 32) is None


@pytest.fixture
def forced_sweep(monkeypatch):
    from tgcn_amd import graph
    monkeypatch.setattr(graph, "SWEEP_MIN_ENTRIES", 1)
    monkeypatch.setattr(graph, "SWEEP_WORKGROUPS", 6)
    return graph


@pytest.mark.gpu
@pytest.mark.parametrize("nb,n,C", [(1, 3000, 64), (3, 2000, 16), (2, 2500, 32), (1, 1800, 128), (2, 1500, 256), (1, 2200, 60), (2, 4000, 20)])
def test_sweep_hop_vs_oracle(nb, n, C, gpu_device, forced_sweep):
    from tgcn_amd import functional as F
    rng = np.random.default_rng(n + C)
    hubs = ((5, 700), (n - 1, 1300), (11, 9000), (n // 2, 33), (n // 3, 4100)) + tuple((int(h), 33 + int(h) % 500) for h in rng.integers(20, n - 20, 80))
    row, col, val = _hub_graph(n, rng, hubs)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    op = forced_sweep.GraphOperand.from_coo(n, dev(row), dev(col), dev(val))
    sched = op.schedule_for(C, True)
    assert sched.sweep is not None and sched.nseg == 0 and sched.sweep.rounds >= 1
    L = O.coo_to_csr(row, col, val, n)
    x = rng.standard_normal((nb, n, C)).astype(np.float32)
    z = rng.standard_normal((nb, n, C)).astype(np.float32)
    s = O._apply(L, x)
    y = F.csr_hop(op, dev(x))
    assert rel_err(y.cpu().numpy(), s) <= TOL
    y, p = F.csr_hop(op, dev(x), z=dev(z), alpha=2.0, beta=-1.0, want_p=True)
    assert rel_err(p.cpu().numpy(), s) <= TOL
    assert rel_err(y.cpu().numpy(), 2 * s - z) <= TOL
    # several lane groups add into one row's LDS slot: the order of those adds is not fixed, the result is (to parity tolerance)
    assert rel_err(F.csr_hop(op, dev(x), z=dev(z), alpha=2.0, beta=-1.0).cpu().numpy(), y.cpu().numpy()) <= 1e-6
    # rows at or below the threshold keep the fixed order of the row kernel: bitwise identical
    short = torch.as_tensor(np.bincount(row, minlength=n) <= 32).cuda()
    assert torch.equal(F.csr_hop(op, dev(x))[:, short], F.csr_hop(op, dev(x))[:, short])
    # rows of X holding Inf / NaN propagate as in a per-entry evaluation
    x2 = x.copy()
    x2[0, 7, :] = np.inf
    ref = O._apply(L, x2.astype(np.float64))
    got = F.csr_hop(op, dev(x2)).cpu().numpy()
    assert np.array_equal(np.isfinite(got), np.isfinite(ref))


@pytest.mark.gpu
def test_sweep_unaligned_rows_take_the_segment_path(gpu_device, forced_sweep):
    """C % 4 != 0 (scalar loads) cannot use the sweep kernel: such calls get a schedule with segments instead"""
    from tgcn_amd import functional as F
    rng = np.random.default_rng(9)
    n = 1500
    row, col, val = _hub_graph(n, rng, ((5, 900), (9, 2000)))
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    op = forced_sweep.GraphOperand.from_coo(n, dev(row), dev(col), dev(val))
    assert op.schedule_for(15, False).sweep is None
    x = rng.standard_normal((2, n, 15)).astype(np.float32)
    assert rel_err(F.csr_hop(op, dev(x)).cpu().numpy(), O._apply(O.coo_to_csr(row, col, val, n), x)) <= TOL


@pytest.mark.gpu
def test_sweep_linearity_adjoint_large(gpu_device, monkeypatch):
    """shipped geometry (256 workgroups, 2 MB panels) on a 14 M-entry operand with skewed rows and columns"""
    from tgcn_amd import functional as F, graph
    monkeypatch.setattr(graph, "SWEEP_MIN_ENTRIES", 1_000_000)
    g = torch.Generator(device="cuda").manual_seed(5)
    n, m, C = 600_000, 14_000_000, 64
    row = (torch.rand(m, device="cuda", generator=g) ** 4 * n).long().clamp_(max=n - 1)          # long rows
    col = (torch.rand(m, device="cuda", generator=g) ** 3 * n).long().clamp_(max=n - 1)          # popular columns
    val = torch.randn(m, device="cuda", generator=g) * 0.1
    op = graph.GraphOperand.from_coo(n, row, col, val)
    assert op.schedule_for(C, True).sweep is not None
    x1 = torch.randn(1, n, C, device="cuda", generator=g)
    x2 = torch.randn(1, n, C, device="cuda", generator=g)
    y1 = F.csr_hop(op, x1)
    lhs = F.csr_hop(op, 0.5 * x1 + x2)
    rhs = 0.5 * y1 + F.csr_hop(op, x2)
    assert rel_err(lhs.cpu().numpy(), rhs.cpu().numpy()) <= TOL
    opT = op.transpose()
    a = (y1.double() * x2.double()).sum()
    b = (x1.double() * F.csr_hop(opT, x2).double()).sum()
    assert abs(a - b) <= 1e-6 * max(abs(a), abs(b), 1.0)
    # same rows as the segment schedule gives (different summation order: tolerance, not bits)
    monkeypatch.setattr(graph, "SWEEP", False)
    op2 = graph.GraphOperand.from_coo(n, row, col, val)
    assert op2.schedule_for(C, True).sweep is None
    assert rel_err(F.csr_hop(op2, x1).cpu().numpy(), y1.cpu().numpy()) <= TOL
