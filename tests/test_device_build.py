"""Operand and schedule construction on the device (csrc/device_build.h: own stable radix sort, prefix sums, binary-search marks)
against the torch index-op builders of tgcn_amd/graph.py, array by array -- integer work, so the bar is exact equality.
Covers what the reference does on the host before its layers: COO in any order with duplicates (scatter order,
tgcn/nn/gcn.py:308,343), the edge-list operand of ChebConv (gcn.py:398-413), CSR input, rectangular shard operands."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _coo(n, n_cols, nnz, rng, hubs=True):
    row = rng.integers(0, n, nnz)
    col = rng.integers(0, n_cols, nnz)
    if hubs and nnz > 1000:
        row[: nnz // 5] = rng.integers(0, min(3, n), nnz // 5)     # a few very long rows
        col[nnz // 5: nnz // 3] = rng.integers(0, min(2, n_cols), nnz // 3 - nnz // 5)   # hub columns, many duplicates of one (row, col)
    val = rng.standard_normal(nnz).astype(np.float32)
    return torch.as_tensor(row).cuda(), torch.as_tensor(col).cuda(), torch.as_tensor(val).cuda()


@pytest.mark.parametrize("n,n_cols,nnz", [(1, 1, 0), (7, 7, 1), (100, 100, 37), (1000, 1300, 50_000), (300_000, 300_000, 5_000_000),
                                          (70_000, 20, 900_000), (5, 100_000, 400_000), (4096, 4096, 4096 * 16),
                                          (1, 1, 5000), (1, 70_000, 300_000), (70_000, 1, 300_000), (20_000_000, 20_000_000, 3_000_000),
                                          (257, 65_537, 4097), (65_536, 256, 4096), (3, 3, 2_000_000)])
def test_csr_build_equals_torch_builder(n, n_cols, nnz, gpu_device, monkeypatch):
    from tgcn_amd import graph
    rng = np.random.default_rng(n + nnz)
    row, col, val = _coo(n, n_cols, nnz, rng)
    monkeypatch.setattr(graph, "BUILDER", "library")
    a = graph.GraphOperand.from_coo(n, row, col, val, n_cols=n_cols)
    monkeypatch.setattr(graph, "BUILDER", "torch")
    b = graph.GraphOperand.from_coo(n, row, col, val, n_cols=n_cols)
    assert a.nnz == b.nnz == nnz and a.n_cols == b.n_cols
    assert torch.equal(a.rowptr, b.rowptr)
    assert torch.equal(a.edges[: nnz], b.edges[: nnz])          # same entries in the same order: duplicates keep their given order


def test_csr_build_rejects_out_of_range_indices(gpu_device):
    from tgcn_amd import graph, _lib
    row = torch.tensor([0, 1, 5], device="cuda")
    col = torch.tensor([0, 1, 2], device="cuda")
    val = torch.ones(3, device="cuda")
    for r, c in ((row, col), (col, row), (torch.tensor([0, -1, 2], device="cuda"), col)):
        with pytest.raises(_lib.TgcnError, match="vertex index outside"):
            graph.GraphOperand.from_coo(3, r, c, val)


@pytest.mark.parametrize("C_row", [64, 16, 4, 300, 1])
@pytest.mark.parametrize("n,nnz", [(50, 200), (3000, 40_000), (400_000, 6_000_000), (1, 40), (1, 5000), (40, 0), (2000, 2000 * 40), (5, 700_000)])
def test_schedule_equals_torch_builder(n, nnz, C_row, gpu_device, monkeypatch):
    from tgcn_amd import graph, _lib
    rng = np.random.default_rng(n + C_row)
    row, col, val = _coo(n, n, nnz, rng, hubs=n > 10)       # (1, 5000): one row of 157 segments; (2000, 80000): every row is cut into segments
    op = graph.GraphOperand.from_coo(n, row, col, val)
    lanes = _lib.lib().tgcn_hop_lanes_per_row(C_row, 1)
    lib_s = graph.Schedule(op.rowptr, op.n, lanes, edges=op.edges, builder="library")
    py_s = graph.Schedule(op.rowptr, op.n, lanes, edges=op.edges, builder="torch")
    for f in ("lanes_per_row", "row_thresh", "nblk", "nseg", "nlong", "nhuge", "npartial", "seg_mode", "row_mix", "nwseg"):
        assert getattr(lib_s, f) == getattr(py_s, f), f
    for f, cnt in (("blk_row", py_s.nblk + 1), ("seg_row", py_s.nseg), ("seg_e0", py_s.nseg), ("seg_e1", py_s.nseg), ("seg_slot", py_s.nseg),
                   ("long_row", py_s.nlong), ("long_slot", py_s.nlong + 1 if py_s.nlong else 0)):
        assert torch.equal(getattr(lib_s, f)[:cnt], getattr(py_s, f)[:cnt]), f


def test_library_graph_entry_points_from_csr_and_edge_index(gpu_device):
    """tgcn_graph_create_from_csr (columns unsorted inside the rows, int64 row pointers) and _from_edge_index (self loops, isolated
    vertices, weights) against the torch builders"""
    from tgcn_amd import graph, _lib
    L = _lib.lib()
    rng = np.random.default_rng(4)
    n = 5000
    deg = rng.integers(0, 30, n)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, rp[-1]).astype(np.int32)
    val = rng.standard_normal(rp[-1]).astype(np.float32)
    d = lambda a: torch.as_tensor(a).cuda()
    rp_d, col_d, val_d = d(rp), d(col), d(val)
    g = C.c_void_p()
    _lib.check(L.tgcn_graph_create_from_csr(n, n, _lib.ptr(rp_d), _lib.ptr(col_d), _lib.ptr(val_d), C.byref(g)))
    csr = L.tgcn_graph_csr(g).contents
    row = np.repeat(np.arange(n), deg)
    ref = graph.GraphOperand.from_coo(n, d(row), d(col.astype(np.int64)), val_d)
    hip = C.cdll.LoadLibrary("libamdhip64.so")
    got_rp = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    got_e = torch.empty((csr.nnz, 2), dtype=torch.int32, device="cuda")
    hip.hipMemcpy(_lib.ptr(got_rp), C.c_void_p(csr.rowptr), C.c_size_t(4 * (n + 1)), 3)
    hip.hipMemcpy(_lib.ptr(got_e), C.c_void_p(csr.edges), C.c_size_t(8 * csr.nnz), 3)
    assert torch.equal(got_rp, ref.rowptr) and torch.equal(got_e, ref.edges[: csr.nnz])
    L.tgcn_graph_destroy(g)
    # edge list with self loops, a vertex without outgoing edges, weights
    E = 40_000
    ei = rng.integers(0, n, (2, E)).astype(np.int64)
    ei[:, :500] = ei[0, :500]                      # self loops
    ei[0][ei[0] == 17] = 18                        # vertex 17 has no outgoing edge
    w = rng.random(E).astype(np.float32)
    ei_d, w_d = d(ei), d(w)
    _lib.check(L.tgcn_graph_create_from_edge_index(n, E, _lib.ptr(ei_d), _lib.ptr(w_d), C.byref(g)))
    csr = L.tgcn_graph_csr(g).contents
    ref = graph.GraphOperand.from_edge_index(ei_d, w_d, n)
    got_rp = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    got_e = torch.empty((csr.nnz, 2), dtype=torch.int32, device="cuda")
    hip.hipMemcpy(_lib.ptr(got_rp), C.c_void_p(csr.rowptr), C.c_size_t(4 * (n + 1)), 3)
    hip.hipMemcpy(_lib.ptr(got_e), C.c_void_p(csr.edges), C.c_size_t(8 * csr.nnz), 3)
    assert csr.nnz == ref.nnz and torch.equal(got_rp, ref.rowptr)
    assert torch.equal(got_e[:, 0], ref.edges[: csr.nnz, 0])
    a, b = got_e[:, 1].contiguous().view(torch.float32), ref.edges[: csr.nnz, 1].contiguous().view(torch.float32)
    assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())       # rsqrt forms differ in the last bit (1/sqrtf vs pow(-0.5))
    L.tgcn_graph_destroy(g)


@pytest.mark.parametrize("n,E,weighted", [(1, 0, False), (5, 9, True), (3000, 40_000, False), (3000, 40_000, True), (200_000, 3_000_000, True), (50, 5000, True)])
def test_edge_list_operand_by_the_library_equals_torch_builder(n, E, weighted, gpu_device, monkeypatch):
    """GraphOperand.from_edge_index on the product path = tgcn_edge_normalise_f32 + tgcn_csr_build_f32 (one builder).
    Against the torch index-op form kept as the cross-check: same entries in the same order (self loops, duplicates, vertices without
    outgoing edges, weights); the values agree to a few ulp (integer degrees on both sides; torch.pow(-0.5) against 1 / sqrtf)."""
    from tgcn_amd import graph
    rng = np.random.default_rng(n + E)
    ei = torch.as_tensor(rng.integers(0, n, (2, E))).cuda()
    if E > 10:
        ei[1, : E // 10] = ei[0, : E // 10]                         # self loops
        ei[0, E // 2:] = ei[0, E // 2:] % max(1, n // 2)            # the upper half of the vertices mostly without outgoing edges
    w = torch.as_tensor(rng.uniform(0.1, 2.0, E).astype(np.float32)).cuda() if weighted else None
    monkeypatch.setattr(graph, "BUILDER", "library")
    a = graph.GraphOperand.from_edge_index(ei, w, n)
    monkeypatch.setattr(graph, "BUILDER", "torch")
    b = graph.GraphOperand.from_edge_index(ei, w, n)
    assert a.nnz == b.nnz and torch.equal(a.rowptr, b.rowptr)
    assert torch.equal(a.edges[: a.nnz, 0], b.edges[: b.nnz, 0])
    va, vb = a.edges[: a.nnz, 1].contiguous().view(torch.float32), b.edges[: b.nnz, 1].contiguous().view(torch.float32)
    assert torch.allclose(va, vb, rtol=1e-6, atol=0)          # deg.pow(-0.5) against 1 / sqrtf(deg): a few ulp


def test_edge_list_operand_rejects_bad_input(gpu_device):
    from tgcn_amd import graph, _lib
    with pytest.raises(_lib.TgcnError, match="vertex index outside"):
        graph.GraphOperand.from_edge_index(torch.tensor([[0, 1, 7], [1, 2, 0]], device="cuda"), None, 3)
    with pytest.raises(_lib.TgcnError, match="requires_grad"):
        graph.GraphOperand.from_edge_index(torch.tensor([[0, 1], [1, 2]], device="cuda"), torch.ones(2, device="cuda", requires_grad=True), 3)


@pytest.mark.parametrize("n,m,lmax", [(1, 0, 2.0), (40, 300, 2.0), (40, 300, 1.3), (5000, 80_000, 2.0), (300_000, 4_000_000, 1.7), (7, 50_000, 2.0)])
def test_adjacency_operand_by_the_library_equals_torch_builder(n, m, lmax, gpu_device, monkeypatch):
    """GraphOperand.from_adjacency on the product path = tgcn_adjacency_normalise_f32 (stable sort by column, one wave per column in a fixed
    order, values, diagonal) + tgcn_csr_build_f32, against the torch form (index_add_: float atomics) and bitwise against itself."""
    from tgcn_amd import graph
    rng = np.random.default_rng(n + m)
    row = torch.as_tensor(rng.integers(0, n, m)).cuda()
    col = torch.as_tensor(rng.integers(0, n, m)).cuda()
    if m > 1000:
        col[: m // 4] = col[: m // 4] % 3                           # hub columns: thousands of terms in one column sum
    w = torch.as_tensor(rng.uniform(0.1, 2.0, m).astype(np.float32)).cuda()
    monkeypatch.setattr(graph, "BUILDER", "library")
    a = graph.GraphOperand.from_adjacency(n, row, col, w, lmax=lmax)
    a2 = graph.GraphOperand.from_adjacency(n, row, col, w, lmax=lmax)
    monkeypatch.setattr(graph, "BUILDER", "torch")
    b = graph.GraphOperand.from_adjacency(n, row, col, w, lmax=lmax)
    assert a.nnz == b.nnz == m + (n if lmax != 2.0 else 0) and torch.equal(a.rowptr, b.rowptr)
    assert torch.equal(a.edges[: a.nnz], a2.edges[: a.nnz])          # deterministic: no float atomics
    assert torch.equal(a.edges[: a.nnz, 0], b.edges[: b.nnz, 0])
    va, vb = a.edges[: a.nnz, 1].contiguous().view(torch.float32), b.edges[: b.nnz, 1].contiguous().view(torch.float32)
    assert torch.allclose(va, vb, rtol=1e-3, atol=1e-30)             # the torch form sums a hub column's 10^5 terms with fp32 atomics in any order
    # the bar: the same values computed in float64 (the reference computes d = W.sum(axis=0) in W's dtype on the host, gcn/graph.py:124)
    d = torch.zeros(n, dtype=torch.float64, device="cuda").index_add_(0, col, w.double()) + 1.401298464324817e-45
    dis = 1.0 / torch.sqrt(d)
    ref = -(2.0 / lmax) * dis[row] * w.double() * dis[col]
    # entries of the operand are sorted by (row, col) with duplicates in their given order: sort the reference the same way
    if lmax != 2.0:
        diag = torch.arange(n, device="cuda")
        row2, col2, ref = torch.cat([row, diag]), torch.cat([col, diag]), torch.cat([ref, torch.full((n,), 2.0 / lmax - 1.0, dtype=torch.float64, device="cuda")])
    else:
        row2, col2 = row, col
    order = torch.argsort(row2 * n + col2, stable=True)
    assert torch.allclose(va.double(), ref[order], rtol=1e-5, atol=1e-30)
