"""The C ABI on its own: operand and schedule built by the library (tgcn_graph_* / tgcn_sched_*), a hop and a whole layer
run through bare ctypes -- nothing of tgcn_amd/graph.py, functional.py or nn.py is used; torch only hands out device
memory.  This is the binding INTEGRATION.md shows for a maintainer of the reference."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_files, load_golden, rel_err
from oracle import cheb_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


class Csr(C.Structure):
    _fields_ = [("n", C.c_int64), ("nnz", C.c_int64), ("rowptr", C.c_void_p), ("edges", C.c_void_p), ("dense", C.c_void_p)]


class Dense(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("batch_stride", C.c_int64), ("row_stride", C.c_int64)]


@pytest.fixture(scope="module")
def lib(gpu_device):
    h = C.CDLL(os.path.join(ROOT, "tgcn_amd", "lib", "libtgcn_hip.so"))
    h.tgcn_last_error.restype = C.c_char_p
    h.tgcn_graph_csr.restype = C.POINTER(Csr)
    h.tgcn_graph_csr.argtypes = [C.c_void_p]
    h.tgcn_sched_get.restype = C.c_void_p
    h.tgcn_sched_get.argtypes = [C.c_void_p]
    h.tgcn_graph_destroy.argtypes = [C.c_void_p]
    h.tgcn_sched_destroy.argtypes = [C.c_void_p]
    h.tgcn_csr_hop_workspace_bytes.restype = C.c_size_t
    h.tgcn_csr_hop_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int]
    h.tgcn_cheb_forward_workspace_bytes.restype = C.c_size_t
    h.tgcn_cheb_forward_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64]
    return h


def _ok(lib, rc):
    assert rc == 0, lib.tgcn_last_error().decode()


def _p(t):
    return C.c_void_p(t.data_ptr())


def _graph(lib, n, n_cols, row, col, val):
    r, c, v = (torch.as_tensor(a).cuda() for a in (row.astype(np.int64), col.astype(np.int64), val.astype(np.float32)))
    g = C.c_void_p()
    _ok(lib, lib.tgcn_graph_create_from_coo(C.c_int64(n), C.c_int64(n_cols), C.c_int64(len(row)), _p(r), _p(c), _p(v), C.byref(g)))
    return g


@pytest.mark.parametrize("C_row", [64, 28, 300])
def test_hop_through_bare_ctypes(lib, C_row):
    rng = np.random.default_rng(C_row)
    n, nb = 2500, 2
    deg = rng.integers(0, 12, n)
    deg[[3, 77, 1500]] = [2400, 700, 40]                        # long rows -> segments, partial sums, fix-up
    row = np.repeat(np.arange(n), deg)
    col = rng.integers(0, n, row.shape[0])
    val = (rng.standard_normal(row.shape[0]) / 3).astype(np.float32)
    g = _graph(lib, n, n, row, col, val)
    csr = lib.tgcn_graph_csr(g)
    assert csr.contents.n == n and csr.contents.nnz == len(row)
    al = 1 if C_row % 4 == 0 else 0
    s = C.c_void_p()
    _ok(lib, lib.tgcn_sched_build(g, C.c_int32(C_row), C.c_int(al), C.byref(s)))
    sched = lib.tgcn_sched_get(s)
    x = torch.randn(nb, n, C_row, device="cuda")
    z = torch.randn(nb, n, C_row, device="cuda")
    y = torch.empty_like(x)
    ws = torch.empty(max(16, lib.tgcn_csr_hop_workspace_bytes(sched, nb, C_row, al)), dtype=torch.uint8, device="cuda")
    X, Z, Y = (Dense(t.data_ptr(), n * C_row, C_row) for t in (x, z, y))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _ok(lib, lib.tgcn_csr_hop_f32(stream, csr, C.c_void_p(sched), C.c_int32(nb), C.c_int32(C_row), C.byref(X), C.byref(Z),
                                  C.c_float(2.0), C.c_float(-1.0), C.byref(Y), None, _p(ws), C.c_size_t(ws.numel())))
    ref = 2 * O._apply(O.coo_to_csr(row, col, val, n), x.cpu().numpy()) - z.cpu().numpy()
    assert rel_err(y.cpu().numpy(), ref) <= TOL
    lib.tgcn_sched_destroy(s)
    lib.tgcn_graph_destroy(g)


@pytest.mark.parametrize("n", [4000, 200000])
def test_library_schedule_equals_the_python_builder(lib, n):
    """tgcn_sched_build against tgcn_amd/graph.py::Schedule, array by array (the Python builder is what the modules use);
    n = 200000 is large enough for the block-size cap of narrow rows to apply"""
    from tgcn_amd.graph import GraphOperand
    rng = np.random.default_rng(2)
    deg = rng.integers(0, 25, n)
    deg[rng.integers(0, n, 30)] = rng.integers(33, 3000, 30)
    row = np.repeat(np.arange(n), deg)
    col = rng.integers(0, n, row.shape[0])
    val = rng.standard_normal(row.shape[0]).astype(np.float32)
    g = _graph(lib, n, n, row, col, val)
    op = GraphOperand.from_coo(n, torch.as_tensor(row).cuda(), torch.as_tensor(col).cuda(), torch.as_tensor(val).cuda())
    csr = lib.tgcn_graph_csr(g).contents
    rp = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(_p(rp), C.c_void_p(csr.rowptr), C.c_size_t(4 * (n + 1)), 3)
    assert torch.equal(rp, op.rowptr)
    from tgcn_amd._lib import SchedStruct
    for C_row in (64, 16, 300):
        s = C.c_void_p()
        _ok(lib, lib.tgcn_sched_build(g, C.c_int32(C_row), C.c_int(1), C.byref(s)))
        mine = C.cast(lib.tgcn_sched_get(s), C.POINTER(SchedStruct)).contents
        from tgcn_amd.graph import Schedule
        lanes = lib.tgcn_hop_lanes_per_row(C_row, 1)
        py = Schedule(op.rowptr, op.n, lanes, edges=op.edges, builder="torch")        # the torch index-op builder: the cross-check
        for f in ("lanes_per_row", "row_thresh", "nblk", "nseg", "nlong", "nhuge", "npartial", "seg_mode", "row_mix", "nwseg"):
            assert getattr(mine, f) == getattr(py.struct, f), f
        for f, cnt in (("blk_row", py.nblk + 1), ("seg_row", py.nseg), ("seg_e0", py.nseg), ("seg_e1", py.nseg), ("seg_slot", py.nseg),
                       ("long_row", py.nlong), ("long_slot", py.nlong + 1)):
            got = torch.empty(cnt, dtype=torch.int32, device="cuda")
            C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(_p(got), C.c_void_p(getattr(mine, f)), C.c_size_t(4 * cnt), 3)
            assert torch.equal(got, getattr(py, f)[:cnt]), f
        lib.tgcn_sched_destroy(s)
    lib.tgcn_graph_destroy(g)


def test_chebconv_layer_through_bare_ctypes(lib):
    """the whole ChebConv forward of a reference fixture: operand from the edge list (tgcn_graph_create_from_edge_index),
    schedule, tgcn_cheb_forward_f32 mode 1 -- against the reference's output"""
    path = [p for p in golden_files("ChebConv_") if "rmat1024" in p][0]
    g = load_golden(path)
    n, E = int(g["n"]), g["edge_index"].shape[1]
    ei = torch.as_tensor(g["edge_index"].astype(np.int64)).cuda().contiguous()
    w = torch.as_tensor(g["edge_weight"].astype(np.float32)).cuda() if int(g["use_weight"]) else None
    gr = C.c_void_p()
    _ok(lib, lib.tgcn_graph_create_from_edge_index(C.c_int64(n), C.c_int64(E), _p(ei), _p(w) if w is not None else None, C.byref(gr)))
    x = torch.as_tensor(g["x"]).cuda()
    x = x if x.dim() == 3 else x.unsqueeze(-1)
    q, _, f = x.shape
    K, _, gg = g["weight"].shape
    s = C.c_void_p()
    _ok(lib, lib.tgcn_sched_build(gr, C.c_int32(f), C.c_int(1 if f % 4 == 0 else 0), C.byref(s)))
    sched = lib.tgcn_sched_get(s)
    W = torch.as_tensor(g["weight"]).cuda().reshape(K * f, gg).contiguous()
    b = torch.as_tensor(g["bias"]).cuda() if int(g["has_bias"]) else None
    out = torch.empty(q, n, gg, device="cuda")
    ws = torch.empty(max(256, lib.tgcn_cheb_forward_workspace_bytes(sched, K, q, n, f, 0, 0)), dtype=torch.uint8, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _ok(lib, lib.tgcn_cheb_forward_f32(stream, lib.tgcn_graph_csr(gr), C.c_void_p(sched), C.c_int32(1), C.c_int32(K), C.c_int64(q), C.c_int64(n),
                                       C.c_int32(f), C.c_int32(gg), _p(x.contiguous()), _p(W), _p(b) if b is not None else None,
                                       C.c_int32(1 if b is not None else 0), _p(out), C.c_int32(0), C.c_int64(0), _p(ws), C.c_size_t(ws.numel())))
    assert rel_err(out.cpu().numpy(), g["out"]) <= TOL
    lib.tgcn_sched_destroy(s)
    lib.tgcn_graph_destroy(gr)
