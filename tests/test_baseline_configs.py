"""BASELINE.json's configurations under -m gpu, at their own sizes where the oracle finishes in seconds and at a reduced size
for the 160 M-entry graph: the module forward through the HIP path against oracle/cheb_ref.c (the C restatement pinned by the
reference's fixtures) on the same seeded inputs.  Tolerance max|a-b| / max|b| <= 1e-5."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _host_csr(op):
    row, col, val = op.coo()
    return op.rowptr.cpu().numpy(), col.to(torch.int32).cpu().numpy(), val.cpu().numpy()


def _check(layer, op, x, kind, q_check=None, samples=None):
    """layer(x) against the C restatement on the first q_check samples, or on the listed `samples` (reference basis weight, unfolded recursion)"""
    from oracle import c_port
    with torch.no_grad():
        out = layer(x)
    sel = list(range(x.shape[0] if q_check is None else q_check)) if samples is None else list(samples)
    rowptr, col, val = _host_csr(op)
    K, g = layer.weight.shape[0], layer.weight.shape[-1]
    W = layer.weight.detach().reshape(K, -1, g).cpu().numpy()
    b = layer.bias.detach().reshape(-1).cpu().numpy()
    xs = np.stack([x[i].reshape(op.n, -1).cpu().numpy() for i in sel])
    ref = c_port.forward(0, rowptr, col, val, xs, W, b, kind)
    got = np.stack([out[i].cpu().numpy() for i in sel])
    assert rel_err(got, ref) <= TOL
    return out


def _mnist_grid(device):
    from tgcn_amd.graph import GraphOperand
    z = np.load(os.path.join(GOLDEN, "GCNCheb_grid784_q3_f1_g8_K5_x2d.npz"))    # the reference's own 28x28 8-NN grid, rescale_L(laplacian)
    n = int(z["n"])
    rowptr = torch.as_tensor(z["rowptr"])
    row = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
    return GraphOperand.from_coo(n, row.to(device), torch.as_tensor(z["col"]).long().to(device), torch.as_tensor(z["val"]).to(device))


@pytest.mark.parametrize("f", [1, 64])
def test_cfg2_mnist_gcncheb_batch128(f, gpu_device):
    """configs[1]: MNIST 8-NN graph (784 vertices, 6396 entries), K=5, C=64, batch 128 -- GCNCheb(L, f, 64, 5), f = 1 (first layer)
    and f = 64 (SURVEY.md 8d asks for both)"""
    import tgcn_amd
    op = _mnist_grid(gpu_device)
    torch.manual_seed(1)
    layer = tgcn_amd.GCNCheb(op, f, 64, 5).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((128, 784) if f == 1 else (128, 784, f), device="cuda", generator=g)
    if f > 1:
        _check(layer, op, x, 1)
    else:           # 2-D input gains its channel axis inside the module (gcn.py:216-217)
        from oracle import c_port
        with torch.no_grad():
            out = layer(x)
        rowptr, col, val = _host_csr(op)
        ref = c_port.forward(0, rowptr, col, val, x.unsqueeze(-1).cpu().numpy(), layer.weight.detach().cpu().numpy(),
                             layer.bias.detach().reshape(-1).cpu().numpy(), 1)
        assert rel_err(out.cpu().numpy(), ref) <= TOL


def test_cfg3_tgcn_mnist_temporal_batch64(gpu_device):
    """configs[2]: 784 vertices x T=28 steps, K=5, C=64, batch 64 -- TGCNCheb_H(L, 1, 64, 5, 28)"""
    import tgcn_amd
    op = _mnist_grid(gpu_device)
    torch.manual_seed(1)
    layer = tgcn_amd.TGCNCheb_H(op, 1, 64, 5, 28).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((64, 784, 28), device="cuda", generator=g)
    _check(layer, op, x, 2)


def test_cfg4_hcp_style_mesh_T1200(gpu_device):
    """configs[3]: HCP-style synthetic brain graph, 90 k vertices / ~1 M entries, T=1200, K=5, C=32 --
    TGCNCheb_H(L, 1, 32, 5, 1200), q = 1 on tools/synth.sheet_mesh(300) (project-first path, bf16x3 projection)"""
    import tgcn_amd
    from tools import synth
    n, row, col, val = synth.sheet_mesh(300, device=gpu_device)
    op = tgcn_amd.GraphOperand.from_coo(n, row, col, val, gpu_device)
    assert op.n == 90000 and 900_000 <= op.nnz <= 1_100_000
    torch.manual_seed(1)
    layer = tgcn_amd.TGCNCheb_H(op, 1, 32, 5, 1200).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((1, n, 1200), device="cuda", generator=g)
    out = _check(layer, op, x, 2)
    # and against fp64 scipy (the hops-first evaluation order of the reference), same bar
    import scipy.sparse as sp
    rowptr, colh, valh = _host_csr(op)
    L = sp.csr_matrix((valh.astype(np.float64), colh, rowptr), shape=(n, n))
    X = x[0].double().cpu().numpy()
    W = layer.weight.detach().double().cpu().numpy().reshape(5, 1200, 32)
    terms = [X, L @ X]
    P = terms[1]
    for k in range(2, 5):
        P = L @ P
        terms.append(2 * P - terms[k - 2])
    ref = sum(t @ W[k] for k, t in enumerate(terms)) + layer.bias.detach().double().cpu().numpy()[0]
    assert rel_err(out[0].cpu().numpy(), ref) <= TOL


@pytest.mark.parametrize("labeling", ["random", "degree"])
def test_cfg5_reduced_rmat_tgcncheb(labeling, gpu_device):
    """configs[4] at 1/10 scale: R-MAT (0.57, 0.19, 0.19, 0.05), 1 M vertices / 16 M entries, TGCNCheb(L, 64, 64, K=5), 3 time
    steps, random (worst case) and degree-sorted labels -- the general path the 160 M-entry bench runs: row blocks, column-ordered
    segments and the fix-up for the hops, bf16x3 projection, per-vertex bias"""
    import tgcn_amd
    from tools import synth
    n, nnz = 1_000_000, 16_000_000
    _, row, col, val = synth.rmat(n, nnz, seed=12345, labeling=labeling, device=gpu_device)
    op = tgcn_amd.GraphOperand.from_coo(n, row, col, val, gpu_device)
    del row, col, val
    assert op.nnz == nnz
    sched = op.schedule_for(64, True)
    assert sched.nseg > 0 and sched.nlong > 0 and sched.nhuge > 0          # every long-row path is exercised
    torch.manual_seed(1)
    layer = tgcn_amd.TGCNCheb(op, 64, 64, 5).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((3, n, 64), device="cuda", generator=g)
    _check(layer, op, x, 2, q_check=2)


@pytest.mark.parametrize("labeling", ["random", "degree"])
def test_cfg5_full_size_one_time_step(labeling, gpu_device):
    """configs[4] AT FULL SIZE -- the workload BENCH times: R-MAT 10 M vertices / 160 M entries, both labelings SURVEY 8(d) names (random =
    the headline, degree-sorted = the "friendly" one), TGCNCheb(L, 64, 64, K=5) -- TWO of its 16 time steps through the module in one call,
    the SECOND one (it sits behind the first's 640 M floats of x and out, and shares the pass with it: time steps per pass 2) against
    oracle/cheb_ref.c (about 12 s on the box's host cores).  Asserts that it is
    the compacted path (5.27 M structurally empty rows -> compact hop tensors, row-mapped projections) with hop outputs beyond the
    Infinity Cache, i.e. the streaming form of hop_kernel with whole-row wave segments, lane-group segments and the fix-up: what
    bench.py's own check covers, under pytest.  (> 2^31-element offsets: tests/test_compact_wave.py::test_offsets_beyond_2_31_elements.)"""
    import tgcn_amd
    from tgcn_amd import functional as F
    from tools import synth
    n, nnz = 10_000_000, 160_000_000
    _, row, col, val = synth.rmat(n, nnz, seed=12345, labeling=labeling, device=gpu_device)
    op = tgcn_amd.GraphOperand.from_coo(n, row, col, val, gpu_device)
    del row, col, val
    assert op.nnz == nnz
    plan = op.compact_plan()
    assert plan is not None and plan.n_empty > 5_000_000 and plan.n_c + plan.n_empty == n
    assert F.COMPACT and F.choose_layout(1, n, 64) == 0                      # layer_forward takes cheb_forward_compact for this shape
    assert plan.n_c * 64 * 4 > (256 << 20)                                   # a hop's output exceeds the Infinity Cache: tgcn_csr_hop_f32 streams
    sched = plan.schedule_for(64, True)
    assert sched.nwseg > 0 and sched.nseg > sched.nwseg and sched.nlong > 0 and sched.nhuge > 0 and sched.npartial > 0
    torch.manual_seed(1)
    layer = tgcn_amd.TGCNCheb(op, 64, 64, 5).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((2, n, 64), device="cuda", generator=g)
    calls = []
    real = F.cheb_forward_compact
    F.cheb_forward_compact = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        _check(layer, op, x, 2, samples=[1])
    finally:
        F.cheb_forward_compact = real
    assert calls, "the module forward did not take the compacted driver"
