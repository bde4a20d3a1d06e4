"""SURVEY.md 8f rows against the oracle: the fused bias + relu + pool epilogue (f2) and the operand builders / vertex
reordering on the device (f4)."""
import numpy as np
import pytest
import torch

from conftest import golden_files, golden_ids, load_golden, rel_err
from oracle import cheb_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _sym_graph(n, avg, rng):
    m = n * avg // 2
    u, v = rng.integers(0, n, m), rng.integers(0, n, m)
    keep = u != v
    u, v = u[keep], v[keep]
    row, col = np.concatenate([u, v]), np.concatenate([v, u])
    key = np.unique(row.astype(np.int64) * n + col)
    return key // n, key % n


def _dense_autograd_reference(L64, x, W, b, mode, pool, gz):
    """fp64 torch autograd of  pool(relu(sum_k T_k(L) x W_k + b)): the caller pattern of
    examples/pytorch_based/pytorch_hcp_tgcn.py:134-141 written with dense tensors"""
    Ld = torch.tensor(L64.toarray(), dtype=torch.float64)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    K = W.shape[0]
    terms = [xt]
    P = xt
    for k in range(1, K):
        if mode == "power":
            P = torch.einsum("nm,qmc->qnc", Ld, P)
            terms.append(P if k == 1 else 2 * P - terms[k - 2])
        else:
            terms.append(torch.einsum("nm,qmc->qnc", Ld, terms[k - 1]) if k == 1 else 2 * torch.einsum("nm,qmc->qnc", Ld, terms[k - 1]) - terms[k - 2])
    y = sum(t @ Wt[k] for k, t in enumerate(terms)) + bt
    y = torch.relu(y)
    q, n, g = y.shape
    z = y.reshape(q, n // pool, pool, g).max(dim=2)[0]
    z.backward(torch.tensor(gz, dtype=torch.float64))
    return z.detach().numpy(), xt.grad.numpy(), Wt.grad.numpy(), bt.grad.numpy()


@pytest.mark.parametrize("cls,n,pool,C,g,K", [("TGCNCheb_H", 784, 4, 6, 12, 4), ("GCNCheb", 400, 2, 3, 10, 5), ("ChebConv", 5000, 4, 2, 9, 4),
                                              ("GCNCheb", 2048, 4, 64, 32, 3), ("TGCNCheb_H", 148, 4, 15, 32, 6), ("TGCNCheb", 96, 2, 8, 8, 3)])
def test_fused_relu_pool_vs_oracle(cls, n, pool, C, g, K, gpu_device):
    """cheb_relu_pool(layer, x) against the ORACLE's pool(relu(forward)) -- not against the unfused HIP path -- and its
    gradients against fp64 dense autograd (one-launch fused kernels, the matrix-pipe path of dense operands, and the
    layer + relu/pool pass of the larger graphs)."""
    import tgcn_amd
    rng = np.random.default_rng(n + K)
    row, col = _sym_graph(n, 120 if n in (148, 96) else 6, rng)
    ei = _dev(np.stack([row, col]).astype(np.int64))
    q = 3
    torch.manual_seed(1)
    if cls == "ChebConv":
        layer, extra, mode = tgcn_amd.ChebConv(C, g, K).cuda(), (ei,), "chebyshev"
        r2, c2, lap = O.edge_laplacian(np.stack([row, col]), None, n, np.float32)
        L = O.coo_to_csr(r2, c2, lap, n)
        fwd = lambda xx, W, b: O.cheb_conv_forward(xx, np.stack([row, col]), None, W, b)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
    else:
        deg = np.bincount(row, minlength=n).astype(np.float64)
        dis = np.where(deg > 0, 1 / np.sqrt(np.maximum(deg, 1)), 0)
        val = (-dis[row] * dis[col]).astype(np.float32)
        L = O.coo_to_csr(row, col, val, n)
        op = tgcn_amd.GraphOperand.from_coo(n, _dev(row), _dev(col), _dev(val))
        extra, mode = (), "power"
        if cls == "TGCNCheb_H":
            layer = tgcn_amd.TGCNCheb_H(op, 1, g, K, C).cuda()
            fwd = lambda xx, W, b: O.tgcn_cheb_h_forward(L, xx, W, b)
        elif cls == "TGCNCheb":
            layer = tgcn_amd.TGCNCheb(op, C, g, K).cuda()
            fwd = lambda xx, W, b: O.tgcn_cheb_forward(L, xx, W, b)
        else:
            layer = tgcn_amd.GCNCheb(op, C, g, K).cuda()
            fwd = lambda xx, W, b: O.gcn_cheb_forward(L, xx, W, b)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
    W = layer.weight.detach().cpu().numpy()
    b = layer.bias.detach().cpu().numpy()
    ref_z = O.gcn_pool(np.maximum(fwd(x, W, b), 0), pool)
    xt = _dev(x).requires_grad_(True)
    z = tgcn_amd.cheb_relu_pool(layer, xt, *extra, pool=pool)
    assert rel_err(z.detach().cpu().numpy(), ref_z) <= TOL
    gz = rng.standard_normal(ref_z.shape).astype(np.float32)
    z.backward(_dev(gz))
    W3 = W.reshape(K, -1, g)
    bb = b.reshape(1, -1, g) if cls in ("TGCNCheb", "TGCNCheb_H") else b.reshape(1, 1, g)
    z64, gx, gW, gb = _dense_autograd_reference(L.astype(np.float64), x.reshape(q, n, -1), W3, bb, mode, pool, gz)
    assert rel_err(z.detach().cpu().numpy(), z64) <= TOL
    assert rel_err(xt.grad.cpu().numpy().reshape(gx.shape), gx) <= 2e-5
    assert rel_err(layer.weight.grad.cpu().numpy().reshape(gW.shape), gW) <= 2e-5
    assert rel_err(layer.bias.grad.cpu().numpy().reshape(gb.shape), gb) <= 2e-5


@pytest.mark.parametrize("path", golden_files("operand_"), ids=golden_ids(golden_files("operand_")))
def test_operand_builder_on_device_golden(path, gpu_device):
    """rescale_L(laplacian(A, normalized=True), lmax) built on the GPU (GraphOperand.from_adjacency) against the reference's
    own output (gcn/graph.py:117-136, 232-238)"""
    from tgcn_amd.graph import GraphOperand
    g = load_golden(path)
    n = int(g["n"])
    ref = O.csr_from_arrays(n, g["L_rowptr"], g["L_col"], g["L_val"])
    op = GraphOperand.from_adjacency(n, _dev(g["a_row"]), _dev(g["a_col"]), _dev(g["a_val"]), lmax=float(g["lmax"]))
    assert op.device.type == "cuda"
    assert abs(op.to_scipy() - ref).max() <= 1e-6
    # and the layer on it equals the oracle on the reference's operand
    from tgcn_amd import functional as F
    x = np.random.default_rng(0).standard_normal((2, n, 8)).astype(np.float32)
    assert rel_err(F.csr_hop(op, _dev(x)).cpu().numpy(), O._apply(ref, x)) <= TOL


@pytest.mark.filterwarnings("ignore:GraphOperand.reordered")
@pytest.mark.parametrize("kind", ["hub_first", "degree_sorted", "rcm"])
@pytest.mark.parametrize("cls", ["TGCNCheb", "GCNCheb", "GCNCheb_small"])
def test_reordered_operand_is_permutation_invariant(kind, cls, gpu_device):
    """GraphOperand.reordered(kind): same outputs and gradients in the caller's labels (per-vertex bias included), against
    the oracle on the original operand; the stack methods too."""
    import tgcn_amd
    rng = np.random.default_rng(11)
    n = 300 if cls == "GCNCheb_small" else 6000
    row, col = _sym_graph(n, 8, rng)
    val = (rng.standard_normal(row.shape[0]) / 4).astype(np.float32)
    L = O.coo_to_csr(row, col, val, n)
    op = tgcn_amd.GraphOperand.from_coo(n, _dev(row), _dev(col), _dev(val))
    opr = op.reordered(kind)
    assert opr.perm is not None and torch.equal(torch.sort(opr.perm)[0], torch.arange(n, device="cuda"))
    assert abs(opr.to_scipy() - L[opr.perm.cpu().numpy()][:, opr.perm.cpu().numpy()]).max() <= 1e-7
    torch.manual_seed(2)
    if cls == "TGCNCheb":
        mk = lambda o: tgcn_amd.TGCNCheb(o, 8, 12, 4).cuda()
        ref = lambda x, W, b: O.tgcn_cheb_forward(L, x, W, b)
        C = 8
    else:
        mk = lambda o: tgcn_amd.GCNCheb(o, 5, 7, 5).cuda()
        ref = lambda x, W, b: O.gcn_cheb_forward(L, x, W, b)
        C = 5
    la, lb = mk(op), mk(opr)
    lb.load_state_dict(la.state_dict())
    x = rng.standard_normal((2, n, C)).astype(np.float32)
    want = ref(x, la.weight.detach().cpu().numpy(), la.bias.detach().cpu().numpy())
    xa, xb = _dev(x).requires_grad_(True), _dev(x).requires_grad_(True)
    oa, ob = la(xa), lb(xb)
    assert rel_err(ob.detach().cpu().numpy(), want) <= TOL
    go = torch.randn_like(oa)
    oa.backward(go)
    ob.backward(go)
    assert rel_err(xb.grad.cpu().numpy(), xa.grad.cpu().numpy()) <= 2e-5
    assert rel_err(lb.weight.grad.cpu().numpy(), la.weight.grad.cpu().numpy()) <= 2e-5
    assert rel_err(lb.bias.grad.cpu().numpy(), la.bias.grad.cpu().numpy()) <= 2e-5
    stack = lb._time_chebyshev(_dev(x)) if cls == "TGCNCheb" else lb._chebyshev(_dev(x))
    assert rel_err(stack.cpu().numpy(), O.stack_reference_power(L, x, la.filter_order)) <= TOL
    z = tgcn_amd.cheb_relu_pool(lb, _dev(x), pool=2)
    assert rel_err(z.cpu().detach().numpy(), O.gcn_pool(np.maximum(want, 0), 2)) <= TOL


@pytest.mark.filterwarnings("ignore:GraphOperand.reordered")
@pytest.mark.parametrize("kind", ["degree", "rcm"])
def test_reordered_operand_in_forward_series_and_to(kind, gpu_device):
    """a reordered operand must stay one through forward_series (hops of the streaming-window layer, both directions)
    and through GraphOperand.to(): same outputs and gradients in the caller's labels as the plain operand."""
    import tgcn_amd
    rng = np.random.default_rng(5)
    n, S, T, H, g, K = 700, 2, 24, 6, 5, 4
    row, col = _sym_graph(n, 6, rng)
    val = (rng.standard_normal(row.shape[0]) / 4).astype(np.float32)
    op = tgcn_amd.GraphOperand.from_coo(n, _dev(row), _dev(col), _dev(val))
    opr = op.reordered(kind)
    moved = opr.to("cuda:0")
    assert moved.perm is not None and torch.equal(moved.perm, opr.perm) and torch.equal(moved.inv_perm, opr.inv_perm)
    assert abs(moved.to_scipy() - opr.to_scipy()).max() == 0
    torch.manual_seed(3)
    la = tgcn_amd.TGCNCheb_H(op, 1, g, K, H).cuda()
    lb = tgcn_amd.TGCNCheb_H(moved, 1, g, K, H).cuda()
    lb.load_state_dict(la.state_dict())
    series = rng.standard_normal((S, n, T)).astype(np.float32)
    sa, sb = _dev(series).requires_grad_(True), _dev(series).requires_grad_(True)
    oa, ob = la.forward_series(sa), lb.forward_series(sb)
    assert rel_err(ob.detach().cpu().numpy(), oa.detach().cpu().numpy()) <= TOL
    go = torch.randn_like(oa)
    oa.backward(go)
    ob.backward(go)
    assert rel_err(sb.grad.cpu().numpy(), sa.grad.cpu().numpy()) <= 2e-5
    assert rel_err(lb.weight.grad.cpu().numpy(), la.weight.grad.cpu().numpy()) <= 2e-5
    assert rel_err(lb.bias.grad.cpu().numpy(), la.bias.grad.cpu().numpy()) <= 2e-5


@pytest.mark.parametrize("case", ["mesh59k_gcn32x64_pool4", "mesh90k_tgcn64x64_pool2", "grid_exact_pool4", "mesh_layout1_fallback",
                                  "mesh_k1_wide", "mesh59k_gcn64x128_pool4_wide", "mesh_tgcn32x160_pool2_wide", "mesh_wide_pool8_fallback"])
def test_relu_pool_in_the_projection_epilogue(case, gpu_device):
    """SURVEY 8f-2 for graphs that do not fit in LDS: bias + relu + max over 2 / 4 consecutive vertices inside the projection's
    epilogue (tgcn_cheb_forward_pool_f32) -- against the ORACLE's pool(relu(forward)) on the 59,536-vertex mesh and at 90 k
    vertices, bitwise against layer + separate relu/pool pass (values and arg-max bytes), gradients against the unfused
    modules.  The last case is a shape the epilogue does not cover (vertex-major rows): same entry point, scratch + extra pass."""
    import tgcn_amd
    from tgcn_amd import functional as F, _lib
    from tools import synth
    rng = np.random.default_rng(3)
    if case == "mesh59k_gcn32x64_pool4":          # second layer of the reference's HCP model on the cortical-mesh size
        n, row, col, val = synth.sheet_mesh(244)
        mk, q, C, pool, fused = (lambda o: tgcn_amd.GCNCheb(o, 32, 64, 5)), 2, 32, 4, True
        ref = lambda L, x, W, b: O.gcn_cheb_forward(L, x, W, b)
    elif case == "mesh90k_tgcn64x64_pool2":
        n, row, col, val = synth.sheet_mesh(300)
        mk, q, C, pool, fused = (lambda o: tgcn_amd.TGCNCheb(o, 64, 64, 4)), 1, 64, 2, True
        ref = lambda L, x, W, b: O.tgcn_cheb_forward(L, x, W, b)
    elif case == "grid_exact_pool4":              # small problem: the exact-fp32 W-resident kernel's epilogue
        n, row, col, val = synth.sheet_mesh(40)
        mk, q, C, pool, fused = (lambda o: tgcn_amd.GCNCheb(o, 32, 16, 3)), 2, 32, 4, True
        ref = lambda L, x, W, b: O.gcn_cheb_forward(L, x, W, b)
    elif case == "mesh_k1_wide":                  # K = 1, 96 output columns: since round 6 the wide bf16x3 kernel folds groups of 2 / 4 rows in registers
        n, row, col, val = synth.sheet_mesh(100)
        mk, q, C, pool, fused = (lambda o: tgcn_amd.GCNCheb(o, 64, 96, 1)), 1, 64, 4, True
        ref = lambda L, x, W, b: O.gcn_cheb_forward(L, x, W, b)
    elif case == "mesh59k_gcn64x128_pool4_wide":  # >= 96 output columns (project_x3v2_kernel), per-channel bias, ragged last tile
        n, row, col, val = synth.sheet_mesh(244)
        mk, q, C, pool, fused = (lambda o: tgcn_amd.GCNCheb(o, 64, 128, 3)), 2, 64, 4, True
        ref = lambda L, x, W, b: O.gcn_cheb_forward(L, x, W, b)
    elif case == "mesh_tgcn32x160_pool2_wide":    # ten column tiles, per-vertex bias, groups of two, two samples in one launch
        n, row, col, val = synth.sheet_mesh(150)
        mk, q, C, pool, fused = (lambda o: tgcn_amd.TGCNCheb(o, 32, 160, 3)), 2, 32, 2, True
        ref = lambda L, x, W, b: O.tgcn_cheb_forward(L, x, W, b)
    elif case == "mesh_wide_pool8_fallback":      # K = 1 on a schedule without partial rows has a base workspace of 0 bytes, and groups of 8
        n, row, col, val = synth.sheet_mesh(100)  # rows do not fit a lane's four accumulators: the query must still size the output scratch
        mk, q, C, pool, fused = (lambda o: tgcn_amd.GCNCheb(o, 64, 96, 1)), 1, 64, 8, False
        ref = lambda L, x, W, b: O.gcn_cheb_forward(L, x, W, b)
    else:
        n, row, col, val = synth.sheet_mesh(60)
        mk, q, C, pool, fused = (lambda o: tgcn_amd.GCNCheb(o, 8, 16, 3)), 3, 8, 4, False
        ref = lambda L, x, W, b: O.gcn_cheb_forward(L, x, W, b)
    assert n % pool == 0
    op = tgcn_amd.GraphOperand.from_coo(n, row, col, val)
    torch.manual_seed(4)
    layer = mk(op).cuda()
    K = layer.filter_order
    N = layer.out_channels
    assert F.small_path_tile(op, C, F.MODE_POWER) == 0
    assert F.pool_epilogue_is_fused(op, q, C, N, K, pool) == fused
    x = rng.standard_normal((q, n, C)).astype(np.float32)
    L = op.to_scipy()
    want = O.gcn_pool(np.maximum(ref(L, x, layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()), 0), pool)
    x1 = _dev(x).requires_grad_(True)
    z1 = tgcn_amd.cheb_relu_pool(layer, x1, pool=pool)
    assert rel_err(z1.detach().cpu().numpy(), want) <= TOL
    # the same kernels without the fusion: layer, then the relu + pool pass -- identical values and arg-max bytes
    with torch.no_grad():
        # bit for bit needs the SAME projection kernel on both sides: the fused epilogue lives in the tiled kernels, while the unfused layer
        # of the large shapes takes the streaming bf16x3 kernel (round 5; same arithmetic, another order inside one MFMA)
        big = n * q >= 32768 and C in (32, 64) and N <= 64
        if big:
            y_stream = layer(_dev(x))
            _lib.check(_lib.lib().tgcn_set_tuning(b"project_variant", 3))
        y = layer(_dev(x))
        if big:         # (the switch stays on for the gradient comparison below: a 1e-7 difference flips arg-max ties; conftest resets it)
            assert rel_err(y_stream.cpu().numpy(), y.cpu().numpy()) <= 2e-6          # the streaming kernel against the tiled one
        z2 = torch.empty_like(z1)
        i2 = torch.empty(z1.shape, dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().tgcn_relu_pool_f32(_lib.stream_ptr(), _lib.ptr(y), _lib.ptr(z2), _lib.ptr(i2), q, n, N, pool))
    assert torch.equal(z1.detach(), z2)
    # gradients against the unfused modules
    gz = torch.randn_like(z1)
    z1.backward(gz)
    g1 = [x1.grad.clone(), layer.weight.grad.clone(), layer.bias.grad.clone()]
    layer.zero_grad()
    x2 = _dev(x).requires_grad_(True)
    yy = torch.relu(layer(x2))
    (tgcn_amd.gcn_pool_4(yy) if pool == 4 else (tgcn_amd.gcn_pool(yy) if pool == 2 else F.PoolMaxFn.apply(yy, pool))).backward(gz)
    for a, b in zip(g1, [x2.grad, layer.weight.grad, layer.bias.grad]):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 2e-5
