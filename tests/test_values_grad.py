"""Gradients w.r.t. the VALUES of the sparse operand (round 4): `edge_weight` of ChebConv / ChebTimeConv and `value` of spmm* are differentiable in
the reference (gather / scale / scatter_add, tgcn/nn/gcn.py:296-308; lap = -deg[row] * edge_weight * deg[col], :413, :510).  Here the operand is
packed outside autograd and the gradient is a sampled dense-dense product over the stored pattern (tgcn_csr_sddmm_f32) behind the Clenshaw adjoints
of the recurrence.  Checked against float64 torch autograd of a dense restatement of the reference's formulas.  Tolerance 2e-5 (gradients)."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
GRAD_TOL = 2e-5


def _edges(n, E, rng, self_loops=True, duplicates=True):
    ei = rng.integers(0, n, (2, E))
    if self_loops:
        ei[1, : E // 8] = ei[0, : E // 8]
    if duplicates:
        ei[:, E // 2: E // 2 + E // 10] = ei[:, : E // 10]
    return ei


def _dense_lap(ei, w, n):
    """the reference's operand as a dense float64 matrix, differentiable in w (gcn.py:398-413): self loops removed, unweighted source degree"""
    row, col = ei[0], ei[1]
    keep = row != col
    r, c, ww = row[keep], col[keep], w[keep]
    deg = torch.bincount(r, minlength=n).double()
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0
    lap = -dis[r] * ww * dis[c]
    return torch.zeros(n, n, dtype=torch.float64, device=w.device).index_put((r, c), lap, accumulate=True)


@pytest.mark.parametrize("n,E,q,f,g,K,H", [(60, 400, 3, 4, 5, 3, 0), (60, 400, 2, 1, 8, 1, 0), (60, 400, 2, 3, 4, 2, 0), (3000, 30000, 2, 16, 8, 5, 0),
                                           (148, 3000, 2, 1, 6, 4, 5), (3000, 30000, 1, 2, 4, 3, 6), (300, 2400, 2, 3, 5, 10, 0)])
def test_edge_weight_gradient_of_the_chebyshev_layers(n, E, q, f, g, K, H, gpu_device):
    import tgcn_amd
    rng = np.random.default_rng(n + K + H)
    ei = torch.as_tensor(_edges(n, E, rng)).cuda()
    w0 = torch.as_tensor(rng.uniform(0.5, 1.5, E).astype(np.float32)).cuda()
    torch.manual_seed(0)
    timed = H > 0
    layer = (tgcn_amd.ChebTimeConv(f, g, K, H) if timed else tgcn_amd.ChebConv(f, g, K)).cuda()
    x0 = torch.as_tensor(rng.standard_normal((q, n, H, f) if timed else (q, n, f)).astype(np.float32)).cuda()
    gout = torch.as_tensor(rng.standard_normal((q, n, g)).astype(np.float32)).cuda()
    # --- the HIP path
    w = w0.clone().requires_grad_(True)
    x = x0.clone().requires_grad_(True)
    out = layer(x, ei, w)
    out.backward(gout)
    got = [t.detach().cpu().numpy() for t in (out, w.grad, x.grad, layer.weight.grad, layer.bias.grad)]
    # --- float64 autograd of the reference's formulas on a dense operand
    wd = w0.double().requires_grad_(True)
    xd = x0.double().requires_grad_(True)
    Wd = layer.weight.detach().double().requires_grad_(True)
    bd = layer.bias.detach().double().requires_grad_(True)
    Ld = _dense_lap(ei, wd, n)
    x3 = xd.reshape(q, n, -1)
    Wk = Wd.reshape(K, -1, g)
    T = [x3]
    if K > 1:
        T.append(torch.einsum("nm,qmc->qnc", Ld, x3))
    for k in range(2, K):
        T.append(2 * torch.einsum("nm,qmc->qnc", Ld, T[k - 1]) - T[k - 2])
    ref = sum(T[k] @ Wk[k] for k in range(K)) + bd
    ref.backward(gout.double())
    wgrad = wd.grad if wd.grad is not None else torch.zeros_like(wd)          # K = 1: the operand does not enter
    want = [t.detach().cpu().numpy() for t in (ref, wgrad, xd.grad, Wd.grad.reshape(layer.weight.shape), bd.grad)]
    assert rel_err(got[0], want[0]) <= 1e-5
    for a, b, name in zip(got[1:], want[1:], ("edge_weight", "x", "weight", "bias")):
        if K == 1 and name == "edge_weight":
            assert np.abs(a).max() == 0 and np.abs(b).max() == 0          # no hop: the weights do not enter
            continue
        assert rel_err(a, b) <= GRAD_TOL, name
    # self loops get no gradient (they are removed, gcn.py:398); run-to-run determinism of the sampled product
    loops = (ei[0] == ei[1]).cpu().numpy()
    assert np.abs(got[1][loops]).max() == 0
    layer.zero_grad()
    w2 = w0.clone().requires_grad_(True)
    layer(x0, ei, w2).backward(gout)
    assert np.array_equal(w2.grad.cpu().numpy(), got[1])


@pytest.mark.parametrize("fn,shape", [("spmm", (500, 7)), ("spmm_batch_2", (3, 500)), ("spmm_batch_3", (3, 500, 4, 5))])
def test_spmm_is_differentiable_in_value_and_matrix(fn, shape, gpu_device):
    import tgcn_amd
    rng = np.random.default_rng(11)
    n_cols, m, E = 500, 420, 6000                      # rectangular: m result rows from n_cols source rows (gcn.py:296-308 takes any)
    idx = torch.as_tensor(np.stack([rng.integers(0, m, E), rng.integers(0, n_cols, E)])).cuda()
    idx[:, E // 2: E // 2 + 300] = idx[:, :300]        # duplicates of one (row, col): separate terms
    v0 = torch.as_tensor(rng.standard_normal(E).astype(np.float32)).cuda()
    M0 = torch.as_tensor(rng.standard_normal(shape).astype(np.float32)).cuda()
    v = v0.clone().requires_grad_(True)
    M = M0.clone().requires_grad_(True)
    out = getattr(tgcn_amd, fn)(idx, v, m, M)
    gout = torch.as_tensor(rng.standard_normal(tuple(out.shape)).astype(np.float32)).cuda()
    out.backward(gout)
    vd = v0.double().requires_grad_(True)
    Md = M0.double().requires_grad_(True)
    A = torch.zeros(m, n_cols, dtype=torch.float64, device="cuda").index_put((idx[0], idx[1]), vd, accumulate=True)
    if fn == "spmm":
        ref = A @ Md
    else:
        Mx = Md.unsqueeze(-1) if fn == "spmm_batch_2" else Md
        ref = torch.einsum("nm,qm...->qn...", A, Mx)
    ref.backward(gout.double().reshape(ref.shape))
    assert rel_err(out.detach().cpu().numpy().reshape(ref.shape), ref.detach().cpu().numpy()) <= 1e-5
    assert rel_err(v.grad.cpu().numpy(), vd.grad.cpu().numpy()) <= GRAD_TOL
    assert rel_err(M.grad.cpu().numpy(), Md.grad.cpu().numpy()) <= GRAD_TOL


@pytest.mark.parametrize("n,E,f,g,K,small", [(60, 400, 4, 5, 3, True), (3000, 30000, 16, 8, 4, False), (70000, 200000, 8, 8, 3, False),
                                             (32, 900, 4, 5, 3, True)])          # the last: an operand that also keeps a DENSE copy (matrix-pipe kernels), refreshed too
def test_training_with_learnable_edge_weights_keeps_one_operand(n, E, f, g, K, small, gpu_device, monkeypatch):
    """every optimizer step bumps edge_weight's version; the operand cache used to key on it -- a full edge normalise +
    COO -> CSR sort + schedule per step and up to 16 stale operands (each with its lazily built transpose and compact plans) kept alive.  Now
    the operand is cached by PATTERN and its packed values are refreshed in place.  Several SGD steps: the cache holds one operand + the
    link map, the operand object (and its schedules, its transpose) are the same objects throughout, and every step's output and
    gradients equal those of a FRESH module that has never seen another weight."""
    import copy
    import tgcn_amd
    from tgcn_amd import functional as F, graph
    if not small:
        monkeypatch.setattr(F, "SMALL_PATH", False)
    monkeypatch.setattr(graph, "COMPACT_MIN_ROWS", 50000)           # the 70 k case: half of its vertices are isolated -> compact plans in play
    rng = np.random.default_rng(n + K)
    ei_np = _edges(n if n < 70000 else n // 2, E, rng)              # 70 k: edges among the first half only
    ei = torch.as_tensor(ei_np).cuda()
    w = torch.nn.Parameter(torch.as_tensor(rng.uniform(0.5, 1.5, E).astype(np.float32)).cuda())
    torch.manual_seed(3)
    layer = tgcn_amd.ChebConv(f, g, K).cuda()
    opt = torch.optim.SGD([w] + list(layer.parameters()), lr=0.05)
    x = torch.as_tensor(rng.standard_normal((2, n, f)).astype(np.float32)).cuda()
    gout = torch.as_tensor(rng.standard_normal((2, n, g)).astype(np.float32)).cuda()
    first_op = None
    for step in range(5):
        opt.zero_grad()
        xin = x.clone().requires_grad_(True)
        out = layer(xin, ei, w)
        (out - gout).square().mean().mul(50.0).backward()              # a bounded objective: the parameters stay finite over the steps
        ops = [v[0] for k, v in layer._ops._d.items() if isinstance(v[0], graph.GraphOperand)]
        assert len(ops) == 1 and len(layer._ops._d) <= 2, list(layer._ops._d)          # one operand + the (src, coef) links
        first_op = first_op or ops[0]
        assert ops[0] is first_op and ops[0]._transpose is not None                     # same object: schedules / transpose built once
        assert (ops[0].dense is not None) == (n == 32)
        # a fresh module with the current parameters and a weight tensor that was never seen before
        fresh = copy.deepcopy(layer)
        w2 = w.detach().clone().requires_grad_(True)
        x2 = x.clone().requires_grad_(True)
        out2 = fresh(x2, ei.clone(), w2)
        (out2 - gout).square().mean().mul(50.0).backward()
        assert rel_err(out.detach().cpu().numpy(), out2.detach().cpu().numpy()) <= 2e-6
        for a, b in ((w.grad, w2.grad), (xin.grad, x2.grad), (layer.weight.grad, fresh.weight.grad)):
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= GRAD_TOL
        opt.step()
    # a COMPUTED weight (new tensor every forward, as from a small network): still one operand -- and never stale: each forward's weight is freed
    # before the next one is made, so the allocator hands the SAME address (version 0 again) to a tensor with other values
    scale = torch.nn.Parameter(torch.ones((), device="cuda"))
    seen = set()
    for it in range(4):
        wc = w.detach() * (scale * (1.0 + 0.25 * it))
        seen.add(wc.data_ptr())
        out = layer(x, ei, wc)
        out.sum().backward()
        with torch.no_grad():
            want = copy.deepcopy(layer)(x, ei.clone(), (w.detach() * (1.0 + 0.25 * it)).clone())
        assert rel_err(out.detach().cpu().numpy(), want.cpu().numpy()) <= 2e-6, it
        del wc, out
    assert len([v for v in layer._ops._d.values() if isinstance(v[0], graph.GraphOperand)]) == 1
    assert scale.grad is not None and torch.isfinite(scale.grad)


def test_spmm_with_a_learnable_value_keeps_one_operand(gpu_device):
    import tgcn_amd
    from tgcn_amd import nn as tnn, graph
    rng = np.random.default_rng(9)
    n, E = 500, 4000
    idx = torch.as_tensor(rng.integers(0, n, (2, E))).cuda()
    val = torch.nn.Parameter(torch.as_tensor(rng.standard_normal(E).astype(np.float32)).cuda())
    m = torch.as_tensor(rng.standard_normal((3, n, 6)).astype(np.float32)).cuda()
    before = len(tnn._spmm_ops._d)
    opt = torch.optim.SGD([val], lr=0.1)
    for _ in range(4):
        opt.zero_grad()
        out = tgcn_amd.spmm_batch_2(idx, val, n, m)
        ref = torch.zeros(3, n, 6, dtype=torch.float64, device="cuda").index_add_(1, idx[0], m.double()[:, idx[1]] * val.detach().double().view(1, E, 1))
        assert rel_err(out.detach().cpu().numpy(), ref.cpu().numpy()) <= 1e-5
        out.square().sum().backward()
        opt.step()
    assert len(tnn._spmm_ops._d) - before <= 2          # the operand and the entry order, however many steps


@pytest.mark.parametrize("small", [True, False])
def test_two_learnable_weights_on_one_pattern_backward_after_both_forwards(small, gpu_device, monkeypatch):
    """the operand of a pattern holds ONE set of packed values, refreshed in place; a backward reads them.  Two forwards with
    two different learnable weights on the same edge_index / index, both backwards AFTER both forwards: the first backward must differentiate the
    first forward's operand (functional._values_guard re-packs the values saved in ctx), and a third forward afterwards packs its own weight again."""
    import tgcn_amd
    from tgcn_amd import functional as F
    if not small:
        monkeypatch.setattr(F, "SMALL_PATH", False)
    rng = np.random.default_rng(21)
    n, E, f, g, K = 300, 2400, 4, 5, 4
    ei = torch.as_tensor(_edges(n, E, rng)).cuda()
    torch.manual_seed(0)
    layer = tgcn_amd.ChebConv(f, g, K).cuda()
    x0 = torch.as_tensor(rng.standard_normal((2, n, f)).astype(np.float32)).cuda()
    gout = torch.as_tensor(rng.standard_normal((2, n, g)).astype(np.float32)).cuda()
    wa0 = torch.as_tensor(rng.uniform(0.5, 1.5, E).astype(np.float32)).cuda()
    wb0 = torch.as_tensor(rng.uniform(0.5, 1.5, E).astype(np.float32)).cuda()

    def alone(w0):                       # each weight on its own: forward, backward, nothing in between
        w, x = w0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        layer.zero_grad()
        out = layer(x, ei, w)
        out.backward(gout)
        return [t.detach().clone() for t in (out, w.grad, x.grad, layer.weight.grad)]
    want_a, want_b = alone(wa0), alone(wb0)
    wa, wb = wa0.clone().requires_grad_(True), wb0.clone().requires_grad_(True)
    xa, xb = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    out_a = layer(xa, ei, wa)
    out_b = layer(xb, ei, wb)            # same pattern, other values: the operand now holds wb's
    layer.zero_grad()
    out_a.backward(gout)                 # ... and this backward needs wa's
    got_a = [out_a.detach(), wa.grad, xa.grad, layer.weight.grad.clone()]
    layer.zero_grad()
    out_b.backward(gout)
    got_b = [out_b.detach(), wb.grad, xb.grad, layer.weight.grad.clone()]
    for got, want in ((got_a, want_a), (got_b, want_b)):
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    again = alone(wa0)                   # the stamp of the last packing was dropped with the re-pack: a later forward packs its own weight
    for a, b in zip(again, want_a):
        assert torch.equal(a, b)
    # the same through spmm_batch_2 (module-global operand cache, one index, two value tensors)
    idx = torch.as_tensor(rng.integers(0, n, (2, E))).cuda()
    M = torch.as_tensor(rng.standard_normal((3, n, 6)).astype(np.float32)).cuda()
    g2 = torch.as_tensor(rng.standard_normal((3, n, 6)).astype(np.float32)).cuda()
    va, vb = wa0.clone().requires_grad_(True), wb0.clone().requires_grad_(True)
    Ma, Mb = M.clone().requires_grad_(True), M.clone().requires_grad_(True)
    oa = tgcn_amd.spmm_batch_2(idx, va, n, Ma)
    ob = tgcn_amd.spmm_batch_2(idx, vb, n, Mb)
    oa.backward(g2)
    ob.backward(g2)
    for v0, v, Mx in ((wa0, va, Ma), (wb0, vb, Mb)):
        A = torch.zeros(n, n, dtype=torch.float64, device="cuda").index_put((idx[0], idx[1]), v0.double(), accumulate=True)
        assert rel_err(Mx.grad.cpu().numpy(), torch.einsum("nm,qnc->qmc", A, g2.double()).cpu().numpy()) <= GRAD_TOL
