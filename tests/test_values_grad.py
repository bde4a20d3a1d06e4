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
