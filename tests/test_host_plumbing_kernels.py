"""Round 5: the last torch index / layout ops of the product path became library calls -- weight re-layouts
(tgcn_weight_layout_f32), the wide-row (q, n, C) -> (n, q, C) re-layout, the relabelling of reordered operands and the row gathers of the
compacted layers (tgcn_pack_rows_f32), the signed sum of the left-out vertices' weight (tgcn_fold_weight_f32) -- and the compacted layer is ONE
driver call for both recurrences, with or without handing the hop tensors back (tgcn_cheb_compact_layer_f32).  Integer / copy work: exact."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.as_tensor(a).cuda()


@pytest.mark.parametrize("K,C,N", [(5, 64, 64), (1, 7, 3), (25, 15, 32), (3, 1200, 32),
                                   (25, 1536, 64), (11, 1200, 160)])      # the last two: K*C*N > 2^21 elements, beyond one pass of the capped grid
def test_weight_layouts_equal_torch_permutes(K, C, N, gpu_device):
    from tgcn_amd import functional as F
    W = torch.randn(K, C, N, device="cuda")
    assert torch.equal(F.weight_layout(W, 0), W.permute(1, 0, 2).reshape(C, K * N))
    assert torch.equal(F.weight_layout(W, 1), W.permute(0, 2, 1).contiguous())
    assert torch.equal(F.weight_layout(W, 2), W.permute(2, 0, 1).reshape(N, K * C))


@pytest.mark.parametrize("K,CN", [(5, 4096), (25, 1536 * 64), (3, (1 << 21) + 77)])      # CN beyond 8192 workgroups x 256 threads: grid-stride
def test_fold_weight_beyond_one_grid_pass(K, CN, gpu_device):
    from tgcn_amd import functional as F
    torch.manual_seed(K)
    W = torch.randn(K, CN, 1, device="cuda")
    fold = F.power_fold_matrix(K, W.device)
    for transpose in (False, True):
        got = F.fold_weight(fold, W, transpose=transpose)
        m = fold.double().t() if not transpose else fold.double()
        want = (m @ W.double().view(K, CN)).view(K, CN, 1)
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) <= 2e-6


def test_layer_with_a_weight_beyond_one_grid_pass(gpu_device):
    """K*C*N = 11 * 1200 * 160 > 2^21: the project-first forward (weight layout kind 0) and the input gradient (kinds 1 / 2) against float64"""
    import tgcn_amd
    rng = np.random.default_rng(5)
    n, K, H, g = 300, 11, 1200, 160
    ei = rng.integers(0, n, (2, 2400))
    A = np.zeros((n, n), np.float32)
    A[ei[0], ei[1]] = (rng.standard_normal(2400) / 6).astype(np.float32)
    L = torch.as_tensor(A).cuda()
    torch.manual_seed(2)
    layer = tgcn_amd.TGCNCheb_H(L, 1, g, K, H).cuda()
    x = torch.randn(2, n, H, device="cuda", requires_grad=True)
    out = layer(x)
    gout = torch.randn_like(out)
    out.backward(gout)
    xd = x.detach().double().requires_grad_(True)
    Wd = layer.weight.detach().double().reshape(K, H, g).requires_grad_(True)
    Ld = L.double()
    Xt, P = [xd], xd
    for k in range(1, K):
        P = torch.einsum("nm,qmc->qnc", Ld, P)
        Xt.append(P if k == 1 else 2 * P - Xt[k - 2])
    ref = sum(Xt[k] @ Wd[k] for k in range(K)) + layer.bias.detach().double()
    ref.backward(gout.double())
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 1e-5
    assert rel_err(x.grad.cpu().numpy(), xd.grad.cpu().numpy()) <= 2e-5
    assert rel_err(layer.weight.grad.reshape(K, H, g).cpu().numpy(), Wd.grad.cpu().numpy()) <= 2e-5


@pytest.mark.parametrize("q,n,C", [(3, 1000, 64), (2, 333, 33), (5, 70, 1200), (4, 900, 28), (1, 50, 64)])
def test_relayout_wide_and_short_rows(q, n, C, gpu_device):
    from tgcn_amd import functional as F
    x = torch.randn(q, n, C, device="cuda")
    assert torch.equal(F.relayout_qnc_to_nqc(x), x.permute(1, 0, 2).contiguous())


def test_relabel_rows_is_index_select_with_its_gradient(gpu_device):
    from tgcn_amd import functional as F
    n = 777
    perm = torch.randperm(n, device="cuda")
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device="cuda")
    x = torch.randn(3, n, 20, device="cuda", requires_grad=True)
    y = F.relabel_rows(x, perm, inv)
    assert torch.equal(y, x.index_select(1, perm))
    g = torch.randn_like(y)
    y.backward(g)
    want = torch.zeros_like(x)
    want[:, perm] = g
    assert torch.equal(x.grad, want)
    st = torch.randn(4, 2, n, 6, device="cuda")
    assert torch.equal(F.relabel_rows(st, inv, perm, dim=2), st.index_select(2, inv))
    b = torch.randn(n, 9, device="cuda")
    assert torch.equal(F.relabel_rows(b, perm, inv, dim=0), b.index_select(0, perm))


@pytest.mark.parametrize("K", [1, 2, 5, 8])
def test_left_out_weight_is_the_signed_sum(K, gpu_device):
    from tgcn_amd import functional as F
    W = torch.randn(K, 12, 20, device="cuda")
    sign = torch.tensor([(1.0 if k % 4 == 0 else -1.0) if k % 2 == 0 else 0.0 for k in range(K)], device="cuda")
    want = (W.double() * sign.view(K, 1, 1)).sum(0)
    got = F.left_out_weight(W, F.MODE_CHEBYSHEV)
    assert rel_err(got.cpu().numpy(), want.cpu().numpy()) <= 1e-6
    assert torch.equal(F.left_out_weight(W, F.MODE_POWER), W[0])


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("q,C,N,K,bias_kind", [(3, 64, 64, 5, 2), (2, 16, 24, 3, 1), (1, 15, 8, 4, 0), (5, 32, 16, 2, 2)])
def test_compact_layer_driver_both_recurrences_with_and_without_kept_terms(mode, q, C, N, K, bias_kind, gpu_device, monkeypatch):
    """tgcn_cheb_compact_layer_f32 against the float64 recursion on a graph with isolated vertices: the one-call forward (hop tensors in the
    workspace, 1 and several time steps per pass) equals the call that hands the terms back, bit for bit; the kept terms ARE the basis
    (monomials L^k x for mode 0, Chebyshev T_k x for mode 1) on the kept rows, and zero in the extra row."""
    import scipy.sparse as sp
    from tgcn_amd import functional as F, graph
    monkeypatch.setattr(graph, "COMPACT_MIN_ROWS", 1)
    rng = np.random.default_rng(q * 100 + C + K + mode)
    n = 9000
    live = rng.permutation(n)[: n // 2]                             # the other half of the vertices is isolated
    m = 40000
    u, v = live[rng.integers(0, live.size, m)], live[(rng.random(m) ** 2 * live.size).astype(np.int64)]
    u = np.concatenate([u, np.full(300, live[0])])                  # a row above the segment threshold
    v = np.concatenate([v, live[rng.integers(0, live.size, 300)]])
    row, col = np.concatenate([u, v]), np.concatenate([v, u])       # symmetric pattern: the kept set is closed
    val = (rng.standard_normal(row.size) / 4).astype(np.float32)
    op = graph.GraphOperand.from_coo(n, _dev(row), _dev(col), _dev(val))
    plan = op.compact_plan("rows" if mode == 0 else "closed")
    assert plan is not None and plan.n_empty >= n // 2
    x = rng.standard_normal((q, n, C)).astype(np.float32)
    W = (rng.standard_normal((K, C, N)) / np.sqrt(K * C)).astype(np.float32)
    bias = None if bias_kind == 0 else rng.standard_normal((N,) if bias_kind == 1 else (n, N)).astype(np.float32)
    Ls = sp.coo_matrix((val.astype(np.float64), (row, col)), shape=(n, n)).tocsr()
    T = [x.astype(np.float64)]
    for k in range(1, K):
        LT = np.stack([Ls @ T[-1][b] for b in range(q)])
        T.append(LT if (mode == 0 or k == 1) else 2 * LT - T[-2])
    ref = sum(T[k] @ W[k].astype(np.float64) for k in range(K))
    if bias is not None:
        ref = ref + bias
    xd, Wd = _dev(x), _dev(W)
    bd = None if bias is None else _dev(bias)
    W_left = F.left_out_weight(Wd, mode) if mode == 1 else None
    W2 = Wd.reshape(K * C, N).contiguous()
    outs = [F.cheb_forward_compact(plan, xd, W2, bd, bias_kind, K, q_chunk=qc, mode=mode, W_left=W_left) for qc in (1, 2, q)]
    out_k, terms = F.cheb_forward_compact(plan, xd, W2, bd, bias_kind, K, mode=mode, W_left=W_left, keep=True)
    assert rel_err(outs[0].cpu().numpy(), ref) <= 1e-5
    for o in outs[1:] + [out_k]:
        assert torch.equal(o, outs[0])
    assert len(terms) == K
    rows = plan.rows.long().cpu().numpy()
    for k in range(0 if mode == 1 else 1, K):
        t = terms[k].cpu().numpy()
        assert t.shape == (q, plan.n_c + 1, C) and not t[:, plan.n_c].any()
        assert rel_err(t[:, : plan.n_c], T[k][:, rows]) <= 1e-5
    # the module-level forms agree with the driver: training forward (keeps the basis) and the recomputed basis of a backward without one
    if F.choose_layout(q, n, C) == 0:        # (short per-sample rows take the vertex-major pipeline there: other tests)
        out2, terms2 = F.compact_forward(plan, xd, Wd, bd, bias_kind, mode)
        assert torch.equal(out2, outs[0]) and all(torch.equal(a, b) for a, b in zip(terms2, terms))
    terms3 = F.compact_terms(plan, xd, K, mode)
    assert all(torch.equal(a, b) for a, b in zip(terms3, terms))
