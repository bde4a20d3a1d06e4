"""project_x3_stream_kernel (csrc/project.h, round 5): the barrier-free streaming form of the bf16x3 projection that the row-mapped launches of
the compacted forward take (cfg5: 4.73 M compact rows x 5 terms and 5.27 M empty rows x 1 term per time step).  Against float64 numpy through the
C ABI on every feature it has: row maps (output + mapped terms), samples sharing the tile rows, the three bias kinds, ragged row counts,
16 ... 64 output columns incl. widths that are not a multiple of 16, rows of 32 and 64 floats, 1 ... 7 terms -- and against the tiled
kernel it replaces (same arithmetic up to the order of the products inside one MFMA)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev(a):
    return torch.as_tensor(a).cuda()


def _kinds():
    from tgcn_amd import _lib
    return _lib


@pytest.mark.parametrize("M,Kc,N,T,nbatch", [(40000, 64, 64, 5, 1), (33001, 64, 64, 1, 4), (5000, 64, 64, 5, 3), (4099, 32, 64, 3, 2), (2500, 64, 32, 2, 1),
                                             (3000, 64, 48, 4, 2), (1000, 32, 16, 7, 1), (777, 64, 20, 2, 3), (17, 64, 64, 5, 2), (16, 32, 8, 1, 1)])
@pytest.mark.parametrize("bias_kind", [0, 1, 2])
def test_stream_kernel_row_mapped_vs_numpy(M, Kc, N, T, nbatch, bias_kind, gpu_device):
    from tgcn_amd import functional as F, _lib
    rng = np.random.default_rng(M * 7 + Kc + N + T)
    n_vertices = M * 2 + 5                                    # the map picks M of these vertices, ascending (as graph.CompactPlan.rows)
    rowmap = np.sort(rng.choice(n_vertices, M, replace=False)).astype(np.int32)
    # term 0 lives in the caller's labels (read through the map), the others in tile-row order
    x = rng.standard_normal((nbatch, n_vertices, Kc)).astype(np.float32)
    rest = [rng.standard_normal((nbatch, M, Kc)).astype(np.float32) for _ in range(T - 1)]
    W = (rng.standard_normal((T, Kc, N)) / np.sqrt(T * Kc)).astype(np.float32)
    bias = None if bias_kind == 0 else rng.standard_normal(N if bias_kind == 1 else (n_vertices, N)).astype(np.float32)
    ref = np.einsum("bmk,kn->bmn", x[:, rowmap].astype(np.float64), W[0].astype(np.float64))
    for t in range(1, T):
        ref += np.einsum("bmk,kn->bmn", rest[t - 1].astype(np.float64), W[t].astype(np.float64))
    if bias_kind == 1:
        ref += bias
    elif bias_kind == 2:
        ref += bias[rowmap]
    outs = {}
    for variant in (6, 3):
        _lib.check(_lib.lib().tgcn_set_tuning(b"project_variant", variant))
        _lib.profile_start(64)
        out = torch.full((nbatch, n_vertices, N), float("nan"), device="cuda")
        terms = [_dev(x)] + [_dev(r) for r in rest]
        F.project_mapped(terms, [n_vertices * Kc] + [M * Kc] * (T - 1), _dev(W.reshape(T * Kc, N)), None if bias is None else _dev(bias), bias_kind,
                         n_vertices, _dev(rowmap), 1, nbatch, out)
        torch.cuda.synchronize()
        prof = _lib.profile_stop(64)
        # one launch for all samples in both kernels (the host loops over samples only for kernels without the in-kernel batch)
        assert [k for k, _ in prof] == [2], prof
        got = out.cpu().numpy()
        assert np.isnan(got[:, np.setdiff1d(np.arange(n_vertices), rowmap)]).all(), "rows outside the map were written"
        outs[variant] = got[:, rowmap]
        assert rel_err(outs[variant], ref) <= TOL
    # the two kernels differ only in the order of the 32 products inside one MFMA
    assert rel_err(outs[6], outs[3]) <= 2e-6


def test_stream_kernel_is_the_shipped_choice_for_the_compact_forward_shapes(gpu_device):
    """From 32768 rows the auto dispatch takes the streaming kernel for rows of 64 floats and 64 columns; the plain (unmapped) layer
    driver's projection too.  Checked through tgcn_cheb_project_f32 against float64, several samples inside M with a per-vertex bias."""
    from tgcn_amd import functional as F
    rng = np.random.default_rng(11)
    nv, q, Kc, N, T = 20000, 2, 64, 64, 5
    M = nv * q
    terms = [rng.standard_normal((M, Kc)).astype(np.float32) for _ in range(T)]
    W = (rng.standard_normal((T, Kc, N)) / np.sqrt(T * Kc)).astype(np.float32)
    bias = rng.standard_normal((nv, N)).astype(np.float32)
    ref = sum(t.astype(np.float64) @ w.astype(np.float64) for t, w in zip(terms, W))
    ref = (ref.reshape(q, nv, N) + bias).reshape(M, N)
    out = F.cheb_project([_dev(t) for t in terms], _dev(W), _dev(bias), 2, nv)
    assert rel_err(out.cpu().numpy(), ref) <= TOL


def test_stream_kernel_with_the_map_on_the_terms_only(gpu_device, monkeypatch):
    """tgcn_set_tuning("compact_proj", 1): ONE projection launch over all vertices in order -- output and bias rows are the tile rows, the
    hop tensors are read through the vertex -> compact-id map (empty vertices point at the zero row): the kProjMapTermsOnly form of the row map,
    which the streaming kernel takes from 32768 rows.  Against the two-launch default and against float64."""
    import scipy.sparse as sp
    from tgcn_amd import functional as F, graph, _lib
    monkeypatch.setattr(graph, "COMPACT_MIN_ROWS", 1)
    rng = np.random.default_rng(17)
    n, q, C, N, K = 50000, 3, 64, 64, 4
    live = rng.permutation(n)[: n // 2]
    m = 200000
    u, v = live[rng.integers(0, live.size, m)], live[(rng.random(m) ** 2 * live.size).astype(np.int64)]
    row, col = np.concatenate([u, v]), np.concatenate([v, u])
    val = (rng.standard_normal(row.size) / 5).astype(np.float32)
    op = graph.GraphOperand.from_coo(n, _dev(row), _dev(col), _dev(val))
    plan = op.compact_plan()
    assert plan is not None and plan.n_empty >= n // 2
    x = rng.standard_normal((q, n, C)).astype(np.float32)
    W = (rng.standard_normal((K, C, N)) / np.sqrt(K * C)).astype(np.float32)
    bias = rng.standard_normal((n, N)).astype(np.float32)
    Ls = sp.coo_matrix((val.astype(np.float64), (row, col)), shape=(n, n)).tocsr()
    P = [x.astype(np.float64)]
    for _ in range(1, K):
        P.append(np.stack([Ls @ P[-1][b] for b in range(q)]))
    ref = sum(P[k] @ W[k].astype(np.float64) for k in range(K)) + bias
    W2 = _dev(W.reshape(K * C, N))
    outs = {}
    for mode in (0, 1):
        _lib.check(_lib.lib().tgcn_set_tuning(b"compact_proj", mode))
        _lib.profile_start(256)
        outs[mode] = F.cheb_forward_compact(plan, _dev(x), W2, _dev(bias), 2, K, q_chunk=q).cpu().numpy()
        prof = _lib.profile_stop(256)
        assert sum(1 for kind, _ in prof if kind == 2) == (1 if mode else 2)
        assert rel_err(outs[mode], ref) <= TOL
    assert rel_err(outs[1], outs[0]) <= 2e-6
