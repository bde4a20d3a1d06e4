"""Seeded random sweep of the five module classes over graph sizes / densities / widths / orders that land on every
forward path (one-launch sparse and dense kernels, project-first, hops-first in both layouts, narrow projection) and,
for the small cases, the backward against fp64 dense autograd.  Everything through the modules, i.e. the way a
caller of the reference reaches the path.  Tolerance as everywhere: max|a-b| / max|b| <= 1e-5 (2e-5 for gradients)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import cheb_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _graph(n, density_kind, rng):
    """-> symmetric 0/1 adjacency without self loops as scipy CSR (both ChebConv's edge list and the dense-L classes'
    rescaled Laplacian are derived from it)."""
    import scipy.sparse as sp
    if density_kind == "dense":
        A = (rng.random((n, n)) < 0.6).astype(np.float32)
    elif density_kind == "hub":
        A = (rng.random((n, n)) < 4.0 / n).astype(np.float32)
        A[rng.integers(0, n)] = 1.0
    else:
        A = (rng.random((n, n)) < 6.0 / n).astype(np.float32)
    A = np.triu(A, 1)
    A = A + A.T
    A[:, 0] = A[0, :] = 0                                     # vertex 0 isolated (padding vertex of a coarsened graph)
    return sp.csr_matrix(A)


CASES = []
_rng = np.random.default_rng(2024)
for _ in range(int(os.environ.get("TGCN_FUZZ_CASES", "80"))):       # TGCN_FUZZ_CASES=400 for a longer sweep
    n = int(_rng.choice([20, 60, 148, 200, 500, 1100, 2500, 5000]))
    kind = str(_rng.choice(["sparse", "hub", "dense"])) if n <= 256 else str(_rng.choice(["sparse", "hub"]))
    big_batch = n <= 200 and _rng.random() < 0.25            # fills the chip: dense operands then take the bf16x3 kernels
    CASES.append(dict(n=n, kind=kind, q=int(_rng.integers(200, 500)) if big_batch else int(_rng.integers(1, 6)), K=int(_rng.choice([1, 2, 3, 5, 8])),
                      H=int(_rng.choice([1, 3, 15, 40])), f=int(_rng.choice([1, 2, 4, 16])), g=int(_rng.choice([3, 8, 32, 64])),
                      cls=str(_rng.choice(["GCNCheb", "TGCNCheb", "TGCNCheb_H", "ChebConv", "ChebTimeConv"])),
                      bias=bool(_rng.integers(0, 2)), seed=int(_rng.integers(1 << 30))))


def _check_forward(case):
    """Forward of one random case against the oracle (shared by the sweep below and the tuning-switch test)."""
    import tgcn_amd
    rng = np.random.default_rng(case["seed"])
    n, q, K, H, f, g, cls = (case[k] for k in ("n", "q", "K", "H", "f", "g", "cls"))
    A = _graph(n, case["kind"], rng)
    coo = A.tocoo()
    ei = np.stack([coo.row, coo.col]).astype(np.int64)
    timed = cls in ("TGCNCheb_H", "ChebTimeConv")
    torch.manual_seed(case["seed"] % 1000)
    if cls in ("ChebConv", "ChebTimeConv"):
        layer = (tgcn_amd.ChebTimeConv(f, g, K, H, bias=case["bias"]) if timed else tgcn_amd.ChebConv(f, g, K, bias=case["bias"])).cuda()
        L = None
    else:
        L = O.rescaled_laplacian(A.astype(np.float32), lmax=2)
        ctor = dict(GCNCheb=tgcn_amd.GCNCheb, TGCNCheb=tgcn_amd.TGCNCheb)
        layer = (tgcn_amd.TGCNCheb_H(L, f, g, K, H, bias=case["bias"]) if timed else ctor[cls](L, f, g, K, bias=case["bias"])).cuda()
    x = rng.standard_normal((q, n, H, f) if timed else (q, n, f)).astype(np.float32)
    W = layer.weight.detach().cpu().numpy()
    b = layer.bias.detach().cpu().numpy() if case["bias"] else None
    xt = torch.tensor(x, device="cuda")
    with torch.no_grad():
        out = layer(xt, torch.tensor(ei, device="cuda")) if L is None else layer(xt)
    ref = dict(GCNCheb=lambda: O.gcn_cheb_forward(L, x, W, b), TGCNCheb=lambda: O.tgcn_cheb_forward(L, x, W, b),
               TGCNCheb_H=lambda: O.tgcn_cheb_h_forward(L, x, W, b), ChebConv=lambda: O.cheb_conv_forward(x, ei, None, W, b),
               ChebTimeConv=lambda: O.cheb_time_conv_forward(x, ei, None, W, b))[cls]()
    assert rel_err(out.cpu().numpy(), ref) <= TOL, case


@pytest.mark.parametrize("key,value", [("small_dense", 0), ("small_dense", 1), ("small_narrow", 0), ("project_variant", 1), ("project_variant", 3),
                                       ("project_variant", 4), ("x3_form", 1), ("overlap", 1), ("hop_variant", 1)])
def test_tuning_switches_keep_the_result(key, value, gpu_device):
    """Every tgcn_set_tuning switch selects another kernel for the same arithmetic: the first 24 random cases must still
    match the oracle with the switch thrown."""
    from tgcn_amd import _lib
    _lib.check(_lib.lib().tgcn_set_tuning(key.encode(), value))
    try:
        for case in CASES[:24]:
            _check_forward(case)
    finally:
        default = {"small_dense": 2, "small_narrow": 1, "project_variant": 0, "x3_form": 2, "overlap": 0, "hop_variant": 0}[key]
        _lib.check(_lib.lib().tgcn_set_tuning(key.encode(), default))


@pytest.mark.parametrize("case", CASES, ids=["%s-n%d-%s-q%d-K%d-H%d-f%d-g%d" % (c["cls"], c["n"], c["kind"], c["q"], c["K"], c["H"], c["f"], c["g"]) for c in CASES])
def test_random_module_vs_oracle(case, gpu_device):
    import tgcn_amd
    rng = np.random.default_rng(case["seed"])
    n, q, K, H, f, g, cls = (case[k] for k in ("n", "q", "K", "H", "f", "g", "cls"))
    A = _graph(n, case["kind"], rng)
    coo = A.tocoo()
    ei = np.stack([coo.row, coo.col]).astype(np.int64)
    timed = cls in ("TGCNCheb_H", "ChebTimeConv")
    torch.manual_seed(case["seed"] % 1000)
    if cls in ("ChebConv", "ChebTimeConv"):
        layer = (tgcn_amd.ChebTimeConv(f, g, K, H, bias=case["bias"]) if timed else tgcn_amd.ChebConv(f, g, K, bias=case["bias"])).cuda()
        L = None
    else:
        L = O.rescaled_laplacian(A.astype(np.float32), lmax=2)
        ctor = dict(GCNCheb=tgcn_amd.GCNCheb, TGCNCheb=tgcn_amd.TGCNCheb)
        layer = (tgcn_amd.TGCNCheb_H(L, f, g, K, H, bias=case["bias"]) if timed else ctor[cls](L, f, g, K, bias=case["bias"])).cuda()
    shape = (q, n, H, f) if timed else (q, n, f)
    x = rng.standard_normal(shape).astype(np.float32)
    W = layer.weight.detach().cpu().numpy()
    b = layer.bias.detach().cpu().numpy() if case["bias"] else None
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    out = layer(xt, torch.tensor(ei, device="cuda")) if L is None else layer(xt)
    ref = dict(GCNCheb=lambda: O.gcn_cheb_forward(L, x, W, b), TGCNCheb=lambda: O.tgcn_cheb_forward(L, x, W, b),
               TGCNCheb_H=lambda: O.tgcn_cheb_h_forward(L, x, W, b), ChebConv=lambda: O.cheb_conv_forward(x, ei, None, W, b),
               ChebTimeConv=lambda: O.cheb_time_conv_forward(x, ei, None, W, b))[cls]()
    assert out.shape == ref.shape
    assert rel_err(out.detach().cpu().numpy(), ref) <= TOL
    if n > 256:
        return
    # ---- backward against dense fp64 autograd of the same formula
    go = torch.randn_like(out)
    out.backward(go)
    if L is None:
        r_, c_, lap = O.edge_laplacian(ei, None, n)
        Ld = torch.tensor(O.coo_to_csr(r_, c_, lap, n).toarray(), dtype=torch.float64, device="cuda")
    else:
        Ld = torch.tensor(L.toarray(), dtype=torch.float64, device="cuda")
    xd = xt.detach().double().requires_grad_(True)
    Wd = layer.weight.detach().double().requires_grad_(True)
    x3 = xd.reshape(q, n, -1)
    Xt = [x3]
    P = x3
    for k in range(1, K):
        if L is not None:
            P = torch.einsum("nm,qmc->qnc", Ld, P)
            Xt.append(P if k == 1 else 2 * P - Xt[k - 2])
        else:
            LX = torch.einsum("nm,qmc->qnc", Ld, Xt[k - 1])
            Xt.append(LX if k == 1 else 2 * LX - Xt[k - 2])
    refd = torch.einsum("kqnc,kcg->qng", torch.stack(Xt), Wd.reshape(K, -1, g))
    refd.backward(go.double())
    gtol = 2e-5
    assert rel_err(xt.grad.cpu().numpy(), xd.grad.cpu().numpy()) <= gtol
    assert rel_err(layer.weight.grad.cpu().numpy(), Wd.grad.cpu().numpy()) <= gtol
    if case["bias"]:
        gb = go.double().sum(dim=(0, 1)) if layer.bias.numel() == g else go.double().sum(dim=0)
        assert rel_err(layer.bias.grad.cpu().numpy().reshape(-1), gb.cpu().numpy().reshape(-1)) <= gtol


def test_random_modules_on_compact_hop_tensors(gpu_device, monkeypatch):
    """The same random cases (five classes, vertex 0 isolated in every graph, hubs, K = 1 ... 8, widths 1 ... 640) with the general path
    FORCED onto compact hop tensors -- every layout (incl. the vertex-major form that is off by default), both recursions, the one-call driver
    and the python-level pipeline, with the last hop fused into the projection where that form exists -- against the oracle."""
    from tgcn_amd import functional as F, graph, _lib
    monkeypatch.setattr(graph, "COMPACT_MIN_ROWS", 1)
    monkeypatch.setattr(graph, "COMPACT_MIN_EMPTY", 0.0)
    monkeypatch.setattr(F, "SMALL_PATH", False)
    monkeypatch.setattr(F, "PROJECT_FIRST", False)
    monkeypatch.setattr(F, "COMPACT_LAYOUT1", True)
    used = {"drv": 0, "py": 0}
    real_drv, real_py = F.cheb_forward_compact, F.compact_forward
    monkeypatch.setattr(F, "cheb_forward_compact", lambda *a, **k: (used.__setitem__("drv", used["drv"] + 1), real_drv(*a, **k))[1])
    monkeypatch.setattr(F, "compact_forward", lambda *a, **k: (used.__setitem__("py", used["py"] + 1), real_py(*a, **k))[1])
    for fuse in (0, 1):
        _lib.check(_lib.lib().tgcn_set_tuning(b"fuse_last_hop", fuse))
        for case in CASES[: max(40, len(CASES) // 3)]:        # TGCN_FUZZ_CASES=400: 133 cases per form
            _check_forward(case)
    assert used["drv"] >= 10 and used["py"] >= 10, used
