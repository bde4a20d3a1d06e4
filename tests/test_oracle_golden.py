"""Pins oracle/cheb_oracle.py against the golden vectors produced by running the reference
(tools/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden_files, golden_ids, load_golden, rel_err
from oracle import cheb_oracle as O

TOL = 1e-5   # max|a-b|/max|b| <= 1e-5, fp32


def _L(g):
    return O.csr_from_arrays(g["n"], g["rowptr"], g["col"], g["val"])


def _bias(g):
    return g["bias"] if int(g["has_bias"]) else None


@pytest.mark.parametrize("path", golden_files("GCNCheb_"), ids=golden_ids(golden_files("GCNCheb_")))
def test_gcncheb(path):
    g = load_golden(path)
    out = O.gcn_cheb_forward(_L(g), g["x"], g["weight"], _bias(g))
    assert rel_err(out, g["out"]) <= TOL
    if g["stack"].size:
        x = g["x"][:, :, None] if g["x"].ndim == 2 else g["x"]
        assert rel_err(O.stack_reference_power(_L(g), x, int(g["K"])), g["stack"]) <= TOL


@pytest.mark.parametrize("path", golden_files("TGCNCheb_"), ids=golden_ids(golden_files("TGCNCheb_")))
def test_tgcncheb(path):
    g = load_golden(path)
    assert rel_err(O.tgcn_cheb_forward(_L(g), g["x"], g["weight"], _bias(g)), g["out"]) <= TOL
    if g["stack"].size:
        assert rel_err(O.stack_reference_power(_L(g), g["x"], int(g["K"])), g["stack"]) <= TOL


@pytest.mark.parametrize("path", golden_files("TGCNChebH_"), ids=golden_ids(golden_files("TGCNChebH_")))
def test_tgcncheb_h(path):
    g = load_golden(path)
    assert rel_err(O.tgcn_cheb_h_forward(_L(g), g["x"], g["weight"], _bias(g)), g["out"]) <= TOL
    if g["stack"].size:
        x = g["x"][..., None] if g["x"].ndim == 3 else g["x"]
        assert rel_err(O.stack_reference_power(_L(g), x, int(g["K"])), g["stack"]) <= TOL


@pytest.mark.parametrize("path", golden_files("ChebConv_"), ids=golden_ids(golden_files("ChebConv_")))
def test_chebconv(path):
    g = load_golden(path)
    w = g["edge_weight"] if int(g["use_weight"]) else None
    assert rel_err(O.cheb_conv_forward(g["x"], g["edge_index"], w, g["weight"], _bias(g)), g["out"]) <= TOL


@pytest.mark.parametrize("path", golden_files("ChebTimeConv_"), ids=golden_ids(golden_files("ChebTimeConv_")))
def test_chebtimeconv(path):
    g = load_golden(path)
    w = g["edge_weight"] if int(g["use_weight"]) else None
    assert rel_err(O.cheb_time_conv_forward(g["x"], g["edge_index"], w, g["weight"], _bias(g)), g["out"]) <= TOL


_GRAD_FILES = [p for pre in ("GCNCheb_", "TGCNCheb_", "TGCNChebH_", "ChebConv_", "ChebTimeConv_") for p in golden_files(pre)]


@pytest.mark.parametrize("path", _GRAD_FILES, ids=golden_ids(_GRAD_FILES))
def test_backward_restatement(path):
    """O.layer_backward against the gradients the reference modules' own autograd produced"""
    g = load_golden(path)
    kind = str(g["kind"])
    x = g["x"]
    if kind in ("GCNCheb", "ChebConv") and x.ndim == 2:
        x = x[:, :, None]
    if kind in ("TGCNCheb_H", "ChebTimeConv") and x.ndim == 3:
        x = x[..., None]
    if kind in ("ChebConv", "ChebTimeConv"):
        n = x.shape[1]
        w = g["edge_weight"] if int(g["use_weight"]) else None
        row, col, lap = O.edge_laplacian(g["edge_index"], w, n, np.float32)
        L, mode = O.coo_to_csr(row, col, lap, n), "chebyshev"
    else:
        L, mode = _L(g), "power"
    gx, gW = O.layer_backward(L, x, g["weight"], g["grad_out"], mode)
    assert rel_err(gx.reshape(g["grad_x"].shape), g["grad_x"]) <= 2e-5
    assert rel_err(gW, g["grad_weight"]) <= 2e-5
    if int(g["has_bias"]):
        go = g["grad_out"].astype(np.float64)
        gb = go.sum(axis=0, keepdims=True) if kind in ("TGCNCheb", "TGCNCheb_H") else go.sum(axis=(0, 1))
        assert rel_err(gb.reshape(g["grad_bias"].shape), g["grad_bias"]) <= 2e-5


def test_spmm_helpers():
    g = load_golden(golden_files("spmm_")[0])
    n = int(g["n"])
    assert rel_err(O.spmm(g["edge_index"], g["value"], n, g["m1"]), g["out1"]) <= TOL
    assert rel_err(O.spmm(g["edge_index"], g["value"], n, g["v1"]), g["outv1"]) <= TOL
    assert rel_err(O.spmm_batch(g["edge_index"], g["value"], n, g["m2"]), g["out2"]) <= TOL
    assert rel_err(O.spmm_batch(g["edge_index"], g["value"], n, g["m3"]), g["out3"]) <= TOL


@pytest.mark.parametrize("path", golden_files("graph_chebyshev"), ids=golden_ids(golden_files("graph_chebyshev")))
def test_graph_chebyshev(path):
    g = load_golden(path)
    L = _L(g).astype(g["val"].dtype)
    out = O.graph_chebyshev(L, g["X"], int(g["K"]))
    assert out.dtype == g["out"].dtype
    # same scipy CSR kernels in the same order: bit-exact
    assert np.array_equal(out, g["out"])


def test_uniform_and_pool():
    g = load_golden(golden_files("uniform_pool")[0])
    assert np.abs(g["uniform_out"]).max() <= O.uniform_bound(int(g["uniform_size"]))
    assert np.array_equal(O.gcn_pool(g["pool_x"], 2), g["pool2"])
    assert np.array_equal(O.gcn_pool(g["pool_x"], 4), g["pool4"])


@pytest.mark.parametrize("path", golden_files("operand_"), ids=golden_ids(golden_files("operand_")))
def test_operand_builder(path):
    """rescale_L(laplacian(A)) : the oracle's restatement and the product's device-side builder (torch index ops,
    runs on CPU tensors too) against the reference's own output."""
    import scipy.sparse as sp
    import torch
    from tgcn_amd.graph import GraphOperand
    g = load_golden(path)
    n = int(g["n"])
    ref = O.csr_from_arrays(n, g["L_rowptr"], g["L_col"], g["L_val"])
    A = sp.coo_matrix((g["a_val"], (g["a_row"], g["a_col"])), shape=(n, n)).tocsr()
    mine = O.rescaled_laplacian(A, float(g["lmax"]))
    assert abs(mine - ref).max() <= 1e-6
    op = GraphOperand.from_adjacency(n, torch.as_tensor(g["a_row"]), torch.as_tensor(g["a_col"]), torch.as_tensor(g["a_val"]),
                                     lmax=float(g["lmax"]))
    assert abs(op.to_scipy() - ref).max() <= 1e-6


# ---------------------------------------------------------------------------- plain-C restatement (CPU baseline)
def _c_forward(g, mode, x3, W, bias_kind, L=None):
    from oracle import c_port
    if L is None:
        L = _L(g)
    L = L.tocsr()
    L.sort_indices()
    b = _bias(g)
    return c_port.forward(mode, L.indptr.astype(np.int32), L.indices.astype(np.int32), L.data.astype(np.float32), x3, W,
                          None if b is None else b.reshape(-1), bias_kind if b is not None else 0)


@pytest.mark.parametrize("path", golden_files("GCNCheb_") + golden_files("TGCNCheb_") + golden_files("TGCNChebH_"),
                         ids=golden_ids(golden_files("GCNCheb_") + golden_files("TGCNCheb_") + golden_files("TGCNChebH_")))
def test_c_port_dense_classes(path):
    g = load_golden(path)
    x = g["x"]
    kind = str(g["kind"])
    if kind == "GCNCheb" and x.ndim == 2:
        x = x[:, :, None]
    if kind == "TGCNCheb_H" and x.ndim == 3:
        x = x[..., None]
    q, n = x.shape[:2]
    W = g["weight"].reshape(g["weight"].shape[0], -1, g["weight"].shape[-1])
    out = _c_forward(g, 0, x.reshape(q, n, -1), W, 1 if kind == "GCNCheb" else 2)
    assert rel_err(out, g["out"]) <= TOL


@pytest.mark.parametrize("path", golden_files("ChebConv_") + golden_files("ChebTimeConv_"),
                         ids=golden_ids(golden_files("ChebConv_") + golden_files("ChebTimeConv_")))
def test_c_port_edge_classes(path):
    g = load_golden(path)
    x = g["x"]
    kind = str(g["kind"])
    if x.ndim < (3 if kind == "ChebConv" else 4):
        x = x[..., None]
    q, n = x.shape[:2]
    w = g["edge_weight"] if int(g["use_weight"]) else None
    row, col, lap = O.edge_laplacian(g["edge_index"], w, n)
    W = g["weight"].reshape(g["weight"].shape[0], -1, g["weight"].shape[-1])
    out = _c_forward(g, 1, x.reshape(q, n, -1), W, 1, L=O.coo_to_csr(row, col, lap, n))
    assert rel_err(out, g["out"]) <= TOL


@pytest.mark.parametrize("path", [p for p in golden_files("GCNCheb_") + golden_files("TGCNChebH_") if "rmat" not in p][:6], ids=lambda p: p.split("/")[-1][:-4])
def test_torch_dense_baseline_restatement(path):
    """O.torch_dense_forward (the timed dense-L CPU baseline of the small configurations) against the reference's outputs"""
    import torch
    g = load_golden(path)
    horizon = str(g["kind"]) == "TGCNCheb_H"
    x = torch.tensor(g["x"])
    if horizon and x.dim() == 3:
        x = x.unsqueeze(-1)
    if not horizon and x.dim() == 2:
        x = x.unsqueeze(-1)
    Ld = torch.tensor(_L(g).toarray(), dtype=torch.float32)
    b = torch.tensor(g["bias"]) if int(g["has_bias"]) else None
    out = O.torch_dense_forward(Ld, x, torch.tensor(g["weight"]), b, horizon)
    assert rel_err(out.numpy(), g["out"]) <= TOL
