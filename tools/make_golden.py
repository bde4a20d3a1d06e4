#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference (cassianobecker/tgcn) in this container.

Usage (from anywhere, reference mounted read-only at /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--ref /root/reference] [--out tests/golden]

What is executed from the reference (nothing is copied; only inputs/outputs are stored):
  * gcn.graph            grid / distance_sklearn_metrics / adjacency / laplacian / rescale_L / chebyshev
  * gcn.coarsening       coarsen / metis / compute_perm / perm_data / perm_adjacency
  * tgcn.nn.gcn_matmul   GCNCheb, TGCNCheb, TGCNCheb_H              (imports with plain torch)
  * tgcn.nn.gcn          GCNCheb, TGCNCheb, TGCNCheb_H, ChebConv, ChebTimeConv, spmm, spmm_batch_2,
                         spmm_batch_3, gcn_pool, gcn_pool_4, uniform

tgcn.nn.gcn needs three third-party functions that are not installed here and are not part of the
reference repository (torch_geometric.utils.degree, torch_geometric.utils.remove_self_loops,
torch_scatter.scatter_add; no version is pinned anywhere in the reference).  They are provided below
from their published semantics so that the reference's OWN module code runs unmodified; fixtures
whose result went through them carry `third_party_restated=1`.

The fixtures are data only: graph arrays, inputs, parameters, and the reference's outputs.
"""
import argparse
import os
import sys
import types

import numpy as np


def _install_third_party_standins():
    import torch

    def degree(index, num_nodes=None, dtype=None):
        n = int(index.max()) + 1 if num_nodes is None else num_nodes
        out = torch.zeros((n,), dtype=dtype, device=index.device)
        return out.scatter_add_(0, index, out.new_ones((index.size(0),)))

    def remove_self_loops(edge_index, edge_attr=None):
        row, col = edge_index
        mask = row != col
        edge_attr = edge_attr if edge_attr is None else edge_attr[mask]
        return edge_index[:, mask], edge_attr

    def scatter_add(src, index, dim=-1, out=None, dim_size=None, fill_value=0):
        dim = dim if dim >= 0 else src.dim() + dim
        size = list(src.shape)
        size[dim] = int(index.max()) + 1 if dim_size is None else dim_size
        res = src.new_full(size, fill_value)
        return res.index_add_(dim, index, src)

    tg = types.ModuleType("torch_geometric")
    tgu = types.ModuleType("torch_geometric.utils")
    tgu.degree, tgu.remove_self_loops = degree, remove_self_loops
    tg.utils = tgu
    ts = types.ModuleType("torch_scatter")
    ts.scatter_add = scatter_add
    sys.modules.update({"torch_geometric": tg, "torch_geometric.utils": tgu, "torch_scatter": ts})


def csr_arrays(L):
    L = L.tocsr()
    L.sort_indices()
    return dict(n=np.int64(L.shape[0]), rowptr=L.indptr.astype(np.int64), col=L.indices.astype(np.int32),
                val=L.data.astype(np.float32))


def rmat_edges(scale, nedges, rng, abcd=(0.57, 0.19, 0.19, 0.05)):
    a, b, c, _ = abcd
    src = np.zeros(nedges, np.int64)
    dst = np.zeros(nedges, np.int64)
    for _ in range(scale):
        r = rng.random(nedges)
        bit_s = (r >= a + b).astype(np.int64)
        bit_d = (((r >= a) & (r < a + b)) | (r >= a + b + c)).astype(np.int64)
        src = (src << 1) | bit_s
        dst = (dst << 1) | bit_d
    return src, dst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)

    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    import warnings
    warnings.filterwarnings("ignore")
    import scipy.io
    import scipy.sparse as sp
    import torch
    _install_third_party_standins()
    import gcn.graph as rgraph            # reference
    import tgcn.nn.gcn as rgcn            # reference
    import tgcn.nn.gcn_matmul as rmm      # reference
    assert rgraph.__file__.startswith(args.ref) and rgcn.__file__.startswith(args.ref)

    torch.set_num_threads(1)
    rng = np.random.default_rng(12345)

    # ------------------------------------------------------------------ graphs
    graphs = {}
    z = rgraph.grid(28)
    dist, idx = rgraph.distance_sklearn_metrics(z, k=8)
    A = rgraph.adjacency(dist, idx)
    graphs["grid784"] = rgraph.rescale_L(rgraph.laplacian(A, normalized=True), lmax=2).tocsr()
    A_grid = A

    S = scipy.io.loadmat(os.path.join(args.ref, "load/res/average1.aparc.a2009s.dti.conn.mat"))["S"]
    A_dti = sp.csr_matrix(S.astype(np.float32))
    graphs["dti148"] = rgraph.rescale_L(rgraph.laplacian(A_dti, normalized=True), lmax=2).tocsr()

    s, d = rmat_edges(10, 6000, rng)
    keep = s != d
    s, d = s[keep], d[keep]
    A_r = sp.coo_matrix((np.ones(len(s), np.float32), (s, d)), shape=(1024, 1024)).tocsr()
    A_r = ((A_r + A_r.T) > 0).astype(np.float32).tocsr()
    graphs["rmat1024"] = rgraph.rescale_L(rgraph.laplacian(A_r, normalized=True), lmax=2).tocsr()

    # coarsening-style padding: 6x6 grid graph + 12 isolated (fake) vertices, interleaved
    z6 = rgraph.grid(6)
    d6, i6 = rgraph.distance_sklearn_metrics(z6, k=4)
    A6 = rgraph.adjacency(d6, i6).tocoo()
    perm = rng.permutation(48)
    A_pad = sp.coo_matrix((A6.data, (perm[A6.row], perm[A6.col])), shape=(48, 48)).tocsr()
    L_pad = rgraph.laplacian(A_pad, normalized=True)
    L_pad = rgraph.rescale_L(L_pad, lmax=2).tocsr()
    graphs["pad48"] = L_pad

    def save(name, **kw):
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **kw)
        print("%-44s %7.1f KB" % (name, os.path.getsize(path) / 1024))

    def params(mod, seed):
        torch.manual_seed(seed)
        mod.reset_parameters()
        return mod

    def grads(layer, x, *graph_args):
        """Gradients of the reference module's OWN autograd for a fixed grad_output (what every torch example trains
        with, examples/pytorch_based/pytorch_hcp_tgcn.py:167-169).  Draws from torch's generator only, so the numpy
        stream that shapes the other fixtures is untouched."""
        xg = x.clone().requires_grad_(True)
        layer.zero_grad()
        out = layer(xg, *graph_args)
        torch.manual_seed(2)
        go = torch.randn_like(out)
        out.backward(go)
        return dict(grad_out=go.numpy(), grad_x=xg.grad.numpy(), grad_weight=layer.weight.grad.numpy(),
                    grad_bias=(layer.bias.grad.numpy() if layer.bias is not None else np.zeros(0, np.float32)))

    # ------------------------------------------------------------ a2: GCNCheb
    cases = [("grid784", 3, 1, 8, 5, 2, True), ("grid784", 3, 4, 8, 5, 3, True), ("grid784", 2, 3, 5, 1, 3, True),
             ("grid784", 2, 3, 5, 2, 3, False), ("dti148", 2, 3, 7, 25, 3, True), ("rmat1024", 2, 8, 16, 10, 3, True),
             ("pad48", 5, 2, 3, 3, 3, True), ("grid784", 2, 64, 64, 5, 3, True)]
    for gi, (gname, q, f, g, K, xdim, bias) in enumerate(cases):
        L = graphs[gname]
        n = L.shape[0]
        Ld = torch.tensor(L.toarray(), dtype=torch.float32)
        for modname, M in (("gcn", rgcn), ("gcn_matmul", rmm)):
            layer = params(M.GCNCheb(Ld, f, g, K, bias=bias), 1)
            torch.manual_seed(0)
            x = torch.randn(q, n) if xdim == 2 else torch.randn(q, n, f)
            if xdim == 2:
                assert f == 1
            with torch.no_grad():
                out = layer(x)
                stack = layer._chebyshev(x)
            if modname == "gcn":
                ref_out = out
            else:
                assert torch.equal(out, ref_out), "gcn vs gcn_matmul differ"
        save("GCNCheb_%s_q%d_f%d_g%d_K%d_x%dd%s" % (gname, q, f, g, K, xdim, "" if bias else "_nobias"),
             kind="GCNCheb", K=K, x=x.numpy(), weight=layer.weight.detach().numpy(),
             bias=(layer.bias.detach().numpy() if bias else np.zeros(0, np.float32)), has_bias=int(bias),
             out=out.numpy(), stack=stack.numpy() if stack.numel() < 150000 else np.zeros(0, np.float32),
             third_party_restated=0, **grads(layer, x), **csr_arrays(L))

    # ----------------------------------------------------------- a3: TGCNCheb
    for gname, q, f, g, K, bias in (("grid784", 3, 4, 6, 5, True), ("dti148", 4, 2, 3, 4, False), ("pad48", 2, 3, 4, 6, True),
                                    ("rmat1024", 2, 16, 16, 5, True)):
        L = graphs[gname]
        n = L.shape[0]
        Ld = torch.tensor(L.toarray(), dtype=torch.float32)
        layer = params(rgcn.TGCNCheb(Ld, f, g, K, bias=bias), 1)
        torch.manual_seed(0)
        x = torch.randn(q, n, f)
        with torch.no_grad():
            out = layer(x)
            stack = layer._time_chebyshev(x)
            out_mm = params(rmm.TGCNCheb(Ld, f, g, K, bias=bias), 1)(x)
        assert torch.equal(out, out_mm)
        save("TGCNCheb_%s_q%d_f%d_g%d_K%d%s" % (gname, q, f, g, K, "" if bias else "_nobias"), kind="TGCNCheb", K=K,
             x=x.numpy(), weight=layer.weight.detach().numpy(),
             bias=(layer.bias.detach().numpy() if bias else np.zeros(0, np.float32)), has_bias=int(bias),
             out=out.numpy(), stack=stack.numpy() if stack.numel() < 150000 else np.zeros(0, np.float32), third_party_restated=0,
             **grads(layer, x), **csr_arrays(L))

    # --------------------------------------------------------- a1: TGCNCheb_H
    for gname, q, f, g, K, H, xdim, bias in (("grid784", 3, 1, 8, 5, 28, 3, True), ("dti148", 4, 1, 32, 10, 15, 3, True),
                                             ("grid784", 2, 1, 15, 10, 12, 3, True), ("rmat1024", 2, 3, 5, 4, 6, 4, False),
                                             ("pad48", 2, 2, 4, 3, 5, 4, True), ("dti148", 2, 1, 4, 1, 7, 3, True)):
        L = graphs[gname]
        n = L.shape[0]
        Ld = torch.tensor(L.toarray(), dtype=torch.float32)
        layer = params(rgcn.TGCNCheb_H(Ld, f, g, K, H, bias=bias), 1)
        torch.manual_seed(0)
        x = torch.randn(q, n, H) if xdim == 3 else torch.randn(q, n, H, f)
        with torch.no_grad():
            out = layer(x)
            stack = layer._time_chebyshev(x)
            out_mm = params(rmm.TGCNCheb_H(Ld, f, g, K, H, bias=bias), 1)(x)
        assert torch.equal(out, out_mm)
        save("TGCNChebH_%s_q%d_f%d_g%d_K%d_H%d%s" % (gname, q, f, g, K, H, "" if bias else "_nobias"), kind="TGCNCheb_H",
             K=K, H=H, x=x.numpy(), weight=layer.weight.detach().numpy(),
             bias=(layer.bias.detach().numpy() if bias else np.zeros(0, np.float32)), has_bias=int(bias),
             out=out.numpy(), stack=stack.numpy() if stack.numel() < 150000 else np.zeros(0, np.float32),
             third_party_restated=0, **grads(layer, x), **csr_arrays(L))

    # ----------------------------------------------- a4/a5: ChebConv / ChebTimeConv
    def edge_index_of(Acsr, self_loops=0, isolate=None, shuffle=True):
        coo = Acsr.tocoo()
        row, col = coo.row.astype(np.int64), coo.col.astype(np.int64)
        w = coo.data.astype(np.float32)
        if isolate is not None:                      # make `isolate` a source-degree-0 vertex
            keep = row != isolate
            row, col, w = row[keep], col[keep], w[keep]
        if self_loops:
            v = rng.choice(Acsr.shape[0], self_loops, replace=False).astype(np.int64)
            row, col = np.concatenate([row, v]), np.concatenate([col, v])
            w = np.concatenate([w, np.full(self_loops, 0.5, np.float32)])
        if shuffle:
            p = rng.permutation(len(row))
            row, col, w = row[p], col[p], w[p]
        return np.stack([row, col]), w

    ei_cases = [("grid784", A_grid, 3, 1, 8, 5, 2, True, 0, None, False), ("grid784", A_grid, 2, 4, 6, 25, 3, True, 5, None, False),
                ("dti148", A_dti, 3, 2, 5, 4, 3, False, 3, 7, True), ("rmat1024", A_r, 2, 3, 4, 6, 3, True, 4, 11, True),
                ("grid784", A_grid, 2, 1, 3, 1, 2, True, 0, None, False), ("grid784", A_grid, 2, 2, 3, 2, 3, True, 2, None, True)]
    for gname, Acsr, q, f, g, K, xdim, bias, nloops, iso, use_w in ei_cases:
        n = Acsr.shape[0]
        ei, w = edge_index_of(Acsr, nloops, iso)
        layer = params(rgcn.ChebConv(f, g, K, bias=bias), 1)
        torch.manual_seed(0)
        x = torch.randn(q, n) if xdim == 2 else torch.randn(q, n, f)
        with torch.no_grad():
            out = layer(x, torch.tensor(ei), torch.tensor(w) if use_w else None)
        save("ChebConv_%s_q%d_f%d_g%d_K%d_x%dd%s%s" % (gname, q, f, g, K, xdim, "_w" if use_w else "", "" if bias else "_nobias"),
             kind="ChebConv", K=K, n=np.int64(n), x=x.numpy(), edge_index=ei, edge_weight=w, use_weight=int(use_w),
             weight=layer.weight.detach().numpy(), bias=(layer.bias.detach().numpy() if bias else np.zeros(0, np.float32)),
             has_bias=int(bias), out=out.numpy(), third_party_restated=1,
             **grads(layer, x, torch.tensor(ei), torch.tensor(w) if use_w else None))

    for gname, Acsr, q, f, g, K, H, xdim, bias, nloops, iso, use_w in (
            ("dti148", A_dti, 3, 1, 32, 25, 15, 3, True, 0, None, False), ("grid784", A_grid, 2, 1, 8, 5, 12, 3, True, 3, None, False),
            ("rmat1024", A_r, 2, 2, 3, 4, 5, 4, False, 4, 9, True), ("grid784", A_grid, 2, 1, 4, 1, 6, 3, True, 0, None, False)):
        n = Acsr.shape[0]
        ei, w = edge_index_of(Acsr, nloops, iso)
        layer = params(rgcn.ChebTimeConv(f, g, K, H, bias=bias), 1)
        torch.manual_seed(0)
        x = torch.randn(q, n, H) if xdim == 3 else torch.randn(q, n, H, f)
        with torch.no_grad():
            out = layer(x, torch.tensor(ei), torch.tensor(w) if use_w else None)
        save("ChebTimeConv_%s_q%d_f%d_g%d_K%d_H%d%s%s" % (gname, q, f, g, K, H, "_w" if use_w else "", "" if bias else "_nobias"),
             kind="ChebTimeConv", K=K, H=H, n=np.int64(n), x=x.numpy(), edge_index=ei, edge_weight=w, use_weight=int(use_w),
             weight=layer.weight.detach().numpy(), bias=(layer.bias.detach().numpy() if bias else np.zeros(0, np.float32)),
             has_bias=int(bias), out=out.numpy(), third_party_restated=1,
             **grads(layer, x, torch.tensor(ei), torch.tensor(w) if use_w else None))

    # ------------------------------------------------ a6: spmm / spmm_batch_2 / spmm_batch_3
    ei, w = edge_index_of(A_grid, 3, None)
    tei, tw = torch.tensor(ei), torch.tensor(w)
    torch.manual_seed(0)
    m1, m2, m3 = torch.randn(784, 5), torch.randn(3, 784, 5), torch.randn(2, 784, 6, 3)
    v1 = torch.randn(784)
    save("spmm_grid784", kind="spmm", n=np.int64(784), edge_index=ei, value=w,
         m1=m1.numpy(), out1=rgcn.spmm(tei, tw, 784, m1).numpy(),
         v1=v1.numpy(), outv1=rgcn.spmm(tei, tw, 784, v1).numpy(),
         m2=m2.numpy(), out2=rgcn.spmm_batch_2(tei, tw, 784, m2).numpy(),
         m3=m3.numpy(), out3=rgcn.spmm_batch_3(tei, tw, 784, m3).numpy(), third_party_restated=1)

    # ---------------------------------------------------- a7: gcn.graph.chebyshev
    for gname, Mcols, K, dt in (("grid784", 16, 3, np.float32), ("grid784", 7, 10, np.float32), ("pad48", 5, 2, np.float32),
                                ("dti148", 6, 5, np.float64), ("rmat1024", 4, 1, np.float32)):
        L = graphs[gname].astype(dt)
        X = rng.standard_normal((L.shape[0], Mcols)).astype(dt)
        Xt = rgraph.chebyshev(L, X, K)
        save("graph_chebyshev2d_%s_N%d_K%d_%s" % (gname, Mcols, K, np.dtype(dt).name), kind="graph_chebyshev", K=K, X=X, out=Xt,
             third_party_restated=0, **{k: (v if k != "val" else L.tocsr().data) for k, v in csr_arrays(L).items()})
    # N-D branch (reshape-not-permute quirk, gcn/graph.py:267-283): pinned for the oracle only.
    L = graphs["pad48"].astype(np.float32)
    X = rng.standard_normal((3, 48, 4)).astype(np.float32)
    save("graph_chebyshev3d_pad48_K4", kind="graph_chebyshev_nd", K=4, X=X, out=rgraph.chebyshev(L, X, 4),
         third_party_restated=0, **csr_arrays(L))

    # ------------------------------------------------------------ a8 / a9
    torch.manual_seed(3)
    t = torch.empty(7, 5, 3)
    rgcn.uniform(35, t)
    xp = torch.randn(3, 16, 5)
    save("uniform_pool", kind="misc", uniform_size=35, uniform_seed=3, uniform_out=t.numpy(), pool_x=xp.numpy(),
         pool2=rgcn.gcn_pool(xp).numpy(), pool4=rgcn.gcn_pool_4(xp).numpy(), third_party_restated=0)


    # ------------------------------------------------ graph operand: rescale_L(laplacian(A, normalized=True), lmax)
    # (gcn/graph.py:117-136, 232-238) -- the step the callers run right before the layer (examples/gcn_mnist.py:131)
    for gname, Acsr, lm in (("grid784", A_grid, 2), ("dti148", A_dti, 2), ("rmat1024", A_r, 1.5)):
        Lh = rgraph.rescale_L(rgraph.laplacian(Acsr.astype(np.float32), normalized=True), lmax=lm).tocsr()
        coo = Acsr.tocoo()
        save("operand_%s_lmax%s" % (gname, str(lm).replace(".", "p")), kind="operand", lmax=np.float64(lm), n=np.int64(Acsr.shape[0]),
             a_row=coo.row.astype(np.int64), a_col=coo.col.astype(np.int64), a_val=coo.data.astype(np.float32),
             third_party_restated=0, **{("L_" + k): v for k, v in csr_arrays(Lh).items() if k != "n"})


    # ------------------------------------------------ graph coarsening (gcn/coarsening.py), the step in front of the pooling layers
    import contextlib
    import io
    import gcn.coarsening as rco
    # (the 1024-vertex R-MAT has isolated vertices at the end of its numbering: the reference's matching loop indexes past its
    # arrays there, coarsening.py:121,142, so it is not a fixture)
    for gname, Acsr, levels in (("grid784", A_grid, 4), ("dti148", A_dti, 2), ("grid36", A6.tocsr(), 3)):
        Acsr = Acsr.astype(np.float32).tocsr()
        np.random.seed(7)                                   # coarsen() draws its first visiting order from numpy's global generator
        with contextlib.redirect_stdout(io.StringIO()):
            graphs, perm = rco.coarsen(Acsr, levels=levels, self_connections=False)
        rid = np.random.RandomState(3).permutation(Acsr.shape[0])
        g2, parents = rco.metis(Acsr, levels, rid=rid)
        perms = rco.compute_perm(parents)
        xs = np.random.RandomState(4).standard_normal((3, Acsr.shape[0]))
        kw = dict(kind="coarsening", levels=levels, seed=7, a_rowptr=Acsr.indptr.astype(np.int64), a_col=Acsr.indices.astype(np.int32),
                  a_val=Acsr.data.astype(np.float32), n=np.int64(Acsr.shape[0]), perm=np.asarray(perm, np.int64), rid=rid.astype(np.int64),
                  x=xs, x_perm=rco.perm_data(xs, perm), third_party_restated=0)
        for i, G in enumerate(graphs):
            G = G.tocsr()
            G.sort_indices()
            kw.update({"g%d_rowptr" % i: G.indptr.astype(np.int64), "g%d_col" % i: G.indices.astype(np.int32), "g%d_val" % i: G.data.astype(np.float32),
                       "g%d_n" % i: np.int64(G.shape[0])})
        for i, p_ in enumerate(parents):
            kw["parents%d" % i] = np.asarray(p_, np.int32)
        for i, p_ in enumerate(perms):
            kw["perms%d" % i] = np.asarray(p_, np.int64)
        save("coarsen_%s_levels%d" % (gname, levels), **kw)


if __name__ == "__main__":
    main()
