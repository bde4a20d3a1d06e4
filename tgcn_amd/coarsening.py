"""Multi-level graph coarsening for the pooling layers -- the caller-side step in front of the Chebyshev layers
(reference: gcn/coarsening.py).  Same functions, arguments and results as the reference (`coarsen`, `metis`,
`compute_perm`, `perm_data`, `perm_adjacency`; checked against fixtures made by running it, tests/test_coarsening.py); the
per-vertex matching loop of one level -- pure Python in the reference (coarsening.py:119-165) -- runs as host code inside
libtgcn_hip.so (tgcn_graclus_match_f32/_f64).  One-off host preprocessing: scipy / numpy own the sparse bookkeeping here, and
`perm_data_device` applies the resulting vertex order to a batch on the GPU."""
import ctypes as C

import numpy as np
import scipy.sparse

from . import _lib


def coarsen(A, levels, self_connections=False, verbose=False):
    """Graphs of `levels` successive coarsenings, vertices ordered so that the two children of every coarse vertex are
    neighbours (gcn_pool / gcn_pool_4 then pool siblings), plus the permutation of the finest level for the data
    (reference: coarsening.py:5-32).  -> (graphs, perm)"""
    graphs, parents = metis(A, levels)
    perms = compute_perm(parents)
    for i, G in enumerate(graphs):
        M = G.shape[0]
        if not self_connections:
            G = G.tocoo()
            G.setdiag(0)
        if i < levels:
            G = perm_adjacency(G, perms[i])
        G = G.tocsr()
        G.eliminate_zeros()
        graphs[i] = G
        if verbose:
            print("level %d: %d vertices (%d added), %d edges" % (i, G.shape[0], G.shape[0] - M, G.nnz // 2))
    return graphs, perms[0] if levels > 0 else None


def graclus_match(rr, cc, vv, rid, weights):
    """cluster id of every vertex after one greedy matching pass (reference: metis_one_level, coarsening.py:119-165).
    rr / cc / vv: entries sorted by row; rid: visiting order; weights: vertex weights, same dtype family as vv."""
    single = np.asarray(vv).dtype == np.float32 and np.asarray(weights).dtype == np.float32
    ft = np.float32 if single else np.float64
    rr = np.ascontiguousarray(rr, np.int64)
    cc = np.ascontiguousarray(cc, np.int64)
    vv = np.ascontiguousarray(vv, ft)
    rid = np.ascontiguousarray(rid, np.int64)
    weights = np.ascontiguousarray(weights, ft)
    n = int(rr[-1]) + 1 if rr.size else int(rid.size)
    assert rid.size >= n and weights.size >= n
    out = np.zeros(n, np.int32)
    fn = _lib.lib().tgcn_graclus_match_f32 if single else _lib.lib().tgcn_graclus_match_f64
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.check(fn(rr.size, p(rr), p(cc), p(vv), n, p(rid), p(weights), p(out)))
    return out


def metis(W, levels, rid=None):
    """`levels` coarsenings of the symmetric weight matrix W: graphs[0] = W, graphs[i + 1] = graphs[i] with matched pairs
    merged; parents[i][v] = vertex of graphs[i + 1] that v of graphs[i] went into (reference: coarsening.py:34-116).
    First visiting order: `rid` or a random permutation (numpy's global generator, like the reference); later levels visit
    vertices by increasing weighted degree."""
    N = W.shape[0]
    if rid is None:
        rid = np.random.permutation(range(N))
    degree = W.sum(axis=0) - W.diagonal()
    graphs, parents = [W], []
    for _ in range(levels):
        weights = np.array(degree).squeeze()
        idx_row, idx_col, val = scipy.sparse.find(W)
        by_row = np.argsort(idx_row)                       # the reference's (unstable) sort: the order inside a row breaks ties
        rr, cc, vv = idx_row[by_row], idx_col[by_row], val[by_row]
        cluster = graclus_match(rr, cc, vv, rid, weights)
        parents.append(cluster)
        n_new = int(cluster.max()) + 1
        W = scipy.sparse.csr_matrix((vv, (cluster[rr], cluster[cc])), shape=(n_new, n_new))
        W.eliminate_zeros()
        graphs.append(W)
        degree = W.sum(axis=0)
        rid = np.argsort(np.array(W.sum(axis=0)).squeeze())
    return graphs, parents


def compute_perm(parents):
    """Vertex orders, finest level first, in which the children of every coarse vertex are adjacent; vertices with fewer than
    two children get fake siblings numbered after the real vertices (reference: coarsening.py:167-217)."""
    if not parents:
        return []
    orders = [list(range(int(max(parents[-1])) + 1))]
    for parent in parents[::-1]:
        parent = np.asarray(parent)
        children = [[] for _ in range(max(int(parent.max()) + 1, len(orders[-1])))]
        for v, p in enumerate(parent):
            children[p].append(v)
        fake = len(parent)
        layer = []
        for p in orders[-1]:
            kids = list(children[p]) if p < len(children) else []
            assert len(kids) <= 2
            while len(kids) < 2:                            # a singleton gets one fake sibling, a fake parent two fake children
                kids.append(fake)
                fake += 1
            layer.extend(kids)
        orders.append(layer)
    m_last = len(orders[0])
    for i, layer in enumerate(orders):
        assert sorted(layer) == list(range(m_last * 2 ** i))
    return orders[::-1]


def perm_data(x, indices):
    """Columns of the (samples, vertices) data matrix in the coarsening order, zero columns for fake vertices
    (reference: coarsening.py:222-244)."""
    if indices is None:
        return x
    N, M = x.shape
    idx = np.asarray(indices)
    assert idx.size >= M
    out = np.zeros((N, idx.size))
    real = idx < M
    out[:, real] = x[:, idx[real]]
    return out


def perm_data_device(x, indices):
    """perm_data for a torch tensor (samples, vertices[, ...]) on its device: a gather along the vertex axis, zeros for fake
    vertices; dtype kept."""
    import torch
    if indices is None:
        return x
    idx = torch.as_tensor(np.asarray(indices), device=x.device)
    M = x.shape[1]
    real = idx < M
    out = torch.zeros((x.shape[0], idx.numel()) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
    out[:, real] = x.index_select(1, idx[real])
    return out


def perm_adjacency(A, indices):
    """Adjacency with fake (isolated) vertices appended and vertices renumbered into the coarsening order
    (reference: coarsening.py:246-276).  Returns COO like the reference."""
    if indices is None:
        return A
    M = A.shape[0]
    m_new = len(indices)
    assert m_new >= M
    A = A.tocoo()
    position = np.argsort(indices)                           # new index of old vertex v
    return scipy.sparse.coo_matrix((A.data, (position[A.row], position[A.col])), shape=(m_new, m_new))
