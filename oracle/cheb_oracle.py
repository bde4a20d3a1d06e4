"""CPU oracle (numpy / scipy) for the Chebyshev (time-)graph convolution of cassianobecker/tgcn.

TEST INFRASTRUCTURE ONLY.  Nothing under tgcn_amd/ may import this module; only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() use it, as the checker.

Each function restates one piece of the reference and cites the lines it follows (paths relative to
the reference root).  Parity is PINNED: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which tools/make_golden.py produced by running the reference itself in the build
container (ChebConv / ChebTimeConv / spmm* went through restated torch_scatter / torch_geometric
helpers, which are third-party and absent from the reference tree; see that script's header).

Arithmetic is fp32 unless the caller passes fp64 operands (gcn/graph.py:247 takes the dtype from L).
"""
import numpy as np
import scipy.sparse as sp


# ----------------------------------------------------------------------------- graph operand
def csr_from_arrays(n, rowptr, col, val):
    return sp.csr_matrix((np.asarray(val), np.asarray(col), np.asarray(rowptr)), shape=(int(n), int(n)))


def edge_laplacian(edge_index, edge_weight, n, dtype=np.float32):
    """tgcn/nn/gcn.py:398-413 (ChebConv.forward) == :495-510 (ChebTimeConv.forward).

    remove_self_loops -> deg = COUNT of edges per source vertex (unweighted, gcn.py:408) ->
    deg^-1/2 with inf -> 0 (:411-412) -> lap_e = -deg[row] * w_e * deg[col] (:413).
    Returns (row, col, lap) with self loops removed, in the original edge order.
    """
    row, col = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    keep = row != col
    row, col = row[keep], col[keep]
    w = np.ones(row.shape[0], dtype) if edge_weight is None else np.asarray(edge_weight, dtype).reshape(-1)[keep]
    deg = np.bincount(row, minlength=n).astype(dtype)
    with np.errstate(divide="ignore"):
        dis = deg ** dtype(-0.5)
    dis[np.isinf(dis)] = 0
    lap = -dis[row] * w * dis[col]
    return row, col, lap.astype(dtype)


def coo_to_csr(row, col, val, n):
    """Duplicates are summed, which is what scatter_add does (gcn.py:308,343)."""
    return sp.coo_matrix((val, (row, col)), shape=(n, n)).tocsr()


def rescaled_laplacian(W, lmax=2):
    """gcn/graph.py:117-136 (laplacian, normalized) followed by :232-238 (rescale_L):
    d = W.sum(axis=0) + spacing(0); L = I - D^-1/2 W D^-1/2; L-hat = L / (lmax/2) - I."""
    W = sp.csr_matrix(W)
    d = np.asarray(W.sum(axis=0)).squeeze().astype(W.dtype)
    d = d + np.spacing(np.array(0, W.dtype))
    dis = (1 / np.sqrt(d)).astype(W.dtype)
    D = sp.diags(dis, 0)
    I = sp.identity(d.size, dtype=W.dtype, format="csr")
    L = I - D * W * D
    L = L / (lmax / 2)
    return (L - I).tocsr()


# ----------------------------------------------------------------------------- L x (batched)
def _apply(L, X):
    """L (n x n, scipy sparse) applied along axis 1 of X (q, n, ...): einsum("nm,qm...->qn...") of
    gcn.py:72,147,230 / torch.mm of gcn_matmul.py:154,242 / scatter form gcn.py:296-308,328-343."""
    q, n = X.shape[0], X.shape[1]
    tail = X.shape[2:]
    Xm = np.moveaxis(X.reshape(q, n, -1), 1, 0).reshape(n, -1)
    Y = L.dot(Xm).astype(X.dtype, copy=False)
    return np.moveaxis(Y.reshape((n, q, -1)), 0, 1).reshape((q, n) + tail)


def stack_reference_power(L, X, K):
    """The dense-L classes' recursion: Xt[0]=X, Xt[1]=L X, Xt[k]=2 L^k X - Xt[k-2]
    (tgcn/nn/gcn.py:63-79 TGCNCheb, :137-154 TGCNCheb_H, :219-237 GCNCheb; the running product X is
    re-multiplied, the true-recurrence line is commented out at :76,151,234)."""
    Xt = np.empty((K,) + X.shape, X.dtype)
    Xt[0] = X
    P = X
    if K > 1:
        P = _apply(L, P)
        Xt[1] = P
    for k in range(2, K):
        P = _apply(L, P)
        Xt[k] = 2 * P - Xt[k - 2]
    return Xt


def stack_chebyshev(L, X, K):
    """True recurrence Tx_k = 2 L Tx_{k-1} - Tx_{k-2} (tgcn/nn/gcn.py:420-432, :519-528)."""
    Xt = np.empty((K,) + X.shape, X.dtype)
    Xt[0] = X
    if K > 1:
        Xt[1] = _apply(L, X)
    for k in range(2, K):
        Xt[k] = 2 * _apply(L, Xt[k - 1]) - Xt[k - 2]
    return Xt


# ----------------------------------------------------------------------------- the five layers
def gcn_cheb_forward(L, x, weight, bias):
    """GCNCheb.forward, tgcn/nn/gcn.py:189-200 (+ _chebyshev :208-237; 2-D input unsqueezed :216-217).
    weight (K,f,g); bias (1,1,g) or None."""
    if x.ndim == 2:
        x = x[:, :, None]
    xc = stack_reference_power(L, x, weight.shape[0])
    out = np.einsum("kqnf,kfg->qng", xc, weight, optimize=True).astype(np.float32)
    return out if bias is None else out + bias


def tgcn_cheb_forward(L, x, weight, bias):
    """TGCNCheb.forward, tgcn/nn/gcn.py:34-44 (+ _time_chebyshev :52-79; no 2-D unsqueeze). bias (1,n,g)."""
    xc = stack_reference_power(L, x, weight.shape[0])
    out = np.einsum("kqnf,kfg->qng", xc, weight, optimize=True).astype(np.float32)
    return out if bias is None else out + bias


def tgcn_cheb_h_forward(L, x, weight, bias):
    """TGCNCheb_H.forward, tgcn/nn/gcn.py:108-118 (+ _time_chebyshev :126-154; 3-D input unsqueezed
    :134-135). weight (K,H,f,g); bias (1,n,g)."""
    if x.ndim == 3:
        x = x[..., None]
    xc = stack_reference_power(L, x, weight.shape[0])
    out = np.einsum("kqnhf,khfg->qng", xc, weight, optimize=True).astype(np.float32)
    return out if bias is None else out + bias


def cheb_conv_forward(x, edge_index, edge_weight, weight, bias):
    """ChebConv.forward, tgcn/nn/gcn.py:396-437. weight (K,f,g); bias (g,)."""
    n = x.shape[1]
    row, col, lap = edge_laplacian(edge_index, edge_weight, n, x.dtype.type)
    L = coo_to_csr(row, col, lap, n)
    if x.ndim < 3:
        x = x[..., None]
    xc = stack_chebyshev(L, x, weight.shape[0])
    out = np.einsum("kqnf,kfg->qng", xc, weight, optimize=True).astype(np.float32)
    return out if bias is None else out + bias


def cheb_time_conv_forward(x, edge_index, edge_weight, weight, bias):
    """ChebTimeConv.forward, tgcn/nn/gcn.py:493-533. weight (K,H,f,g); bias (g,)."""
    n = x.shape[1]
    row, col, lap = edge_laplacian(edge_index, edge_weight, n, x.dtype.type)
    L = coo_to_csr(row, col, lap, n)
    if x.ndim < 4:
        x = x[..., None]
    xc = stack_chebyshev(L, x, weight.shape[0])
    out = np.einsum("kqnhf,khfg->qng", xc, weight, optimize=True).astype(np.float32)
    return out if bias is None else out + bias


def torch_dense_forward(L_dense, x, weight, bias, horizon=False):
    """The dense-L forward exactly as the reference evaluates it on the CPU -- torch einsum with the (n, n) matrix
    (tgcn/nn/gcn.py:72-78,147-153,230-236 and :39,113,194) -- for the timed CPU baseline of the small configurations
    (baseline (ii)).  x (q, n, f) or (q, n, h, f) torch CPU tensors; weight (K, f, g) / (K, H, f, g)."""
    import torch
    K = weight.shape[0]
    Xt = [x]
    P = x
    eq = "nm,qmhf->qnhf" if horizon else "nm,qmf->qnf"
    for k in range(1, K):
        P = torch.einsum(eq, L_dense, P)
        Xt.append(P if k == 1 else 2 * P - Xt[k - 2])
    xc = torch.stack(Xt)
    out = torch.einsum("kqnhf,khfg->qng" if horizon else "kqnf,kfg->qng", xc, weight)
    return out if bias is None else out + bias


# ----------------------------------------------------------------------------- gradients (what autograd derives)
def layer_backward(L, x, weight, grad_out, mode):
    """Gradients of out = sum_k T_k(L) x W_k (+ bias) w.r.t. x and weight, in fp64: what torch autograd derives from the
    forwards above (the examples train through them, examples/pytorch_based/pytorch_hcp_tgcn.py:167-169).  T_k is linear,
    so  dW_k = (T_k x)^T g  and  dx = sum_k T_k(L^T) (g W_k^T).  mode "power": the dense-L classes' recursion
    (stack_reference_power), "chebyshev": the edge-list classes' (stack_chebyshev).  x (q, n, *tail), weight (K, *tail, g).
    Pinned by the grad_* arrays of tests/golden (tools/make_golden.py::grads)."""
    stack = stack_reference_power if mode == "power" else stack_chebyshev
    K, gch = weight.shape[0], weight.shape[-1]
    q, n = x.shape[:2]
    L64 = sp.csr_matrix(L, dtype=np.float64)
    x64 = np.asarray(x, np.float64).reshape(q, n, -1)
    W = np.asarray(weight, np.float64).reshape(K, -1, gch)
    g = np.asarray(grad_out, np.float64)
    Xt = stack(L64, x64, K)
    gW = np.einsum("kqnt,qng->ktg", Xt, g, optimize=True).reshape(weight.shape)
    LT = L64.T.tocsr()
    gx = np.zeros_like(x64)
    for k in range(K):
        gx += stack(LT, g @ W[k].T, k + 1)[k]
    return gx.reshape(x.shape), gW


# ----------------------------------------------------------------------------- COO SpMM helpers
def spmm(index, value, m, matrix):
    """tgcn/nn/gcn.py:258-278: out[r] += v_e * matrix[c] over the FIRST axis (1-D input is
    unsqueezed :271)."""
    matrix = matrix if matrix.ndim > 1 else matrix[:, None]
    L = coo_to_csr(np.asarray(index[0]), np.asarray(index[1]), np.asarray(value), m)
    return L.dot(matrix.reshape(matrix.shape[0], -1)).reshape((m,) + matrix.shape[1:]).astype(matrix.dtype)


def spmm_batch(index, value, m, matrix):
    """tgcn/nn/gcn.py:281-310 (spmm_batch_2) and :313-345 (spmm_batch_3): same product over axis 1."""
    L = coo_to_csr(np.asarray(index[0]), np.asarray(index[1]), np.asarray(value), m)
    if matrix.ndim == 2:                      # spmm_batch_2 :297-301 appends a channel axis
        matrix = matrix[..., None]
    return _apply(L, matrix)


# ----------------------------------------------------------------------------- numpy twin
def graph_chebyshev(L, X, K):
    """gcn/graph.py:241-283. 2-D X (M,N): reference_power recursion with scipy CSR .dot (:256-265).
    N-D X: true recurrence on X.reshape(X.shape[1], -1) -- a reshape, NOT a permute (:267-283), so
    samples are mixed; restated literally because the reference does it."""
    Xt = np.empty((K,) + X.shape, L.dtype)
    Xt[0] = X
    if X.ndim == 2:
        P = X
        if K > 1:
            P = L.dot(P)
            Xt[1] = P
        for k in range(2, K):
            P = L.dot(P)
            Xt[k] = 2 * P - Xt[k - 2]
        return Xt
    sh = X.shape
    if K > 1:
        Xt[1] = L.dot(X.reshape(sh[1], -1)).reshape(sh)
    for k in range(2, K):
        Xt[k] = 2 * L.dot(Xt[k - 1].reshape(sh[1], -1)).reshape(sh) - Xt[k - 2]
    return Xt


# ----------------------------------------------------------------------------- init / pooling
def uniform_bound(size):
    """tgcn/nn/gcn.py:240-243: U(-1/sqrt(size), +1/sqrt(size)), size = in_channels * K (:29,103,179,392,489)."""
    return 1.0 / np.sqrt(size)


def gcn_pool(x, p=2):
    """tgcn/nn/gcn.py:246-249 (p=2) and :252-255 (p=4): max over p consecutive vertices."""
    return x.reshape(x.shape[0], x.shape[1] // p, p, x.shape[2]).max(axis=2)
