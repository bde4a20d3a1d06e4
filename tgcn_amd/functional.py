"""Host side of the hot path: thin wrappers over the C ABI + the autograd function of the fused layer.

Every product, sum and axpy of the forward runs in libtgcn_hip.so (include/tgcn_hip.h); torch supplies device
memory, the stream and autograd bookkeeping.  There is no CPU fallback.
"""
import ctypes as C
import functools
import threading

import torch

from . import _lib

SMALL_PATH = True   # developer switch (tools/): route small graphs through the general multi-launch path instead
MODE_POWER = 0      # dense-L classes: Xt[k] = 2 L^k x - Xt[k-2]  (tgcn/nn/gcn.py:75-78,150-153,233-236)
MODE_CHEBYSHEV = 1  # edge-list classes: Tx_k = 2 L Tx_{k-1} - Tx_{k-2}  (gcn.py:427-432,524-528)
BIAS_NONE, BIAS_CHANNEL, BIAS_VERTEX_CHANNEL = 0, 1, 2


def _cuda_device_of(a):
    """device of a CUDA tensor / GraphOperand / CompactPlan argument (or of the first element of a list of them), else None"""
    if isinstance(a, torch.Tensor):
        return a.device if a.is_cuda else None
    if isinstance(a, (list, tuple)):
        return _cuda_device_of(a[0]) if a else None
    d = getattr(a, "device", None)
    return d if isinstance(d, torch.device) and d.type == "cuda" else None


def _on_device(fn):
    """Every entry that calls the library runs with the device of its DATA current, on that device's current stream -- the reference's
    torch ops follow the tensor's device (tgcn/nn/gcn.py:141,147: `model.to('cuda:1')(x.to('cuda:1'))` works whatever device is
    current, and nn.DataParallel's replicas rely on it, examples/pytorch_based/pytorch_hcp_tgcn.py:270-273), while a raw launch goes
    to the calling thread's current device.  Arguments on two different devices are refused here (the kernels would dereference a
    foreign pointer); the C ABI checks the same against hipGetDevice (common.h: check_pointer_device)."""
    @functools.wraps(fn)
    def on_device(*args, **kw):
        dev = None
        for a in args:
            d = _cuda_device_of(a)
            if d is None:
                continue
            if dev is None:
                dev = d
            elif d != dev:
                raise _lib.TgcnError("tgcn_amd.%s: arguments on two devices (%s and %s) -- move the input, the parameters and the operand to one device"
                                     % (fn.__name__, dev, d))
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return on_device


def _dense(t):
    """(nb, rows, C) view with a contiguous last dim -> tgcn_dense (strides in floats)."""
    assert t.dim() == 3 and t.dtype == torch.float32 and (t.shape[2] == 1 or t.stride(2) == 1), (t.shape, t.stride())
    return _lib.DenseStruct(t.data_ptr(), int(t.stride(0)), int(t.stride(1)))


def _aligned16(C_row, *tensors):
    ok = C_row % 4 == 0
    for t in tensors:
        if t is not None:
            ok = ok and t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0
    return ok


def weight_layout(W_kcn, kind):
    """Re-layouts of a (K, C, N) layer weight in libtgcn_hip.so (tgcn_weight_layout_f32): kind 0 -> (C, K*N) column blocks W_0 | ... | W_{K-1}
    (project-first form); kind 1 -> (K, N, C), W_k^T (the input gradient as a layer on (L^T, g, W^T)); kind 2 -> (N, K*C) (G = g W^T for all
    terms in one projection)."""
    _lib.require_device(W_kcn)
    K, Cc, N = W_kcn.shape
    W = W_kcn.contiguous()
    out = torch.empty({0: (Cc, K * N), 1: (K, N, Cc), 2: (N, K * Cc)}[kind], dtype=torch.float32, device=W.device)
    with torch.cuda.device(W.device):
        _lib.check(_lib.lib().tgcn_weight_layout_f32(_lib.stream_ptr(), K, Cc, N, _lib.ptr(W), _lib.ptr(out), kind))
    return out


class RowPermFn(torch.autograd.Function):
    """out[b, i] = x[b, perm[i]] for a (nb, rows, C) tensor -- the relabelling a GraphOperand.reordered() operand needs on the way in (perm)
    and out (inv_perm) -- as a differentiable op on tgcn_pack_rows_f32: the backward is the same kernel with the inverse permutation."""

    @staticmethod
    def forward(ctx, x3, perm, inv):
        ctx.perm, ctx.inv = perm, inv
        return _permute_rows(x3, perm)

    @staticmethod
    def backward(ctx, g):
        return _permute_rows(g, ctx.inv), None, None


def _permute_rows(x3, perm):
    _lib.require_device(x3, perm)
    x3 = x3.float().contiguous()
    out = torch.empty_like(x3)
    nb, rows, Cw = x3.shape
    with torch.cuda.device(x3.device):
        if nb > 1 and nb * rows <= (1 << 22):
            # many small samples (the reference's batches of 64 ... 512 on coarsened graphs): ONE launch over all of them -- the row index of
            # sample b is b * rows + perm (integer plumbing on nb * rows elements)
            idx = (torch.arange(nb, device=x3.device, dtype=torch.int64).unsqueeze(1) * rows + perm.unsqueeze(0)).reshape(-1)
            pack_rows(x3.view(nb * rows, Cw), idx, out.view(nb * rows, Cw))
        else:
            for b in range(nb):
                pack_rows(x3[b], perm, out[b])
    return out


def relabel_rows(t, perm, inv, dim=1):
    """t with its axis `dim` (the vertex axis) permuted: t.index_select(dim, perm), differentiable, through the library's row-packing kernel"""
    sh = t.shape
    lead = 1
    for d in sh[:dim]:
        lead *= d
    t3 = t.reshape(lead, sh[dim], -1)
    return RowPermFn.apply(t3, perm, inv).reshape(sh)


# ----------------------------------------------------------------------------------------- single ops
@_on_device
def csr_hop(op, x, z=None, alpha=1.0, beta=0.0, want_p=False, out=None, p_out=None, z2=None, gamma=0.0):
    """One hop:  S = L x;  y = alpha*S + beta*z (+ gamma*z2);  optionally also S.  x: (nb, op.n_cols, C); y, z, z2, S:
    (nb, op.n, C); any of them may be a strided view as long as the last dim is contiguous.  Returns y (and S when want_p).
    Low-level: rows are in the OPERAND's labels -- for a GraphOperand.reordered() operand the caller relabels (x[:, op.perm] in,
    y[:, op.inv_perm] out), as cheb_layer / cheb_stack / cheb_time_windows / cheb_relu_pool do."""
    _lib.require_device(x, z, z2)
    L = _lib.lib()
    assert x.dim() == 3 and x.dtype == torch.float32 and x.shape[1] == op.n_cols, (x.shape, op.n_cols)
    nb, _, Crow = x.shape
    y = torch.empty((nb, op.n, Crow), dtype=torch.float32, device=x.device) if out is None else out
    p = (torch.empty((nb, op.n, Crow), dtype=torch.float32, device=x.device) if p_out is None else p_out) if want_p else None
    for t in (y, z, p, z2):
        assert t is None or tuple(t.shape) == (nb, op.n, Crow), (t.shape, (nb, op.n, Crow))
    al = _aligned16(Crow, x, z, y, p, z2)
    sched = op.schedule_for(Crow, al)
    # the batch rides on grid.y (x column chunks of wide rows): slice batches beyond its 65535 limit
    nb_max = 65535 // max(1, -(-Crow // (256 if al else 64)))
    if nb > nb_max:
        for b0 in range(0, nb, nb_max):
            sl = slice(b0, min(nb, b0 + nb_max))
            csr_hop(op, x[sl], None if z is None else z[sl], alpha, beta, want_p, y[sl], None if p is None else p[sl],
                    None if z2 is None else z2[sl], gamma)
        return (y, p) if want_p else y
    ws_bytes = L.tgcn_csr_hop_workspace_bytes(C.byref(sched.struct), nb, Crow, 1 if al else 0)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    X, Y = _dense(x), _dense(y)
    Z = _dense(z) if z is not None else None
    P = _dense(p) if p is not None else None
    Z2 = _dense(z2) if z2 is not None else None
    _lib.check(L.tgcn_csr_hop2_f32(_lib.stream_ptr(), C.byref(op.struct), C.byref(sched.struct), nb, Crow, C.byref(X),
                                   C.byref(Z) if Z is not None else None, float(alpha), float(beta),
                                   C.byref(Z2) if Z2 is not None else None, float(gamma), C.byref(Y),
                                   C.byref(P) if P is not None else None, _lib.ptr(ws), ws.numel()))
    return (y, p) if want_p else y


@_on_device
def cheb_project(terms, W, bias, bias_kind, n_vertices, interleave=1, out=None):
    """out[r(m)] = sum_t terms[t][m, :] @ W[t] + bias; terms: list of (M, Kc) contiguous; W: (T, Kc, N)."""
    _lib.require_device(W, *terms)
    L = _lib.lib()
    T = len(terms)
    M, Kc = terms[0].shape
    N = W.shape[-1]
    W = W.reshape(T * Kc, N).contiguous()
    rows_out = M
    if out is None:
        out = torch.empty((rows_out, N), dtype=torch.float32, device=W.device)
    b = bias.contiguous() if bias is not None else None
    for t0 in range(0, T, 32):
        nt = min(32, T - t0)
        last = t0 + nt >= T
        a = (C.c_void_p * nt)(*[terms[t0 + i].data_ptr() for i in range(nt)])
        lda = (C.c_int64 * nt)(*[terms[t0 + i].stride(0) for i in range(nt)])
        _lib.check(L.tgcn_cheb_project_f32(_lib.stream_ptr(), M, Kc, N, nt, a, lda, _lib.ptr(W[t0 * Kc:]),
                                           _lib.ptr(b) if last else None, bias_kind if last else 0, n_vertices,
                                           interleave, 1 if t0 > 0 else 0, _lib.ptr(out), N))
    return out


@_on_device
def project_first(x3, Wcat, bias, bias_kind, K, N, rowmap=None):
    """Z = x . [W_0 | ... | W_{K-1}] with the bias on the first N columns (Z_0): the first step of the project-first form as its own call
    (tgcn_cheb_project_first_f32) for callers that run the recursion themselves -- the vertex-sharded layer exchanges cut rows between its
    hops.  x3: (q, rows, C); Wcat: (C, K*N) (weight_layout kind 0); bias read at the OUTPUT row; rowmap (int32[rows], nullable): output row of
    input row m.  -> (q, rows, K*N)"""
    _lib.require_device(x3, Wcat, bias, rowmap)
    q, rows, Crow = x3.shape
    assert x3.is_contiguous() and Wcat.is_contiguous() and tuple(Wcat.shape) == (Crow, K * N) and (rowmap is None or (rowmap.dtype == torch.int32 and rowmap.numel() == rows))
    if x3.data_ptr() % 16:
        x3 = x3.clone()
    Z = torch.empty((q, rows, K * N), dtype=torch.float32, device=x3.device)
    b = bias.contiguous() if bias is not None else None
    _lib.check(_lib.lib().tgcn_cheb_project_first_f32(_lib.stream_ptr(), q, rows, Crow, K, N, _lib.ptr(x3), _lib.ptr(Wcat), _lib.ptr(b),
                                                      bias_kind if b is not None else BIAS_NONE, _lib.ptr(rowmap), _lib.ptr(Z)))
    return Z


@_on_device
def csr_sddmm(op, rows3, cols3, alpha=1.0, out=None, accumulate=False):
    """dval[e] (+)= alpha * sum_b sum_c rows3[b, row(e), c] * cols3[b, col(e), c] over the stored entries of `op`, in CSR order -> (nnz,).
    The gradient of S = L X w.r.t. the values of L is csr_sddmm(op, dS, X) (tgcn_csr_sddmm_f32)."""
    _lib.require_device(rows3, cols3, out)
    nb, _, Crow = rows3.shape
    assert rows3.shape[1] == op.n and cols3.shape[1] == op.n_cols and cols3.shape[0] == nb and cols3.shape[2] == Crow
    if out is None:
        out = torch.zeros(max(op.nnz, 1), dtype=torch.float32, device=rows3.device)[: op.nnz]
        accumulate = False
    R, Cc = _dense(rows3), _dense(cols3)
    _lib.check(_lib.lib().tgcn_csr_sddmm_f32(_lib.stream_ptr(), C.byref(op.struct), op.n_cols, nb, Crow, C.byref(R), C.byref(Cc), float(alpha),
                                             _lib.ptr(out), 1 if accumulate else 0))
    return out


def _values_guard(op, epoch, values_csr):
    """Learnable values are packed IN PLACE into one operand per pattern (GraphOperand.update_values), and a backward reads the operand's
    CURRENT values (hops on op.transpose(), the recomputed basis).  When the same pattern has been packed with OTHER values between a forward
    and its backward -- two spmm calls with one index and two weight tensors, one ChebConv called twice with computed edge_weights -- the
    operand no longer holds what this backward differentiates: re-pack the values the forward ran with (saved in ctx), and drop the stamp of
    the last packing so that the next forward packs its own weight again.  torch's in-place version check cannot see this (the packed
    entries are not a tensor of the graph); the old per-weight cache key was correct by construction, this keeps the per-pattern operand
    correct too."""
    if values_csr is None or op.values_epoch == epoch:
        return
    with op._lock:
        if op.values_epoch != epoch:
            op.update_values(values_csr)
            op._packed_stamp = None
            op._packed_alias = None


class SpmmFn(torch.autograd.Function):
    """out = L matrix3 (one hop) as a differentiable op: d matrix = L^T g (the hop on the cached transposed operand), d values = the sampled
    product <g[row(e)], matrix[col(e)]> in CSR order (csr_sddmm) -- the reference's spmm* are differentiable in both (gcn.py:296-308)."""

    @staticmethod
    @_on_device
    def forward(ctx, matrix3, values_csr, op):
        ctx.op = op
        ctx.values_epoch = op.values_epoch
        ctx.save_for_backward(matrix3, None if values_csr is None else values_csr.detach())
        return csr_hop(op, matrix3)

    @staticmethod
    @_on_device
    def backward(ctx, g):
        matrix3, vals = ctx.saved_tensors
        _values_guard(ctx.op, ctx.values_epoch, vals)
        g = g.contiguous()
        gm = csr_hop(ctx.op.transpose(), g) if ctx.needs_input_grad[0] else None
        gv = csr_sddmm(ctx.op, g, matrix3) if ctx.needs_input_grad[1] else None
        return gm, gv, None


@_on_device
def chebyshev_values_grad(op, x3, W_kcn, g, basis=None):
    """d loss / d values (CSR order) of the true-recurrence layer out = sum_k T_k W_k, T_1 = L x, T_k = 2 L T_{k-1} - T_{k-2} (ChebConv /
    ChebTimeConv: lap_e = -deg^-1/2[row] w_e deg^-1/2[col] is differentiable in w_e in the reference, tgcn/nn/gcn.py:413,510).
    With G_k = g W_k^T and the Clenshaw adjoints b_{K-1} = G_{K-1}, b_k = G_k + 2 L^T b_{k+1} - b_{k+2}:
        dL = b_1 T_0^T + 2 sum_{k>=2} b_k T_{k-1}^T   sampled on the stored pattern (csr_sddmm).  Hops, projection and SDDMM in libtgcn_hip.so.
    basis: the (q, n, C) terms T_0 .. T_{K-1} when the forward kept them (ChebLayerFn: forward_keeping_basis) -- otherwise T_0 .. T_{K-2} are
    recomputed here.  Every adjoint is consumed by its sampled product as soon as it exists and only the two the recurrence still needs stay
    alive (the K-1 adjoints used to be held until the SDDMM loop)."""
    K, Crow, N = W_kcn.shape
    q, n, _ = x3.shape
    dval = torch.zeros(max(op.nnz, 1), dtype=torch.float32, device=x3.device)[: op.nnz]
    if K < 2 or op.nnz == 0:
        return dval
    if basis is not None:
        T = basis
    else:
        T = cheb_stack(op, x3.contiguous(), K - 1, MODE_CHEBYSHEV) if K > 2 else x3.contiguous().unsqueeze(0)      # T_0 .. T_{K-2}
    g = g.contiguous()
    Wcat = weight_layout(W_kcn, 2).view(1, N, K * Crow)
    Gall = cheb_project([g.reshape(q * n, N)], Wcat, None, BIAS_NONE, n).view(q, n, K * Crow)
    G = lambda k: Gall[:, :, k * Crow:(k + 1) * Crow]
    opT = op.transpose()
    b1, b2 = G(K - 1), None                     # b_k, b_{k+1} as the loop walks k = K-1 .. 1
    first = True
    for k in range(K - 1, 0, -1):
        if k < K - 1:
            b1, b2 = csr_hop(opT, b1, z=b2, alpha=2.0, beta=-1.0, z2=G(k), gamma=1.0), b1
        csr_sddmm(op, b1, T[k - 1], alpha=1.0 if k == 1 else 2.0, out=dval, accumulate=not first)
        first = False
    return dval


@_on_device
def _windows_forward(op, x3, W, bias, bias_kind, mode):
    """x3 (S, n, T) fp32 contiguous, W (K, H, N) in the WORKING basis (folded for MODE_POWER) -> (out, stack (K, S, n, T))"""
    L = _lib.lib()
    S, n, T = x3.shape
    K, H, N = W.shape
    nwin = T - H + 1
    stack = _monomial_stack(op, x3, K) if mode == MODE_POWER else cheb_stack(op, x3, K, MODE_CHEBYSHEV)   # (K, S, n, T)
    out = torch.empty((S * nwin, n, N), dtype=torch.float32, device=x3.device)
    W2 = W.reshape(K * H, N).contiguous()
    b = bias.contiguous() if bias is not None else None
    assert K <= 32, "more than 32 hops: chunk the projection"
    for s in range(S):
        ptrs = (C.c_void_p * K)(*[stack[k, s].data_ptr() for k in range(K)])
        _lib.check(L.tgcn_cheb_project_windows_f32(_lib.stream_ptr(), n, T, H, N, K, ptrs, _lib.ptr(W2), _lib.ptr(b), bias_kind,
                                                   _lib.ptr(out[s * nwin:])))
    return out, stack


class ChebWindowsFn(torch.autograd.Function):
    """Streaming time-window layer with its backward: the hops run once on the T columns of every recording in both directions.
    d series = sum_k T_k(L^T) G_k with G_k the per-term input gradients of the window projection (tgcn_cheb_windows_backward_f32),
    folded by Horner (mode 0) / Clenshaw (mode 1) hops on L^T exactly as in layer_backward; dW from the same entry point."""

    @staticmethod
    @_on_device
    def forward(ctx, series, weight_khg, bias, op, mode, bias_kind):
        x3 = series.float().contiguous()
        W = weight_khg.float().contiguous()
        K = W.shape[0]
        fold = power_fold_matrix(K, W.device) if (mode == MODE_POWER and K > 2) else None
        Wt = fold_weight(fold, W) if fold is not None else W
        out, stack = _windows_forward(op, x3, Wt, bias, bias_kind, mode)
        ctx.save_for_backward(x3, Wt)
        ctx.stack = stack if ctx.needs_input_grad[1] else None       # the basis the weight gradient contracts with g
        ctx.op, ctx.mode, ctx.fold, ctx.bias_kind = op, mode, fold, bias_kind
        ctx.bias_shape = None if bias is None else bias.shape
        return out

    @staticmethod
    @_on_device
    def backward(ctx, g):
        x3, Wt = ctx.saved_tensors
        S, n, T = x3.shape
        K, H, N = Wt.shape
        nwin = T - H + 1
        L = _lib.lib()
        g = g.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        G = torch.empty((K, S, n, T), dtype=torch.float32, device=g.device) if need_x else None
        dW = torch.empty((K, H, N), dtype=torch.float32, device=g.device) if need_w else None
        ws = torch.empty(max(L.tgcn_cheb_windows_wgrad_workspace_bytes(S, n, T, H, N, K), 16), dtype=torch.uint8, device=g.device)
        _lib.check(L.tgcn_cheb_windows_backward_f32(_lib.stream_ptr(), S, n, T, H, N, K, _lib.ptr(ctx.stack), _lib.ptr(g),
                                                    _lib.ptr(Wt.reshape(K * H, N)), _lib.ptr(G), _lib.ptr(dW), _lib.ptr(ws), ws.numel()))
        ctx.stack = None
        gx = gb = None
        if need_x:
            opT = ctx.op.transpose()
            if K == 1:
                gx = G[0]
            elif ctx.mode == MODE_POWER:                              # Horner: b = G_j + L^T b
                b = G[K - 1]
                for j in range(K - 2, -1, -1):
                    b = csr_hop(opT, b, z=G[j], alpha=1.0, beta=1.0)
                gx = b
            else:                                                     # Clenshaw on L^T
                b1, b2 = G[K - 1], None
                for k in range(K - 2, 0, -1):
                    t = csr_hop(opT, b1, z=b2, alpha=2.0, beta=-1.0, z2=G[k], gamma=1.0)
                    b1, b2 = t, b1
                gx = csr_hop(opT, b1, z=b2, alpha=1.0, beta=-1.0, z2=G[0], gamma=1.0)
        if need_w and ctx.fold is not None:
            dW = fold_weight(ctx.fold, dW, transpose=True)
        if ctx.bias_shape is not None and ctx.needs_input_grad[2]:
            g4 = g.view(S * nwin, n, N)
            gb = (g4.sum(dim=(0, 1)) if ctx.bias_kind == BIAS_CHANNEL else g4.sum(dim=0)).reshape(ctx.bias_shape)
        return gx, dW, gb, None, None, None


def cheb_time_windows(op, series, weight_khg, bias, bias_kind, mode=MODE_POWER):
    """Streaming form of TGCNCheb_H / ChebTimeConv on sliding windows (f = 1): series (S, n, T) raw recordings,
    weight (K, H, N) in the reference basis.  Returns (S * (T-H+1), n, N), identical to running the layer on the
    windowed batch x[s*(T-H+1) + w, i, h] = series[s, i, w + h] (load/data_hcp.py:146-152), but the K-1 hops run
    once on the T columns of each recording -- in the backward too (ChebWindowsFn), so the windows it replaces can be
    trained through."""
    _lib.require_device(series, weight_khg, bias)
    if op.perm is not None:        # reordered operand: its hops work in their own labels (differentiable index ops, as in cheb_layer)
        series = relabel_rows(series, op.perm, op.inv_perm)
        if bias is not None and bias_kind == BIAS_VERTEX_CHANNEL:
            bias = relabel_rows(bias.reshape(op.n, -1), op.perm, op.inv_perm, dim=0)
        return relabel_rows(ChebWindowsFn.apply(series, weight_khg, bias, op, mode, bias_kind), op.inv_perm, op.perm)
    return ChebWindowsFn.apply(series, weight_khg, bias, op, mode, bias_kind)


@_on_device
def cheb_wgrad(terms, g2d):
    """dW[t] = terms[t]^T @ g2d  ->  (T, Kc, N); terms: list of (M, Kc) views with contiguous rows, g2d: (M, N)."""
    _lib.require_device(g2d, *terms)
    L = _lib.lib()
    T = len(terms)
    M, Kc = terms[0].shape
    N = g2d.shape[1]
    dW = torch.empty((T, Kc, N), dtype=torch.float32, device=g2d.device)
    for t0 in range(0, T, 32):
        nt = min(32, T - t0)
        ws_bytes = L.tgcn_cheb_wgrad_workspace_bytes(M, Kc, N, nt)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=g2d.device)
        a = (C.c_void_p * nt)(*[terms[t0 + i].data_ptr() for i in range(nt)])
        lda = (C.c_int64 * nt)(*[terms[t0 + i].stride(0) for i in range(nt)])
        _lib.check(L.tgcn_cheb_wgrad_f32(_lib.stream_ptr(), M, Kc, N, nt, a, lda, _lib.ptr(g2d), g2d.stride(0),
                                         _lib.ptr(dW[t0:]), _lib.ptr(ws), ws.numel()))
    return dW


def choose_layout(q, n, C_row):
    """layout 1 (re-lay x to one long row per vertex) when per-sample rows are short: gathers then move
    q*C contiguous floats per neighbour instead of C."""
    return 1 if (C_row < 32 and q > 1) else 0


def choose_q_chunk(q, n, C_row):
    """Samples per pass on layout 0: operands that overflow the caches anyway go one sample at a time, so a
    pass's gather working set is one (n, C) slab (and the hop workspace stays small)."""
    per_q = n * C_row * 4
    return int(max(1, min(q, (64 << 20) // max(per_q, 1))))


@_on_device
def cheb_forward_raw(op, x3, Wt, bias, bias_kind, mode, K, layout=None, q_chunk=None):
    """Fused layer forward through tgcn_cheb_forward_f32.  x3: (q, n, C) contiguous; Wt: (K*C, N)."""
    _lib.require_device(x3, Wt, bias)
    L = _lib.lib()
    q, n, Crow = x3.shape
    N = Wt.shape[1]
    assert x3.is_contiguous() and Wt.is_contiguous() and Wt.shape[0] == K * Crow and n == op.n
    if x3.data_ptr() % 16:          # a batch slice data[i:i+bs] of odd-width rows may start at any float: the driver wants 16 bytes
        x3 = x3.clone()
    if layout is None:
        layout = choose_layout(q, n, Crow)
    if q_chunk is None:
        q_chunk = choose_q_chunk(q, n, Crow)
    hop_C = q * Crow if layout == 1 else Crow
    sched = op.schedule_for(hop_C, hop_C % 4 == 0)
    ws_bytes = L.tgcn_cheb_forward_workspace_bytes(C.byref(sched.struct), K, q, n, Crow, layout, q_chunk)
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=x3.device)
    out = torch.empty((q, n, N), dtype=torch.float32, device=x3.device)
    _lib.check(L.tgcn_cheb_forward_f32(_lib.stream_ptr(), C.byref(op.struct), C.byref(sched.struct), mode, K, q, n, Crow,
                                       N, _lib.ptr(x3), _lib.ptr(Wt), _lib.ptr(bias), bias_kind, _lib.ptr(out), layout,
                                       q_chunk, _lib.ptr(ws), ws.numel()))
    return out


@_on_device
def cheb_forward_compact(plan, x3, Wt, bias, bias_kind, K, q_chunk=None, mode=MODE_POWER, W_left=None, keep=False):
    """Layer forward on an operand with left-out vertices (graph.CompactPlan) through ONE call of tgcn_cheb_compact_layer_f32: hop tensors only
    for the kept vertices, both recurrences.  x3: (q, n, C) contiguous; Wt: (K*C, N) in the working basis (folded for MODE_POWER); W_left:
    (C, N) matrix of the left-out vertices (MODE_CHEBYSHEV: W_0 - W_2 + W_4 - ..., left_out_weight; MODE_POWER: None = W'_0).
    keep=True: the hop tensors are written into one caller-owned buffer and returned as the layer's basis -- the list compact_terms gives
    ([x3, P_1, ..] for MODE_POWER, [T_0, T_1, ..] for MODE_CHEBYSHEV; compact terms are (q, n_c + 1, C) with a zero last row) -> (out, terms)."""
    _lib.require_device(x3, Wt, bias, W_left)
    L = _lib.lib()
    q, n, Crow = x3.shape
    N = Wt.shape[1]
    assert x3.is_contiguous() and Wt.is_contiguous() and Wt.shape[0] == K * Crow and n == plan.n and K >= 2
    if x3.data_ptr() % 16:
        x3 = x3.clone()
    auto_chunk = False
    if q_chunk is None:
        q_chunk = COMPACT_Q_CHUNK
    if q_chunk is None:
        # time steps per pass: the hops still run one time step per launch; a pass's projection reads a per-vertex bias ONCE for all its
        # time steps (the streaming projection keeps the bias row of a tile in registers across the samples of the pass).  cfg5 (round 5, same
        # box): 282.3 / 279.8 / 278.6 ms per forward at 4 / 8 / 16 time steps per pass -- 1.3 % for 77 GB of workspace at 16, so the K-1 compact
        # hop tensors of a pass are held to COMPACT_WS_FRACTION (a quarter) of the memory that is free when the shape is first seen (cfg5: 8 per
        # pass, 39 GB), in passes of equal size; an allocation that fails later (the process has allocated more since, or shares the GPU) halves
        # the choice instead of failing the forward
        q_chunk = 1
        if bias_kind == BIAS_VERTEX_CHANNEL and q > 1:
            key = (K, Crow, q)
            q_chunk = plan.q_chunk_cache.get(key)
            if q_chunk is None:            # asked once per shape: no host query on later forwards (nor under hipGraph capture)
                free = torch.cuda.mem_get_info(x3.device)[0]
                per_q = (K - 1) * (plan.n_c + 1) * Crow * 4
                most = int(max(1, min(q, 16, int(free * COMPACT_WS_FRACTION) // max(per_q, 1))))
                passes = -(-q // most)
                q_chunk = plan.q_chunk_cache[key] = -(-q // passes)
            auto_chunk = True
    sched = plan.schedule_for(Crow, Crow % 4 == 0)
    kept = None
    if keep:
        nt = K if mode == MODE_CHEBYSHEV else K - 1
        kept = torch.empty((nt, q, plan.n_c + 1, Crow), dtype=torch.float32, device=x3.device)
    out = torch.empty((q, n, N), dtype=torch.float32, device=x3.device)
    while True:
        ws_bytes = L.tgcn_cheb_compact_layer_workspace_bytes(C.byref(sched.struct), mode, K, q, plan.n_c, Crow, q_chunk, 1 if keep else 0)
        try:
            ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=x3.device)
            break
        except torch.OutOfMemoryError:
            if not auto_chunk or q_chunk <= 1:
                raise
            torch.cuda.empty_cache()
            q_chunk = plan.q_chunk_cache[(K, Crow, q)] = max(1, q_chunk // 2)        # the cached choice no longer fits: fewer time steps per pass from now on
    _lib.check(L.tgcn_cheb_compact_layer_f32(_lib.stream_ptr(), C.byref(plan.first.struct), C.byref(plan.rest.struct), C.byref(sched.struct), mode, K, q, n,
                                             Crow, N, _lib.ptr(x3), _lib.ptr(Wt), _lib.ptr(W_left), _lib.ptr(bias), bias_kind, _lib.ptr(out),
                                             _lib.ptr(plan.rows), _lib.ptr(plan.empty), plan.n_empty, _lib.ptr(plan.cid), q_chunk, _lib.ptr(kept),
                                             _lib.ptr(ws), ws.numel()))
    if not keep:
        return out
    terms = ([x3] if mode == MODE_POWER else []) + [kept[i] for i in range(kept.shape[0])]
    return out, terms


COMPACT_LAYOUT1 = False           # compact hop tensors for the vertex-major layout too: built and tested, measured on cfg5n (TGCNCheb(L,1,64,5), q = 16 on the
                                  # 10 M-vertex R-MAT): hop 2.08 -> 2.08 ms (64-byte rows: the launch is bound by its 160 M line requests, not by the rows it
                                  # writes), row-mapped projection 10.3 -> 12.2 ms (41 GB of output through a row map instead of one stream) => off
COMPACT_WS_FRACTION = 0.25       # share of the free device memory the compact hop tensors of one pass may take (cheb_forward_compact)
COMPACT_SLAB_BYTES = 64 << 20    # a hop launch covers one sample when a sample's (n_c, C) slab is larger (the gather working set stays one slab)


def compact_plan_for(op, mode, K, q, n, C_row, N=None):
    """The graph.CompactPlan a layer of this shape runs on, or None: square operands with enough vertices to leave out of the hop tensors
    (graph.GraphOperand.compact_plan), 2 <= K <= 32; both row layouts (the vertex-major layout 1 of short per-sample rows runs its hops on
    (n_c, q*C) rows).  Mode 0 keeps the rows with entries, mode 1 also every referenced vertex (closed form T_k[i] = x[i], 0, -x[i], ...
    for the isolated rest)."""
    if not COMPACT or not (2 <= K <= 32) or op.n != op.n_cols:
        return None
    lay1 = choose_layout(q, n, C_row) == 1
    if lay1 and not (COMPACT_LAYOUT1 and N is not None and K * C_row <= 16 and N % 4 == 0 and N <= 1024):
        return None       # the vertex-major form needs the vector-ALU projection (row map + interleave: a few scalars per row)
    plan = op.compact_plan("rows" if mode == MODE_POWER else "closed")
    if plan is not None and lay1 and (plan.n_c * q < 4096 or 0 < plan.n_empty * q < 4096):
        return None       # ... which takes problems of at least 4096 rows: both row classes must qualify (project_choose in the library)
    return plan


def _compact_buffer(plan, q, C_row, device):
    """(q, n_c + 1, C): row n_c of every sample is the zero row entries pointing at a left-out vertex gather from"""
    t = torch.empty((q, plan.n_c + 1, C_row), dtype=torch.float32, device=device)
    t[:, plan.n_c].zero_()
    return t


def _gather_rows(src3, rows64, plan, out=None):
    """src3[:, rows] into a compact buffer (tgcn_pack_rows_f32; one contiguous (n_c, C) block per sample)"""
    out = _compact_buffer(plan, src3.shape[0], src3.shape[2], src3.device) if out is None else out
    for b in range(src3.shape[0]):
        pack_rows(src3[b], rows64, out[b, : plan.n_c])
    return out


def _compact_hop(op, X, Y, plan, z=None, alpha=1.0, beta=0.0):
    """Y[:, :n_c] = alpha * op X + beta * z[:, :n_c]; one launch per sample when a sample's slab is beyond COMPACT_SLAB_BYTES"""
    q = X.shape[0]
    n_c = plan.n_c
    per_sample = q > 1 and n_c * X.shape[2] * 4 > COMPACT_SLAB_BYTES
    for sl in ([slice(b, b + 1) for b in range(q)] if per_sample else [slice(0, q)]):
        csr_hop(op, X[sl], z=None if z is None else z[sl, :n_c], alpha=alpha, beta=beta, out=Y[sl, :n_c])


@_on_device
def compact_terms(plan, x3, K, mode):
    """The K terms of the layer's basis with hop tensors for the plan's n_c kept vertices only (a backward without a kept basis recomputes them
    here; the forward itself is one driver call, cheb_forward_compact).
    mode 0: [x3, P_1, ..., P_{K-1}], P_k = L^k x (monomials: the basis of the FOLDED weight); term 0 is x itself in the caller's labels.
    mode 1: [T_0, ..., T_{K-1}] Chebyshev, every term compact (T_0 = x packed to the kept rows).
    Compact terms are views of ONE (T, q, n_c + 1, C) buffer whose last rows are zero."""
    q, n, Crow = x3.shape
    nt = K if mode == MODE_CHEBYSHEV else K - 1
    buf = torch.empty((max(nt, 1), q, plan.n_c + 1, Crow), dtype=torch.float32, device=x3.device)
    buf[:, :, plan.n_c].zero_()
    if mode == MODE_POWER:
        terms = [x3]
        for k in range(1, K):
            _compact_hop(plan.first if k == 1 else plan.rest, terms[k - 1], buf[k - 1], plan)
            terms.append(buf[k - 1])
        return terms
    terms = [_gather_rows(x3, plan.rows64(), plan, out=buf[0])]
    for k in range(1, K):
        if k == 1:
            _compact_hop(plan.rest, terms[0], buf[1], plan)
        else:
            _compact_hop(plan.rest, terms[k - 1], buf[k], plan, z=terms[k - 2], alpha=2.0, beta=-1.0)
        terms.append(buf[k])
    return terms


_sign_cache = {}


def _left_out_fold(K, device):
    """(K, K) fold matrix whose column 0 holds the signs +1, 0, -1, 0, ... of T_k of an isolated vertex: tgcn_fold_weight_f32 with it puts
    W_0 - W_2 + W_4 - ... into out[0]"""
    key = (K, str(device))
    with _fold_lock:
        m = _sign_cache.get(key)
        if m is None:
            m = torch.zeros(K, K, dtype=torch.float32)
            for k in range(0, K, 2):
                m[k, 0] = 1.0 if k % 4 == 0 else -1.0
            m = _sign_cache[key] = m.to(device)
    return m


def left_out_weight(W_kcn, mode):
    """(C, N) matrix of the vertices a plan leaves out: out[i] = x[i] @ this + bias.  mode 0 (W in the monomial basis): W'_0 -- every
    P_k[i], k >= 1, is zero.  mode 1: W_0 - W_2 + W_4 - ... -- T_k[i] = x[i], 0, -x[i], 0, ... for an isolated vertex (the signed sum
    runs in tgcn_fold_weight_f32)."""
    W_kcn = W_kcn.contiguous()
    if mode == MODE_POWER:
        return W_kcn[0]
    return fold_weight(_left_out_fold(W_kcn.shape[0], W_kcn.device), W_kcn)[0]


@_on_device
def project_mapped(terms, term_bs, W2d, bias, bias_kind, n_vertices, rowmap, mapped_terms, q, out, interleave=1):
    """out[b, rowmap[m]] = sum_t terms[t][b, row_t(m)] @ W[t] + bias through tgcn_cheb_project_mapped_f32; terms: tensors whose sample b starts
    term_bs[t] floats after sample b-1; W2d: (T*Kc, N); out: (q, n_vertices, N) contiguous.
    interleave = q > 1: vertex-major terms ((rows, q, Kc): the samples of a vertex are consecutive rows), one launch for all samples."""
    L = _lib.lib()
    T = len(terms)
    Kc = W2d.shape[0] // T
    N = W2d.shape[1]
    M = int(rowmap.numel()) * interleave
    if M == 0:
        return out
    a = (C.c_void_p * T)(*[t.data_ptr() for t in terms])
    lda = (C.c_int64 * T)(*[Kc] * T)
    a_bs = (C.c_int64 * T)(*term_bs)
    _lib.check(L.tgcn_cheb_project_mapped_f32(_lib.stream_ptr(), M, Kc, N, T, a, lda, _lib.ptr(W2d), _lib.ptr(bias), bias_kind, n_vertices, interleave,
                                              _lib.ptr(rowmap), mapped_terms, 1 if interleave > 1 else q, a_bs, n_vertices * N, _lib.ptr(out), N))
    return out


@_on_device
def compact_forward(plan, x3, Wt_kcn, bias, bias_kind, mode, keep=True):
    """Layer forward with compact hop tensors that also hands back the basis (training forward, and the Chebyshev-recurrence classes): ONE
    driver call (cheb_forward_compact -> tgcn_cheb_compact_layer_f32; round 4 issued the hops and the two row-mapped projections from here).
    Wt_kcn: (K, C, N) in the WORKING basis (folded for mode 0).  -> (out (q, n, N), terms or None)"""
    _lib.require_device(x3, Wt_kcn, bias)
    q, n, Crow = x3.shape
    K, _, N = Wt_kcn.shape
    assert x3.is_contiguous() and n == plan.n and 2 <= K <= 32
    if x3.data_ptr() % 16:
        x3 = x3.clone()
    b = bias.contiguous() if bias is not None else None
    if choose_layout(q, n, Crow) == 1:
        # short per-sample rows (COMPACT_LAYOUT1, off by default: slower on cfg5n): one long row per vertex for the gathers, (n, q*C) -- the hop
        # tensors are (n_c + 1, q*C), the projections read them as (vertex, sample) rows of C floats and write the sample-major output through
        # the row map (interleave = q); primitive calls of the library
        xt = relayout_qnc_to_nqc(x3).view(1, n, q * Crow)
        tt = compact_terms(plan, xt, K, mode)
        out = torch.empty((q, n, N), dtype=torch.float32, device=x3.device)
        W2 = Wt_kcn.reshape(K * Crow, N).contiguous()
        project_mapped(tt, [0] * K, W2, b, bias_kind, n, plan.rows, 1 if mode == MODE_POWER else 0, q, out, interleave=q)
        if plan.n_empty:
            project_mapped([xt], [0], left_out_weight(Wt_kcn, mode), b, bias_kind, n, plan.empty, 1, q, out, interleave=q)
        return out, None
    W_left = left_out_weight(Wt_kcn, mode) if (mode == MODE_CHEBYSHEV and plan.n_empty) else None
    res = cheb_forward_compact(plan, x3, Wt_kcn.reshape(K * Crow, N).contiguous(), b, bias_kind, K, mode=mode, W_left=W_left, keep=keep)
    return res if keep else (res, None)


@_on_device
def compact_wgrad(plan, x3, terms, g, mode):
    """dW (K, C, N) in the working basis from compact terms: the kept rows of g are gathered once (g_c), term k >= 1 contracts with them;
    the left-out vertices enter through x only -- term 0 for mode 0, and with alternating signs at every even k for mode 1 (T_k[i] = +-x[i])."""
    q, n, Crow = x3.shape
    N = g.shape[2]
    K = len(terms)
    g_c = _gather_rows(g, plan.rows64(), plan)                               # (q, n_c + 1, N), zero last row
    gc2 = g_c.reshape(q * (plan.n_c + 1), N)
    S_all = cheb_wgrad([x3.reshape(q * n, Crow)], g.reshape(q * n, N))[0]     # x^T g over every vertex
    flat = lambda t: t.reshape(q * (plan.n_c + 1), Crow)
    if mode == MODE_POWER:
        rest = cheb_wgrad([flat(t) for t in terms[1:]], gc2) if K > 1 else None
        return torch.cat([S_all.unsqueeze(0), rest]) if K > 1 else S_all.unsqueeze(0)
    dW = cheb_wgrad([flat(t) for t in terms], gc2)
    S_out = S_all - dW[0]                                                      # x^T g over the isolated vertices only
    for k in range(0, K, 2):
        dW[k] += S_out if k % 4 == 0 else -S_out
    return dW


@_on_device
def cheb_forward_pool(op, x3, Wt, bias, bias_kind, mode, K, pool, z, idx, layout=None, q_chunk=None):
    """relu + max-pool fused layer on the hops-then-projection path (tgcn_cheb_forward_pool_f32) into z (q, n/pool, N) and the
    arg-max bytes idx.  x3: (q, n, C) contiguous; Wt: (K*C, N) in the working basis."""
    _lib.require_device(x3, Wt, bias, z, idx)
    L = _lib.lib()
    q, n, Crow = x3.shape
    N = Wt.shape[1]
    assert x3.is_contiguous() and Wt.is_contiguous() and Wt.shape[0] == K * Crow and n == op.n and z.is_contiguous() and idx.is_contiguous()
    if x3.data_ptr() % 16:
        x3 = x3.clone()
    if layout is None:
        layout = choose_layout(q, n, Crow)
    if q_chunk is None:
        q_chunk = choose_q_chunk(q, n, Crow)
    hop_C = q * Crow if layout == 1 else Crow
    sched = op.schedule_for(hop_C, hop_C % 4 == 0)
    ws_bytes = L.tgcn_cheb_forward_pool_workspace_bytes(C.byref(sched.struct), K, q, n, Crow, N, layout, q_chunk, pool)
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=x3.device)
    _lib.check(L.tgcn_cheb_forward_pool_f32(_lib.stream_ptr(), C.byref(op.struct), C.byref(sched.struct), mode, K, q, n, Crow, N,
                                            _lib.ptr(x3), _lib.ptr(Wt), _lib.ptr(bias), bias_kind, pool, _lib.ptr(z), _lib.ptr(idx),
                                            layout, q_chunk, _lib.ptr(ws), ws.numel()))
    return z


def pool_epilogue_is_fused(op, q, Crow, N, K, pool, layout=None, q_chunk=None):
    """True when cheb_forward_pool runs relu + pool inside the projection kernel for this shape (no scratch for the layer output)."""
    L = _lib.lib()
    layout = choose_layout(q, op.n, Crow) if layout is None else layout
    q_chunk = choose_q_chunk(q, op.n, Crow) if q_chunk is None else q_chunk
    hop_C = q * Crow if layout == 1 else Crow
    sched = op.schedule_for(hop_C, hop_C % 4 == 0)
    base = L.tgcn_cheb_forward_workspace_bytes(C.byref(sched.struct), K, q, op.n, Crow, layout, q_chunk)
    return L.tgcn_cheb_forward_pool_workspace_bytes(C.byref(sched.struct), K, q, op.n, Crow, N, layout, q_chunk, pool) < base + q * op.n * N * 4


@_on_device
def cheb_stack(op, x3, K, mode, _operand_labels=False):
    """The (K, q, n, C) stack `_chebyshev` / `_time_chebyshev` return (gcn.py:52-79,126-154,208-237), or the
    true-recurrence stack for mode 1.  Materialising path: every hop writes its slice of the stack."""
    _lib.require_device(x3)
    if op.perm is not None and not _operand_labels:       # reordered operand: relabel on the way in and out (the shared operand is never touched)
        return relabel_rows(cheb_stack(op, relabel_rows(x3, op.perm, op.inv_perm), K, mode, _operand_labels=True), op.inv_perm, op.perm, dim=2)
    q, n, Crow = x3.shape
    st = torch.empty((K, q, n, Crow), dtype=torch.float32, device=x3.device)
    st[0].copy_(x3)
    if K > 1:
        csr_hop(op, st[0], out=st[1])
    p_prev = st[1] if K > 1 else None
    for k in range(2, K):
        if mode == MODE_POWER:      # P_k = L P_{k-1} (hidden chain), Xt[k] = 2 P_k - Xt[k-2]
            _, p_prev = csr_hop(op, p_prev, z=st[k - 2], alpha=2.0, beta=-1.0, want_p=True, out=st[k])
        else:
            csr_hop(op, st[k - 1], z=st[k - 2], alpha=2.0, beta=-1.0, out=st[k])
    return st


_fold_cache = {}
_fold_lock = threading.Lock()     # shared by the threads of nn.DataParallel replicas


def power_fold_matrix(K, device=None, dtype=torch.float32):
    key = (K, str(device), dtype)
    with _fold_lock:
        c = _fold_cache.get(key)
        if c is None:
            c = _fold_cache[key] = _power_fold_matrix(K, device, dtype)
    return c


def _power_fold_matrix(K, device=None, dtype=torch.float32):
    """c[k, j] with Xt[k] = sum_j c[k, j] L^j x for the reference_power recursion: c[0]=e0, c[1]=e1,
    c[k] = 2 e_k - c[k-2].  Folding it into the weight (W'_j = sum_k c[k,j] W_k) turns the layer into one
    monomial chain: no subtrahend reads, no second output (verified there to <= 6.3e-7)."""
    c = torch.zeros(K, K, dtype=torch.float64)
    for k in range(K):
        c[k, k] = 1.0 if k < 2 else 2.0
        if k >= 2:
            c[k] -= c[k - 2]
    return c.to(device=device, dtype=dtype)


@_on_device
def fold_weight(fold, W, transpose=False):
    """W'[j] = sum_k fold[k, j] W[k]  (transpose: sum_k fold[j, k] W[k]) for a (K, C, N) weight, in libtgcn_hip.so: the
    forward path issues no vendor-library GEMM"""
    _lib.require_device(fold, W)
    K = W.shape[0]
    Wc = W.contiguous()
    out = torch.empty_like(Wc)
    _lib.check(_lib.lib().tgcn_fold_weight_f32(_lib.stream_ptr(), K, Wc.numel() // K, _lib.ptr(fold), _lib.ptr(Wc), _lib.ptr(out),
                                               1 if transpose else 0))
    return out


# ----------------------------------------------------------------------------------------- autograd
def small_path_tile(op, C_row, mode, pool=False):
    """Channel tile (16 / 8) of the one-launch LDS-resident kernel, or 0 when the shape does not fit it
    (pool: with the fused relu + pool epilogue)."""
    if op.n_cols != op.n or not SMALL_PATH:
        return 0
    fn = _lib.lib().tgcn_cheb_forward_small_pool_supported if pool else _lib.lib().tgcn_cheb_forward_small_supported
    return fn(op.n, op.nnz, int(C_row), int(mode))


@_on_device
def cheb_forward_small(op, x3, W_kcn, fold, bias, bias_kind, mode):
    """Whole layer in one launch (tgcn_cheb_forward_small_f32).  W_kcn: (K, C, N) RAW weight when `fold` is given."""
    _lib.require_device(x3, W_kcn, bias, fold)
    q, n, Crow = x3.shape
    K, _, N = W_kcn.shape
    out = torch.empty((q, n, N), dtype=torch.float32, device=x3.device)
    _lib.check(_lib.lib().tgcn_cheb_forward_small_f32(_lib.stream_ptr(), C.byref(op.struct), mode, K, q, Crow, N, _lib.ptr(x3),
                                                      _lib.ptr(W_kcn), _lib.ptr(fold), _lib.ptr(bias), bias_kind, _lib.ptr(out)))
    return out


def small_basis_tile(op, C_row, mode):
    """Channel tile of the one-launch basis kernel (weight gradient on small graphs), or 0 when the operand does not fit."""
    if op.n_cols != op.n or not SMALL_PATH:
        return 0
    return _lib.lib().tgcn_cheb_basis_small_supported(op.n, op.nnz, int(C_row), int(mode))


@_on_device
def cheb_basis_small(op, x3, K, mode):
    """Terms of the layer's basis as a list of K (q, n, C) tensors (term 0 is x3 itself): monomials L^k x for
    MODE_POWER (the basis of the folded weight), Chebyshev T_k x for MODE_CHEBYSHEV.  One launch."""
    _lib.require_device(x3)
    q, n, Crow = x3.shape
    if K == 1:
        return [x3]
    stack = torch.empty((K - 1, q, n, Crow), dtype=torch.float32, device=x3.device)
    # the kernel indexes terms k = 1 .. K-1 of a (K, q, n, C) stack: pass the address term 0 would have
    base = stack.data_ptr() - q * n * Crow * 4
    _lib.check(_lib.lib().tgcn_cheb_basis_small_f32(_lib.stream_ptr(), C.byref(op.struct), mode, K, q, Crow, _lib.ptr(x3), base))
    return [x3] + [stack[k] for k in range(K - 1)]


@_on_device
def cheb_forward_pf(op, x3, Wt_kcn, bias, bias_kind, mode):
    """Project-first form (tgcn_cheb_forward_pf_f32): ONE projection x . [W_0 | ... | W_{K-1}], then Horner / Clenshaw
    on the (q, n, N) results.  Wt_kcn: (K, C, N), already folded for MODE_POWER."""
    _lib.require_device(x3, Wt_kcn, bias)
    L = _lib.lib()
    q, n, Crow = x3.shape
    K, _, N = Wt_kcn.shape
    Wcat = weight_layout(Wt_kcn, 0)
    sched = op.schedule_for(N, N % 4 == 0)
    ws_bytes = L.tgcn_cheb_forward_pf_workspace_bytes(C.byref(sched.struct), K, q, n, N)
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=x3.device)
    out = torch.empty((q, n, N), dtype=torch.float32, device=x3.device)
    _lib.check(L.tgcn_cheb_forward_pf_f32(_lib.stream_ptr(), C.byref(op.struct), C.byref(sched.struct), mode, K, q, n, Crow, N,
                                          _lib.ptr(x3), _lib.ptr(Wcat), _lib.ptr(bias), bias_kind, _lib.ptr(out), _lib.ptr(ws),
                                          ws.numel()))
    return out


PROJECT_FIRST = True   # developer switch
COMPACT = True         # developer switch: compact hop tensors for operands with many empty rows (graph.CompactPlan)
COMPACT_Q_CHUNK = None # developer switch: time steps per pass of the compacted forward (None: chosen by cheb_forward_compact)


def use_project_first(q, n, C_row, N):
    """Hops move N floats per row instead of C: take it when the output is at most half as wide as the input row."""
    return PROJECT_FIRST and 2 * N <= C_row and q <= 65535


@_on_device
def layer_forward(op, x3, W, fold, b, bias_kind, mode):
    """Forward of the layer on whichever path fits the shape (all in libtgcn_hip.so): the one-launch LDS kernel for
    small graphs, project-first for wide inputs / narrow outputs, hops-then-projection otherwise."""
    K, Crow, N = W.shape
    if small_path_tile(op, Crow, mode):
        return cheb_forward_small(op, x3, W, fold, b, bias_kind, mode)
    Wt = fold_weight(fold, W) if fold is not None else W
    if use_project_first(x3.shape[0], x3.shape[1], Crow, N):
        return cheb_forward_pf(op, x3, Wt, b, bias_kind, mode)
    plan = compact_plan_for(op, mode, K, x3.shape[0], x3.shape[1], Crow, N)
    if plan is not None and mode == MODE_POWER and choose_layout(x3.shape[0], x3.shape[1], Crow) == 0:   # many structurally empty rows: compact hop tensors, one call
        return cheb_forward_compact(plan, x3, Wt.reshape(K * Crow, N).contiguous(), b, bias_kind, K)
    if plan is not None:                                   # the same for the Chebyshev recurrence (closed form for isolated vertices)
        return compact_forward(plan, x3, Wt, b, bias_kind, mode, keep=False)[0]
    return cheb_forward_raw(op, x3, Wt.reshape(K * Crow, N).contiguous(), b, bias_kind, mode, K)


@_on_device
def relayout_qnc_to_nqc(x3):
    """(q, n, C) -> (n, q, C) contiguous (tgcn_relayout_qnc_to_nqc_f32: LDS-tiled transpose for C <= 32, coalesced row copy beyond)."""
    q, n, Crow = x3.shape
    if q == 1:
        return x3.contiguous().view(n, 1, Crow)          # one sample: the same memory
    _lib.require_device(x3)
    out = torch.empty((n, q, Crow), dtype=torch.float32, device=x3.device)
    _lib.check(_lib.lib().tgcn_relayout_qnc_to_nqc_f32(_lib.stream_ptr(), _lib.ptr(x3.contiguous()), _lib.ptr(out), q, n, Crow))
    return out


KEEP_BASIS_BYTES = 2 << 30    # training on the hops-then-projection path keeps the hop tensors for the backward up to this size


@_on_device
def forward_keeping_basis(op, x3, Wt, bias, bias_kind, mode):
    """Hops-then-projection forward that hands the K hop tensors to the caller (they ARE the basis the weight gradient
    needs: monomials L^k x for the folded weight, Chebyshev T_k x for mode 1), instead of recomputing them in backward.
    Same kernels as tgcn_cheb_forward_f32, one call per hop.  -> (out (q, n, N), terms as (q*n, C) row views, row order)
    row order "nq": rows are (vertex, sample) -- layout 1, one long row per vertex for the gathers -- else (sample, vertex)."""
    q, n, Crow = x3.shape
    K, _, N = Wt.shape
    nq = choose_layout(q, n, Crow) == 1
    x0 = relayout_qnc_to_nqc(x3).view(1, n, q * Crow) if nq else x3
    terms = [x0]
    for k in range(1, K):
        if mode == MODE_POWER or k == 1:
            terms.append(csr_hop(op, terms[k - 1]))
        else:
            terms.append(csr_hop(op, terms[k - 1], z=terms[k - 2], alpha=2.0, beta=-1.0))
    rows = [t.reshape(q * n, Crow) for t in terms]
    out = cheb_project(rows, Wt, bias, bias_kind, n, interleave=q if nq else 1).view(q, n, N)
    return out, rows, nq


class ChebLayerFn(torch.autograd.Function):
    """out = sum_k T_k x W_k + bias with T_k given by `mode`; x3 (q,n,C), W (K, C, N) in the REFERENCE basis.
    For MODE_POWER the weight is folded to the monomial basis (W'_j = sum_k c[k,j] W_k): inside the kernel on the
    small-graph path, by tgcn_fold_weight_f32 otherwise; backward applies the transposed fold to the weight gradient."""

    @staticmethod
    @_on_device
    def forward(ctx, x3, W, bias, op, mode, bias_kind, grad_mode=True, values=None):
        # `values`: the operand's stored values in CSR order as a tensor of the autograd graph (learnable edge weights of ChebConv /
        # ChebTimeConv).  The forward computes with the values packed in `op` (equal by construction); the tensor only receives the gradient.
        if values is not None and (mode != MODE_CHEBYSHEV or op.perm is not None or values.numel() != op.nnz):
            raise _lib.TgcnError("learnable operand values: Chebyshev-recurrence layers on an operand in the caller's labels only")
        K, Crow, N = W.shape
        x3 = x3.contiguous()
        W = W.contiguous()
        b = bias.contiguous() if bias is not None else None
        fold = power_fold_matrix(K, W.device) if (mode == MODE_POWER and K > 2) else None
        ctx.basis = None
        general = not small_path_tile(op, Crow, mode) and not use_project_first(x3.shape[0], x3.shape[1], Crow, N)
        # grad_mode: whether the CALLER records gradients (inside forward() grad mode is always off, and needs_input_grad only
        # mirrors requires_grad): an inference call under torch.no_grad() keeps nothing for a backward that never comes
        plan = compact_plan_for(op, mode, K, x3.shape[0], x3.shape[1], Crow, N) if general else None
        lay0 = choose_layout(x3.shape[0], x3.shape[1], Crow) == 0
        if plan is not None and lay0 and grad_mode and ctx.needs_input_grad[1] and K * x3.shape[0] * (plan.n_c + 1) * Crow * 4 <= KEEP_BASIS_BYTES:
            # training forward on an operand with left-out vertices: compact hop tensors, kept for the weight gradient
            Wt = fold_weight(fold, W) if fold is not None else W
            out, terms = compact_forward(plan, x3, Wt, b, bias_kind, mode)
            ctx.basis = ("compact", plan, terms)
        elif general and plan is None and K > 1 and grad_mode and ctx.needs_input_grad[1] and K * x3.numel() * 4 <= KEEP_BASIS_BYTES:
            Wt = fold_weight(fold, W) if fold is not None else W
            out, rows, nq = forward_keeping_basis(op, x3, Wt, b, bias_kind, mode)
            ctx.basis = (rows, nq)
        else:
            out = layer_forward(op, x3, W, fold, b, bias_kind, mode)
        ctx.save_for_backward(x3, W, None if values is None else values.detach())
        ctx.values_epoch = op.values_epoch
        ctx.op, ctx.mode, ctx.bias_kind, ctx.fold = op, mode, bias_kind, fold
        ctx.bias_shape = None if bias is None else bias.shape
        return out

    @staticmethod
    @_on_device
    def backward(ctx, g):
        x3, W, vals = ctx.saved_tensors
        _values_guard(ctx.op, ctx.values_epoch, vals)
        gx, gW, gb = layer_backward(ctx.op, ctx.mode, ctx.fold, x3, W, g, ctx.bias_kind, ctx.bias_shape, ctx.needs_input_grad,
                                    basis=ctx.basis)
        values_basis = ctx.basis if (isinstance(ctx.basis, tuple) and len(ctx.basis) == 2 and ctx.mode == MODE_CHEBYSHEV) else None
        ctx.basis = None
        gv = None
        if len(ctx.needs_input_grad) > 7 and ctx.needs_input_grad[7]:
            kept = None
            if values_basis is not None and not values_basis[1]:             # hop tensors kept by the forward in (sample, vertex) row order: T_0 .. T_{K-1}
                q, n, Crow = x3.shape
                kept = [r.view(q, n, Crow) for r in values_basis[0]]
            gv = chebyshev_values_grad(ctx.op, x3, W, g, basis=kept)
        return gx, gW, gb, None, None, None, None, gv


@_on_device
def _monomial_stack(op, x3, K):
    st = torch.empty((K,) + tuple(x3.shape), dtype=torch.float32, device=x3.device)
    st[0].copy_(x3)
    for k in range(1, K):
        csr_hop(op, st[k - 1], out=st[k])
    return st


def _pad_rows(op, x3, weight_kcn, mode):
    """Rows whose length is not a multiple of 4 floats (the HCP horizon H = 15) take the scalar-load forms of the hop and
    projection kernels.  On the hops-then-projection path pad them with zero channels to the next multiple of 4 (zero
    rows in the weight): 7 % more hop traffic at C = 15, projection 0.37 -> 0.2 ms on the 59 k-vertex mesh shape.
    torch's pad is differentiable, so the gradients come back sliced."""
    C, N = x3.shape[2], weight_kcn.shape[2]
    if C % 4 == 0 or C < 7 or small_path_tile(op, C, mode) or use_project_first(x3.shape[0], x3.shape[1], C, N):
        return x3, weight_kcn
    pad = (-C) % 4
    return torch.nn.functional.pad(x3, (0, pad)), torch.nn.functional.pad(weight_kcn, (0, 0, 0, pad))


def _to_operand_labels(op, x3, bias, bias_kind):
    """A reordered operand (GraphOperand.reordered) works in its own vertex labels: gather the rows of x and of a per-vertex
    bias into them (differentiable torch index ops: plumbing, no arithmetic)."""
    if op.perm is None:
        return x3, bias
    x3 = relabel_rows(x3, op.perm, op.inv_perm)
    if bias is not None and bias_kind == BIAS_VERTEX_CHANNEL:
        bias = relabel_rows(bias.reshape(op.n, -1), op.perm, op.inv_perm, dim=0)
    return x3, bias


def cheb_layer(op, x3, weight_kcn, bias, bias_kind, mode, values=None):
    """Differentiable fused layer; weight_kcn: (K, C, N) in the reference basis.  values: the operand's values (CSR order) as an autograd
    tensor when they are learnable (nn._EdgeBase builds it from edge_weight)."""
    x3, weight_kcn = _pad_rows(op, x3, weight_kcn, mode)
    x3, bias = _to_operand_labels(op, x3, bias, bias_kind)
    out = ChebLayerFn.apply(x3, weight_kcn, bias, op, mode, bias_kind, torch.is_grad_enabled(), values)
    return out if op.perm is None else relabel_rows(out, op.inv_perm, op.perm)


@_on_device
def layer_backward(op, mode, fold, x3, W, g, bias_kind, bias_shape, needs, basis=None):
    """Gradients of the layer w.r.t. (x3, W, bias).  All contractions run in libtgcn_hip.so: the basis is recomputed
    with the hop kernel, dW is the MFMA weight-gradient kernel, G = g W^T is the projection kernel with the transposed
    weight, dx is Horner (mode 0) / Clenshaw (mode 1) on L^T with the hop kernel; only the bias reduction and the
    K x K fold of the weight gradient are torch ops.  Graphs that fit in LDS take two one-launch kernels instead of the
    2(K-1) hops: the basis kernel for dW and the forward kernel on L^T for dx."""
    K, Crow, N = W.shape
    q, n, _ = x3.shape
    Wt = W                                                            # the basis the kernels work in
    if fold is not None and needs[0]:
        Wt = fold_weight(fold, W)
    g = g.contiguous()
    g2d = g.reshape(q * n, N)
    gx = gW = gb = None
    general = not small_path_tile(op, Crow, mode) and not use_project_first(q, n, Crow, N)
    plan = compact_plan_for(op, mode, K, q, n, Crow, N) if general else None
    if needs[1] and (plan is not None or (basis is not None and basis[0] == "compact")):
        # operand with left-out vertices: the basis exists (kept by the forward, or recomputed here) for the kept vertices only
        if basis is not None and basis[0] == "compact":
            plan_b, terms = basis[1], basis[2]
        else:
            plan_b, terms = plan, compact_terms(plan, x3.contiguous(), K, mode)
        gW = compact_wgrad(plan_b, x3.contiguous(), terms, g, mode)
        if fold is not None:
            gW = fold_weight(fold, gW, transpose=True)
    elif needs[1] and basis is not None:                              # hop tensors kept by the forward (forward_keeping_basis)
        rows, nq = basis
        g_rows = relayout_qnc_to_nqc(g).view(q * n, N) if nq else g2d     # same (vertex, sample) row order as the terms
        gW = cheb_wgrad(rows, g_rows)
        if fold is not None:
            gW = fold_weight(fold, gW, transpose=True)
    elif needs[1]:
        x3c = x3.contiguous()
        if small_basis_tile(op, Crow, mode):                          # small graphs: the whole basis in one launch
            basis = cheb_basis_small(op, x3c, K, mode)
        else:
            basis = cheb_stack(op, x3c, K, MODE_CHEBYSHEV) if mode == MODE_CHEBYSHEV else _monomial_stack(op, x3c, K)
        gW = cheb_wgrad([basis[k].reshape(q * n, Crow) for k in range(K)], g2d)
        if fold is not None:                                          # back to the reference basis
            gW = fold_weight(fold, gW, transpose=True)
    if needs[0] and small_path_tile(op.transpose(), N, mode):
        # small graphs: dx = sum_j (L^T)^j g W_j^T is the one-launch forward kernel on (L^T, g, W^T)
        gx = cheb_forward_small(op.transpose(), g, weight_layout(Wt, 1), None, None, BIAS_NONE, mode)
    elif needs[0] and general and compact_plan_for(op.transpose(), mode, K, q, n, N, Crow) is not None:
        # dx = sum_k T_k(L^T) g W_k^T IS the layer on (L^T, g, W^T): with left-out vertices in L^T it runs on compact hop tensors too
        planT = compact_plan_for(op.transpose(), mode, K, q, n, N, Crow)
        WtT = weight_layout(Wt, 1)                                   # (K, N, C)
        if mode == MODE_POWER:
            gx = cheb_forward_compact(planT, g, WtT.reshape(K * N, Crow), None, BIAS_NONE, K)
        else:
            gx = compact_forward(planT, g, WtT, None, BIAS_NONE, mode, keep=False)[0]
    elif needs[0]:
        opT = op.transpose()
        # G[m, k*C + c] = sum_n g[m, n] W[k, c, n]: one projection with the (N, K*C) transposed weight
        Wcat = weight_layout(Wt, 2).view(1, N, K * Crow)
        Gall = cheb_project([g2d], Wcat, None, BIAS_NONE, n).view(q, n, K * Crow)
        G = [Gall[:, :, k * Crow:(k + 1) * Crow] for k in range(K)]        # strided views, rows contiguous
        if mode == MODE_POWER:                                       # Horner: b = G_j + L^T b
            b = G[K - 1]
            for j in range(K - 2, -1, -1):
                b = csr_hop(opT, b, z=G[j], alpha=1.0, beta=1.0)
            gx = b.contiguous()
        elif K == 1:
            gx = G[0].contiguous()
        else:                                                        # Clenshaw on L^T
            b1, b2 = G[K - 1], None                                   # b_{K-1} = G_{K-1} (b_K = b_{K+1} = 0)
            for k in range(K - 2, 0, -1):                             # b_k = G_k + 2 L^T b_{k+1} - b_{k+2}, one launch each
                t = csr_hop(opT, b1, z=b2, alpha=2.0, beta=-1.0, z2=G[k], gamma=1.0)
                b1, b2 = t, b1
            gx = csr_hop(opT, b1, z=b2, alpha=1.0, beta=-1.0, z2=G[0], gamma=1.0)   # dx = G_0 + L^T b_1 - b_2
    if bias_shape is not None and needs[2]:
        gb = g.sum(dim=(0, 1)) if bias_kind == BIAS_CHANNEL else g.sum(dim=0)
        gb = gb.reshape(bias_shape)
    return gx, gW, gb


class ChebReluPoolFn(torch.autograd.Function):
    """z = max over `pool` consecutive vertices of relu(layer(x)): the callers' `gcn_pool_4(F.relu(layer(x)))`
    (examples/pytorch_based/pytorch_hcp_tgcn.py:134-141) with the epilogue fused -- inside the one-launch kernel on
    small graphs (the layer output never reaches HBM), as one extra pass otherwise."""

    @staticmethod
    @_on_device
    def forward(ctx, x3, W, bias, op, mode, bias_kind, pool):
        K, Crow, N = W.shape
        x3 = x3.contiguous()
        W = W.contiguous()
        b = bias.contiguous() if bias is not None else None
        q, n, _ = x3.shape
        assert n % pool == 0, "pooling needs n divisible by the pool size"
        fold = power_fold_matrix(K, W.device) if (mode == MODE_POWER and K > 2) else None
        z = torch.empty((q, n // pool, N), dtype=torch.float32, device=x3.device)
        idx = torch.empty((q, n // pool, N), dtype=torch.uint8, device=x3.device)
        L = _lib.lib()
        # dense small operands run the layer on the matrix pipe (no fused epilogue there): layer + one relu/pool pass beats
        # the fused vector-ALU kernel (HCP shape: 136 + 15 us against 330 us)
        dense_mfma = op.dense is not None and (Crow <= 32 or (Crow <= 64 and op.n <= 128))
        if small_path_tile(op, Crow, mode, pool=True) and not dense_mfma:
            _lib.check(L.tgcn_cheb_forward_small_pool_f32(_lib.stream_ptr(), C.byref(op.struct), mode, K, q, Crow, N, _lib.ptr(x3),
                                                          _lib.ptr(W), _lib.ptr(fold), _lib.ptr(b), bias_kind, 1, pool,
                                                          _lib.ptr(z), _lib.ptr(idx)))
        elif (not small_path_tile(op, Crow, mode) and not use_project_first(q, n, Crow, N)
              and compact_plan_for(op, mode, K, q, n, Crow, N) is None):
            # hops-then-projection path: bias + relu + max over `pool` vertices inside the projection's epilogue where the shape allows
            # (tgcn_cheb_forward_pool_f32: the (q, n, N) layer output is then never written), one extra pass over scratch otherwise
            Wt = fold_weight(fold, W) if fold is not None else W
            cheb_forward_pool(op, x3, Wt.reshape(K * Crow, N).contiguous(), b, bias_kind, mode, K, pool, z, idx)
        else:
            y = layer_forward(op, x3, W, fold, b, bias_kind, mode)
            _lib.check(L.tgcn_relu_pool_f32(_lib.stream_ptr(), _lib.ptr(y), _lib.ptr(z), _lib.ptr(idx), q, n, N, pool))
        ctx.save_for_backward(x3, W, z, idx)
        ctx.op, ctx.mode, ctx.bias_kind, ctx.fold, ctx.pool = op, mode, bias_kind, fold, pool
        ctx.bias_shape = None if bias is None else bias.shape
        return z

    @staticmethod
    @_on_device
    def backward(ctx, gz):
        x3, W, z, idx = ctx.saved_tensors
        q, n, _ = x3.shape
        N = W.shape[2]
        gy = torch.empty((q, n, N), dtype=torch.float32, device=gz.device)
        _lib.check(_lib.lib().tgcn_relu_pool_bwd_f32(_lib.stream_ptr(), _lib.ptr(gz.contiguous()), _lib.ptr(z), _lib.ptr(idx),
                                                     _lib.ptr(gy), q, n, N, ctx.pool))
        gx, gW, gb = layer_backward(ctx.op, ctx.mode, ctx.fold, x3, W, gy, ctx.bias_kind, ctx.bias_shape, ctx.needs_input_grad)
        return gx, gW, gb, None, None, None, None


def cheb_relu_pool(op, x3, weight_kcn, bias, bias_kind, mode, pool):
    """Differentiable relu + max-pool fused layer; weight in the reference basis."""
    x3, weight_kcn = _pad_rows(op, x3, weight_kcn, mode)
    if op.perm is not None:
        # pooling groups consecutive vertices of the CALLER's labelling, which are scattered rows of a reordered operand: the
        # epilogue cannot be fused there -- the layer's output is relabelled back first, then one relu + pool pass (HIP)
        return ReluPoolFn.apply(cheb_layer(op, x3, weight_kcn, bias, bias_kind, mode), pool)
    return ChebReluPoolFn.apply(x3, weight_kcn, bias, op, mode, bias_kind, pool)


class ReluPoolFn(torch.autograd.Function):
    """z = max over `pool` consecutive vertices of relu(y) as its own pass (tgcn_relu_pool_f32 / _bwd)."""

    @staticmethod
    @_on_device
    def forward(ctx, y, pool):
        _lib.require_device(y)
        y = y.float().contiguous()
        q, n, N = y.shape
        assert n % pool == 0, "pooling needs n divisible by the pool size"
        z = torch.empty((q, n // pool, N), dtype=torch.float32, device=y.device)
        idx = torch.empty((q, n // pool, N), dtype=torch.uint8, device=y.device)
        _lib.check(_lib.lib().tgcn_relu_pool_f32(_lib.stream_ptr(), _lib.ptr(y), _lib.ptr(z), _lib.ptr(idx), q, n, N, pool))
        ctx.save_for_backward(z, idx)
        ctx.shape, ctx.pool = (q, n, N), pool
        return z

    @staticmethod
    @_on_device
    def backward(ctx, gz):
        z, idx = ctx.saved_tensors
        q, n, N = ctx.shape
        gy = torch.empty((q, n, N), dtype=torch.float32, device=gz.device)
        _lib.check(_lib.lib().tgcn_relu_pool_bwd_f32(_lib.stream_ptr(), _lib.ptr(gz.contiguous()), _lib.ptr(z), _lib.ptr(idx), _lib.ptr(gy),
                                                     q, n, N, ctx.pool))
        return gy, None


@_on_device
def pack_rows(src, idx, out):
    """out[i] = src[idx[i]] for a (rows, C) view with contiguous rows (halo messages of the vertex-sharded layer)"""
    _lib.require_device(src, idx, out)
    assert src.dim() == 2 and src.stride(1) == 1 and out.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous()
    _lib.check(_lib.lib().tgcn_pack_rows_f32(_lib.stream_ptr(), _lib.ptr(src), src.stride(0), _lib.ptr(idx), idx.numel(), src.shape[1],
                                             _lib.ptr(out)))
    return out


# ----------------------------------------------------------------------------------------- pooling
class PoolMaxFn(torch.autograd.Function):
    @staticmethod
    @_on_device
    def forward(ctx, x, p):
        _lib.require_device(x)
        x = x.float().contiguous()       # the kernel reads fp32; the reference's torch.max takes any dtype (gcn.py:246-255)
        q, n, f = x.shape
        out = torch.empty((q, n // p, f), dtype=torch.float32, device=x.device)
        idx = torch.empty((q, n // p, f), dtype=torch.int32, device=x.device)
        _lib.check(_lib.lib().tgcn_pool_max_f32(_lib.stream_ptr(), _lib.ptr(x), _lib.ptr(out), _lib.ptr(idx), q, n, f, p))
        ctx.save_for_backward(idx)
        ctx.shape, ctx.p = (q, n, f), p
        return out

    @staticmethod
    @_on_device
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        q, n, f = ctx.shape
        gi = torch.empty((q, n, f), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().tgcn_pool_max_bwd_f32(_lib.stream_ptr(), _lib.ptr(g.contiguous()), _lib.ptr(idx), _lib.ptr(gi),
                                                    q, n, f, ctx.p))
        return gi, None
