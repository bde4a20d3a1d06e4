"""Drop-in modules: the class surface of tgcn/nn/gcn.py (and tgcn/nn/gcn_matmul.py) on the HIP path.

Same constructor / forward signatures, parameter names, shapes and initialisation as the reference
(state_dict compatible); `L` stays a plain attribute (not saved, not moved by .to(), gcn.py:18,92,168).
The forward of every class is one call into libtgcn_hip.so (K-1 CSR hops + MFMA projection); nothing here
falls back to torch ops or the CPU.
"""
import collections
import math
import threading

import torch
from torch.nn import Parameter

from . import _lib, functional as F
from .graph import GraphOperand


def uniform(size, tensor):
    """U(-1/sqrt(size), 1/sqrt(size)) in place; None is ignored (reference: tgcn/nn/gcn.py:240-243)."""
    bound = 1.0 / math.sqrt(size)
    if tensor is not None:
        tensor.data.uniform_(-bound, bound)


def gcn_pool(x):
    """Max over pairs of consecutive vertices, (q,n,f)->(q,n/2,f) (reference: gcn.py:246-249)."""
    return F.PoolMaxFn.apply(x, 2)


def gcn_pool_4(x):
    """Max over 4 consecutive vertices (reference: gcn.py:252-255)."""
    return F.PoolMaxFn.apply(x, 4)


class _OperandCache:
    """One GraphOperand per (source object identity, version, device): DataParallel replicas on other devices
    and in-place edits of L / edge_index get their own.  An entry keeps its source objects alive: identity and
    data_ptr only name a tensor while it exists -- once freed, the allocator may hand the same address (and Python the
    same id) to a new edge_index of the same shape, which must not find the old operand."""

    MAX_ENTRIES = 16      # least recently used beyond this (a per-subject edge_index in a training loop: pygeo_hcp.py:284 swaps the graph per subject)

    def __init__(self):
        self._d = collections.OrderedDict()
        self._lock = threading.Lock()     # nn.DataParallel runs the replicas' forwards in threads that share this object

    # copies and pickles of a module start with an empty cache (entries hold device pointers in ctypes structs)
    def __deepcopy__(self, memo):
        return _OperandCache()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self._d = collections.OrderedDict()
        self._lock = threading.Lock()

    def get(self, key, build, sources=()):
        with self._lock:
            hit = self._d.get(key)
            if hit is None:
                hit = self._d[key] = (build(), tuple(sources))
                while len(self._d) > self.MAX_ENTRIES:        # evict the least recently used entry only: the others keep their operands
                    self._d.popitem(last=False)
            else:
                self._d.move_to_end(key)
            return hit[0]


def _np_fingerprint(L):
    """scipy / ndarray operands carry no version counter: tag the first and last stored values so that an in-place
    edit is (very likely) noticed."""
    from .numpy_api import _fingerprint
    return _fingerprint(L)


def _tensor_key(t):
    return None if t is None else (t.data_ptr(), t._version, tuple(t.shape), str(t.device))


class _DenseLBase(torch.nn.Module):
    """Shared plumbing of the three classes that take L in the constructor."""

    def _operand(self, device):
        L = self.L
        dense = isinstance(L, torch.Tensor) and L.layout == torch.strided
        key = (id(L), L.data_ptr() if dense else None, getattr(L, "_version", 0), tuple(L.shape) if hasattr(L, "shape") else None,
               str(device), None if isinstance(L, torch.Tensor) else _np_fingerprint(L))
        return self._ops.get(key, lambda: GraphOperand.from_any(L, device), sources=(L,))

    def _num_vertices(self):
        return self.L.shape[0] if hasattr(self.L, "shape") else self.L[0].shape[0]

    def reset_parameters(self):
        size = self.in_channels * self.weight.size(0)
        uniform(size, self.weight)
        uniform(size, self.bias)

    def __repr__(self):
        return '{}({}, {}, filter_order={})'.format(self.__class__.__name__, self.in_channels, self.out_channels,
                                                    self.weight.size(0))

    def _stack(self, X4):
        """X4: (q, n, h, f) or (q, n, f) -> stack (K, ...) in the reference_power recursion."""
        sh = X4.shape
        x3 = X4.reshape(sh[0], sh[1], -1).float().contiguous()
        st = F.cheb_stack(self._operand(x3.device), x3, self.filter_order, F.MODE_POWER)
        return st.reshape((self.filter_order,) + tuple(sh))


class TGCNCheb(_DenseLBase):
    """reference: tgcn/nn/gcn.py:8-79.  x (q, n, f) -> (q, n, g); weight (K, f, g); bias (1, n, g)."""

    def __init__(self, L, in_channels, out_channels, filter_order, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = Parameter(torch.Tensor(filter_order, in_channels, out_channels))
        self.L = L
        self.filter_order = filter_order
        self._ops = _OperandCache()
        if bias:
            self.bias = Parameter(torch.Tensor(1, self._num_vertices(), out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def _layer_args(self, x):
        x3 = x.float().contiguous()
        return (self._operand(x3.device), x3, self.weight, self.bias,
                F.BIAS_NONE if self.bias is None else F.BIAS_VERTEX_CHANNEL, F.MODE_POWER)

    def forward(self, x):
        return F.cheb_layer(*self._layer_args(x))

    def _time_chebyshev(self, X):
        return self._stack(X)


class TGCNCheb_H(_DenseLBase):
    """reference: tgcn/nn/gcn.py:82-154.  x (q, n, h[, f]) -> (q, n, g); weight (K, H, f, g); bias (1, n, g)."""

    def __init__(self, L, in_channels, out_channels, filter_order, horizon, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = Parameter(torch.Tensor(filter_order, horizon, in_channels, out_channels))
        self.L = L
        self.filter_order = filter_order
        self._ops = _OperandCache()
        if bias:
            self.bias = Parameter(torch.Tensor(1, self._num_vertices(), out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def _layer_args(self, x):
        if x.dim() == 3:
            x = x.unsqueeze(3)
        q, n, h, f = x.shape
        x3 = x.float().reshape(q, n, h * f).contiguous()
        W = self.weight.reshape(self.weight.shape[0], h * f, self.out_channels)
        return (self._operand(x3.device), x3, W, self.bias,
                F.BIAS_NONE if self.bias is None else F.BIAS_VERTEX_CHANNEL, F.MODE_POWER)

    def forward(self, x):
        return F.cheb_layer(*self._layer_args(x))

    def _time_chebyshev(self, X):
        if X.dim() == 3:
            X = X.unsqueeze(3)
        return self._stack(X)

    def forward_series(self, series):
        """Additive API (not in the reference): series (S, n, T) raw recordings -> the layer's output for all
        T-H+1 sliding windows of every recording, (S*(T-H+1), n, g), without materialising the windows
        (load/data_hcp.py:116-154 builds them on the host and the hops then run H times too often).
        in_channels must be 1.  Differentiable: the backward runs the hops once per recording too."""
        assert self.in_channels == 1, "forward_series: in_channels must be 1"
        K, H = self.weight.shape[0], self.weight.shape[1]
        W = self.weight.reshape(K, H, self.out_channels)
        return F.cheb_time_windows(self._operand(series.device), series, W,
                                   None if self.bias is None else self.bias.reshape(-1),
                                   F.BIAS_NONE if self.bias is None else F.BIAS_VERTEX_CHANNEL, F.MODE_POWER)


class GCNCheb(_DenseLBase):
    """reference: tgcn/nn/gcn.py:158-237.  x (q, n[, f]) -> (q, n, g); weight (K, f, g); bias (1, 1, g)."""

    def __init__(self, L, in_channels, out_channels, filter_order, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = Parameter(torch.Tensor(filter_order, in_channels, out_channels))
        self.L = L
        self.filter_order = filter_order
        self._ops = _OperandCache()
        if bias:
            self.bias = Parameter(torch.Tensor(1, 1, out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def _layer_args(self, x):
        if x.dim() == 2:
            x = x.unsqueeze(2)
        x3 = x.float().contiguous()
        return (self._operand(x3.device), x3, self.weight, self.bias,
                F.BIAS_NONE if self.bias is None else F.BIAS_CHANNEL, F.MODE_POWER)

    def forward(self, x):
        return F.cheb_layer(*self._layer_args(x))

    def _chebyshev(self, X):
        if X.dim() == 2:
            X = X.unsqueeze(2)
        return self._stack(X)


# ------------------------------------------------------------------------------------ COO helpers
_spmm_ops = _OperandCache()


def _stamp(t):
    return (t.data_ptr(), t._version)


def _refresh_values(op, weight, make_vals):
    """A LEARNABLE weight changes every optimizer step (and a computed one is a new tensor every forward) while the pattern stays: the operand is
    cached by pattern and only its packed values are refreshed in place (GraphOperand.update_values) -- a cache keyed on the weight's version would
    rebuild CSR, schedule and transpose per step and keep up to 16 stale operands alive.
    "Has the weight changed" = (data_ptr, _version) against the stamp of the last packing.  That pair only names a tensor while its storage is
    alive -- a freed weight's address can be handed to the next one, version 0 again -- so the operand holds the detached alias of the tensor it
    was packed from (shares storage and version counter, keeps no autograd graph): same address then means same storage, and its version counter
    says whether it was written."""
    stamp = _stamp(weight)
    if getattr(op, "_packed_stamp", None) != stamp:
        with op._lock:
            if getattr(op, "_packed_stamp", None) != stamp:
                op.update_values(make_vals())
                op._packed_alias = weight.detach()
                op._packed_stamp = stamp
    return op


def _coo_operand(index, value, m, device, n_cols=None):
    """m x n_cols operand: the reference's gather / scatter_add form takes any number of source rows (gcn.py:296-308)."""
    n_cols = int(m if n_cols is None else n_cols)
    if value.requires_grad:          # learnable values: one operand per PATTERN, values refreshed in place
        key = ("learnable", _tensor_key(index), int(m), n_cols, str(device))
        op = _spmm_ops.get(key, lambda: GraphOperand.from_coo(int(m), index[0], index[1], value.detach(), device, n_cols=n_cols), sources=(index,))
        return _refresh_values(op, value, lambda: value.detach().to(device=op.device, dtype=torch.float32).reshape(-1)[_coo_order(index, op)])
    key = (_tensor_key(index), _tensor_key(value), int(m), n_cols, str(device))
    return _spmm_ops.get(key, lambda: GraphOperand.from_coo(int(m), index[0], index[1], value.detach(), device, n_cols=n_cols),
                         sources=(index, value.detach()))      # the detached alias shares storage and version counter: the address cannot be reused, no graph is kept


def _coo_order(index, op):
    """entry order of GraphOperand.from_coo: sorted by (row, col), duplicates in their given order (tests/test_device_build.py pins both builders to it)"""
    key = ("order", _tensor_key(index), op.n, op.n_cols, str(op.device))
    return _spmm_ops.get(key, lambda: torch.argsort(index[0].to(op.device).long() * max(op.n, op.n_cols) + index[1].to(op.device).long(), stable=True),
                         sources=(index,))


def _coo_values(index, value, op):
    """`value` in the operand's CSR order as a tensor of the autograd graph (None when it needs no gradient): entries sorted by (row, col),
    duplicates in their given order -- the order GraphOperand.from_coo packs (tests/test_device_build.py pins the two builders to it)."""
    if value is None or not value.requires_grad:
        return None
    return value.to(device=op.device, dtype=torch.float32).reshape(-1)[_coo_order(index, op)]


def _spmm3(op, x3, values):
    """one hop, differentiable in the matrix and (when given) in the operand's values"""
    if values is None and not x3.requires_grad:
        return F.csr_hop(op, x3)
    return F.SpmmFn.apply(x3, values, op)


def spmm(index, value, m, matrix):
    """out[r] += v_e * matrix[c] over axis 0 (reference: gcn.py:258-278)."""
    matrix = matrix if matrix.dim() > 1 else matrix.unsqueeze(-1)
    op = _coo_operand(index, value, m, matrix.device, matrix.shape[0])
    x3 = matrix.float().reshape(1, matrix.shape[0], -1).contiguous()
    return _spmm3(op, x3, _coo_values(index, value, op)).reshape((m,) + tuple(matrix.shape[1:]))


def spmm_batch_2(index, value, m, matrix):
    """Same product over axis 1 of (q, n[, f]) (reference: gcn.py:281-310; a 2-D input gains a channel axis)."""
    if matrix.dim() == 2:
        matrix = matrix.unsqueeze(-1)
    return spmm_batch_3(index, value, m, matrix)


def spmm_batch_3(index, value, m, matrix):
    """Same product over axis 1 of (q, n, h, f) (reference: gcn.py:313-345)."""
    op = _coo_operand(index, value, m, matrix.device, matrix.shape[1])
    sh = matrix.shape
    x3 = matrix.float().reshape(sh[0], sh[1], -1).contiguous()
    return _spmm3(op, x3, _coo_values(index, value, op)).reshape((sh[0], m) + tuple(sh[2:]))


# ------------------------------------------------------------------------------------ edge-list classes
class _EdgeBase(torch.nn.Module):
    def _operand(self, x, edge_index, edge_weight):
        n = x.size(1)
        if edge_weight is not None:
            assert edge_weight.reshape(-1).size(0) == edge_index.size(1)
        w = None if edge_weight is None else edge_weight.detach()        # shares storage and version counter; keeps no autograd graph alive
        if edge_weight is not None and edge_weight.requires_grad:
            # learnable weights: ONE operand per edge_index; the values lap_e = coef_e * w_e are refreshed in place when the weight has changed
            key = ("learnable", _tensor_key(edge_index), n, str(x.device))
            op = self._ops.get(key, lambda: GraphOperand.from_edge_index(edge_index, w, n, x.device), sources=(edge_index,))

            def vals():
                src, coef = self._links(edge_index, n, x.device)
                return coef * w.to(device=x.device, dtype=torch.float32).reshape(-1)[src]
            # (the first build's values are re-packed by the same formula as every later refresh, so that equal weights give bit-equal
            # operands whatever the history of the module: the builder rounds -d^-1/2 w d^-1/2 in another order)
            return _refresh_values(op, edge_weight, vals)
        key = (_tensor_key(edge_index), _tensor_key(edge_weight), n, str(x.device))
        return self._ops.get(key, lambda: GraphOperand.from_edge_index(edge_index, w, n, x.device), sources=(edge_index, w))

    def _links(self, edge_index, n, dev):
        """(source edge of every stored entry, its constant coefficient -deg^-1/2[row] deg^-1/2[col]) in the operand's CSR order, once per edge_index"""
        def links():
            row, col = edge_index[0].to(dev).long(), edge_index[1].to(dev).long()
            ids = (row != col).nonzero().flatten()
            r, c = row[ids], col[ids]
            order = torch.argsort(r * n + c, stable=True)            # the order GraphOperand.from_coo packs: (row, col), duplicates as given
            dis = torch.bincount(r, minlength=n).to(torch.float32).pow(-0.5)
            dis[torch.isinf(dis)] = 0
            return ids[order], (-(dis[r] * dis[c]))[order]
        return self._ops.get(("links", _tensor_key(edge_index), n, str(dev)), links, sources=(edge_index,))

    def _values(self, x, edge_index, edge_weight, op):
        """The operand's values in CSR order as a function of a LEARNABLE edge_weight (None otherwise): lap_e = -deg^-1/2[row] w_e deg^-1/2[col]
        with the unweighted source degree (gcn.py:408-413), i.e. a constant coefficient per kept edge times its weight -- differentiable index
        plumbing; the arithmetic of the gradient itself is F.chebyshev_values_grad."""
        if edge_weight is None or not edge_weight.requires_grad:
            return None
        n, dev = x.size(1), x.device
        src, coef = self._links(edge_index, n, dev)
        assert src.numel() == op.nnz
        return coef * edge_weight.to(device=dev, dtype=torch.float32).reshape(-1)[src]

    def reset_parameters(self):
        size = self.in_channels * self.weight.size(0)
        uniform(size, self.weight)
        uniform(size, self.bias)

    def __repr__(self):
        return '{}({}, {}, K={})'.format(self.__class__.__name__, self.in_channels, self.out_channels, self.weight.size(0))


class ChebConv(_EdgeBase):
    """reference: tgcn/nn/gcn.py:348-442.  forward(x (q,n[,f]), edge_index (2,E), edge_weight=None) -> (q,n,g)."""

    def __init__(self, in_channels, out_channels, K, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = Parameter(torch.Tensor(K, in_channels, out_channels))
        self._ops = _OperandCache()
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def _layer_args(self, x, edge_index, edge_weight=None):
        op = self._operand(x, edge_index, edge_weight)
        if x.dim() < 3:
            x = x.unsqueeze(-1)
        return (op, x.float().contiguous(), self.weight, self.bias,
                F.BIAS_NONE if self.bias is None else F.BIAS_CHANNEL, F.MODE_CHEBYSHEV)

    def forward(self, x, edge_index, edge_weight=None):
        args = self._layer_args(x, edge_index, edge_weight)
        return F.cheb_layer(*args, values=self._values(x, edge_index, edge_weight, args[0]))


class ChebTimeConv(_EdgeBase):
    """reference: tgcn/nn/gcn.py:445-538.  forward(x (q,n,h[,f]), edge_index, edge_weight=None) -> (q,n,g)."""

    def __init__(self, in_channels, out_channels, K, H, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = Parameter(torch.Tensor(K, H, in_channels, out_channels))
        self._ops = _OperandCache()
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def _layer_args(self, x, edge_index, edge_weight=None):
        op = self._operand(x, edge_index, edge_weight)
        if x.dim() < 4:
            x = x.unsqueeze(-1)
        q, n, h, f = x.shape
        W = self.weight.reshape(self.weight.shape[0], h * f, self.out_channels)
        return (op, x.float().reshape(q, n, h * f).contiguous(), W, self.bias,
                F.BIAS_NONE if self.bias is None else F.BIAS_CHANNEL, F.MODE_CHEBYSHEV)

    def forward(self, x, edge_index, edge_weight=None):
        args = self._layer_args(x, edge_index, edge_weight)
        return F.cheb_layer(*args, values=self._values(x, edge_index, edge_weight, args[0]))


# ------------------------------------------------------------------------------------ fused caller pattern
def cheb_relu_pool(layer, x, *graph_args, pool=4):
    """gcn_pool_4(F.relu(layer(x, ...)))  (pool=4)  /  gcn_pool(F.relu(layer(x, ...)))  (pool=2)  in one fused op:
    additive API for the pattern of examples/pytorch_based/pytorch_hcp_tgcn.py:134-141 and pytorch_mnist_gcn.py.
    `layer` is any of the five modules above; graph_args are ChebConv's (edge_index[, edge_weight])."""
    if len(graph_args) > 1 and graph_args[1] is not None and graph_args[1].requires_grad:
        # learnable edge weights: the gradient w.r.t. them belongs to the layer function; relu + pool as their own pass behind it
        return F.ReluPoolFn.apply(layer(x, *graph_args), pool)
    return F.cheb_relu_pool(*layer._layer_args(x, *graph_args), pool)
