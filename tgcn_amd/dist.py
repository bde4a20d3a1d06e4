"""Multi-GPU sharding of the Chebyshev layer: one process per GPU, torch.distributed ("nccl" is RCCL on ROCm).

The reference's only multi-GPU mechanism is nn.DataParallel's batch split (examples/pytorch_based/
pytorch_hcp_tgcn.py:270-273).  The path shards two ways:

  * by sample / time step: every column of X is filtered independently and the projection contracts inside a
    sample, so ranks that hold the CSR need NO data-path communication (`shard_time_steps`; bench.py --gpus N);
  * by vertices: 1-D row partition of L-hat, X and out; one exchange of the previous hop's cut-edge neighbour
    rows per hop (`VertexShardedCheb`): point-to-point halo rows when the cut is small, an RCCL all-gather of the
    owned row blocks when the halo is most of the graph (R-MAT);
  * both at once (`hybrid_groups`): vertex shards inside a group of ranks, time steps across groups.

Round 6: the vertex-sharded layer has the single-GPU path's drivers and the reference's module surface.

  * project-first inside the shard (2 C_out <= C_in H, e.g. TGCNCheb_H(L, 1, 32, 5, 1200), pytorch_hcp_tgcn.py:103-104): ONE local
    projection Z = x [W'_0 | ... | W'_{K-1}], then Horner / Clenshaw on C_out-wide rows -- the hops AND the halo messages move C_out
    instead of C_in H floats per row (cfg4: 32 instead of 1200);
  * the weight arrives in the reference's basis and is folded inside (`layer`), the projection writes / reads the shard's
    [interior | boundary] row order through its row map (no separate permutation pass of x, bias or out);
  * backward (`layer_backward`): dX through the TRANSPOSED shard (`transpose()`: the owned entries go to the owners of their columns by
    all_to_all, the reverse halo lists follow from the same constructor), dW and a per-channel dbias all-reduced over the group; the
    adjoint terms T_k(L^T) g are computed on whichever side has the narrower rows and serve both gradients;
  * `ShardedTGCNCheb / ShardedTGCNCheb_H / ShardedGCNCheb`: the reference's constructor arguments + a process group; `weight` / `bias`
    keep their names and GLOBAL shapes (a reference state_dict loads), forward(x_local) -> out_local on the owned rows;
  * the constructor talks in tensors only (count exchange + all_to_all_single of id tensors; no pickled Python lists), and a rank may
    pass just its own rows when the row bounds are given.

Vertex sharding overlaps communication with compute: a shard's rows are kept in the order
[interior | boundary]; the interior rows of hop k (no remote column) run while the halo rows of hop k-1 are in flight,
messages are packed by a HIP kernel and received in place, time steps are pipelined in groups of `depth` (the exchange
of one under the hops of the others), and every buffer lives as long as the object.

The arithmetic sits behind one small interface (`HipOps`: the entry points of libtgcn_hip.so, nothing else) so that the communication
logic is testable with gloo on the CPU against the oracle with a stand-in injected (tools/cpu_standins.py, tests/test_dist_gloo.py,
tests/test_sharded_modules.py: overlapped against non-overlapped form bit for bit, gradients against oracle.layer_backward).
"""
import torch
import torch.distributed as dist

from . import nn as _nn

PROJECT_FIRST = True      # developer switch: project first inside a shard when the output is at most half as wide as an input row


def shard_time_steps(q, rank, world):
    """Contiguous, balanced slice of the q samples / time steps owned by `rank`."""
    base, extra = divmod(q, world)
    lo = rank * base + min(rank, extra)
    return slice(lo, lo + base + (1 if rank < extra else 0))


def hybrid_groups(world, vertex_shards, rank=None):
    """2-D layout of SURVEY.md section 8e: `vertex_shards` consecutive ranks share one copy of the graph (vertex
    sharding, one exchange per hop inside the group) and the world // vertex_shards groups split the samples / time
    steps between them with no communication.  Every rank must call this (new_group is collective).
    -> (group of this rank, index of its group, number of groups)."""
    if world % vertex_shards:
        raise ValueError("world size %d is not a multiple of %d vertex shards" % (world, vertex_shards))
    rank = dist.get_rank() if rank is None else rank
    ngroups = world // vertex_shards
    mine = None
    for g in range(ngroups):
        grp = dist.new_group(ranks=list(range(g * vertex_shards, (g + 1) * vertex_shards)))
        if rank // vertex_shards == g:
            mine = grp
    return mine, rank // vertex_shards, ngroups


def balanced_row_bounds(row, n, world, multiple=1):
    """world+1 row boundaries with ~equal stored entries per shard (power-law rows => not equal row counts).  `row`: the row ids of the
    WHOLE entry list (any order).  multiple > 1: inner boundaries rounded to a multiple of it, so that groups of `multiple` consecutive
    vertices (gcn_pool / gcn_pool_4 between two sharded layers, tgcn/nn/gcn.py:246-255) never straddle two ranks."""
    counts = torch.bincount(row, minlength=n)
    cum = torch.cumsum(counts, 0)
    total = int(cum[-1].item()) if n else 0
    marks = (torch.arange(1, world, device=row.device, dtype=torch.int64) * total) // world
    inner = (torch.searchsorted(cum, marks) + 1).clamp_(max=n)
    if multiple > 1:
        inner = ((inner + multiple // 2) // multiple * multiple).clamp_(max=n // multiple * multiple)
    b = torch.cat([torch.zeros(1, dtype=torch.int64, device=row.device), inner,
                   torch.full((1,), n, dtype=torch.int64, device=row.device)])
    return torch.cummax(b, 0)[0]


class HipOps:
    """The arithmetic of the sharded layer: the entry points of libtgcn_hip.so through tgcn_amd.functional (no CPU path: a CPU tensor
    raises there).  tools/cpu_standins.py has the numpy object the gloo tests inject instead."""
    name = "hip"

    def operand(self, n_rows, n_cols, row, col, val, device):
        from .graph import GraphOperand
        return GraphOperand.from_coo(n_rows, row, col, val, device, n_cols=n_cols)

    def edge_coo(self, edge_index, edge_weight, n, device):
        """(row, col, lap) of the edge-list classes' operand (tgcn/nn/gcn.py:398-413, :495-510: self loops removed, unweighted source degree,
        lap_e = -deg^-1/2[row] w_e deg^-1/2[col]) through the library's edge normalisation kernels"""
        from .graph import GraphOperand
        w = None if edge_weight is None else edge_weight.detach()
        return GraphOperand.from_edge_index(edge_index, w, n, device).coo()

    def hop(self, op, x, z, alpha, beta, out, z2=None, gamma=0.0):
        from . import functional as F
        return F.csr_hop(op, x, z=z, alpha=alpha, beta=beta, out=out, z2=z2, gamma=gamma)

    def project(self, terms, W, bias, bias_kind, n_vertices, rowmap=None, out=None):
        """out[r(m)] = sum_t terms[t][m] W[t] + bias[r(m)]; r = rowmap (int32: the shard's row order -> the caller's) or the identity"""
        from . import functional as F
        if rowmap is None:
            return F.cheb_project(terms, W, bias, bias_kind, n_vertices, out=out)
        T = len(terms)
        M, Kc = terms[0].shape
        N = W.shape[-1]
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=W.device)
        b = bias.contiguous() if bias is not None else None
        F.project_mapped(terms, [0] * T, W.reshape(T * Kc, N).contiguous(), b, bias_kind if b is not None else 0, n_vertices, rowmap, 0, 1,
                         out.view(1, n_vertices, N))
        return out

    def project_first(self, x3, Wcat, bias, bias_kind, K, N, rowmap=None):
        from . import functional as F
        return F.project_first(x3, Wcat, bias, bias_kind, K, N, rowmap=rowmap)

    def pack(self, src, idx, out):
        """out[i, :] = src[idx[i], :] -- the rows a peer asked for, packed for one message (tgcn_pack_rows_f32)"""
        from . import functional as F
        return F.pack_rows(src, idx, out)

    def wgrad(self, terms, g2d):
        from . import functional as F
        return F.cheb_wgrad(terms, g2d)

    def fold(self, W, transpose=False):
        """reference basis -> monomial basis of the dense-L classes' recursion (tgcn/nn/gcn.py:75-78); transpose: the adjoint, for dW"""
        from . import functional as F
        K = W.shape[0]
        return W if K <= 2 else F.fold_weight(F.power_fold_matrix(K, W.device), W, transpose=transpose)

    def weight_layout(self, W, kind):
        from . import functional as F
        return F.weight_layout(W, kind)


_HIP_OPS = HipOps()


class CooGraph:
    """A global entry list handed to the sharded modules in place of a dense / sparse L: (n, row, col, val) with the `.shape` the
    reference's constructors read the vertex count from (tgcn/nn/gcn.py:22,96: L[0].shape[0])."""

    def __init__(self, n, row, col, val):
        self.n, self.row, self.col, self.val = int(n), row, col, val
        self.shape = (self.n, self.n)


def _coo_of(L, device):
    """(n, row, col, val) of whatever the reference's constructors take as L (index plumbing; duplicates are kept as separate entries)"""
    from .graph import GraphOperand
    if isinstance(L, CooGraph):
        n, row, col, val = L.n, L.row, L.col, L.val
    elif isinstance(L, GraphOperand):
        n = L.n
        row, col, val = L.coo()
    elif isinstance(L, torch.Tensor):
        n = L.shape[0]
        if L.layout != torch.strided:
            Lc = (L.to_sparse_coo() if L.layout == torch.sparse_csr else L).coalesce()
            row, col, val = Lc.indices()[0], Lc.indices()[1], Lc.values()
        else:
            idx = L.nonzero()
            row, col = idx[:, 0], idx[:, 1]
            val = L[row, col]
    elif hasattr(L, "tocoo"):
        coo = L.tocoo()
        n = coo.shape[0]
        row, col, val = torch.as_tensor(coo.row), torch.as_tensor(coo.col), torch.as_tensor(coo.data)
    else:
        return _coo_of(torch.as_tensor(L), device)
    return n, row.to(device=device, dtype=torch.int64), col.to(device=device, dtype=torch.int64), val.to(device=device, dtype=torch.float32)


class VertexShardedCheb:
    """Vertex-sharded layer.  Rank r owns rows [bounds[r], bounds[r+1]) of L-hat, x, bias and out.

    Entries: every rank passes the same global COO of L-hat (bounds=None: nnz-balanced bounds are computed from it), or -- with `bounds`
    given -- at least the entries of its own rows.
    exchange = "halo": ext operand = [owned rows ; halo rows], halo rows arrive by point-to-point messages from
                       their owners (index lists agreed once at construction);
               "allgather": ext operand = all shards' row blocks padded to the largest, one all_gather per hop;
               "auto": allgather when the halo is more than half of the remote vertices.
    """

    def __init__(self, n, row, col, val, group=None, device=None, exchange="auto", bounds=None, ops=None, row_multiple=1):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # point-to-point peers are named by GLOBAL rank; inside a sub-group (hybrid layout) translate the group index
        self.peer = [dist.get_global_rank(group, p) if group is not None else p for p in range(self.world)]
        self.device = torch.device(device) if device is not None else row.device
        # set-up collectives (counts, id lists, entries of the transpose) travel on the backend's own device
        self.comm_device = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.ops = _HIP_OPS if ops is None else ops
        self._bufs = {}
        self._T = None
        self.collect_stats = False      # True: forward() times its phases on this rank (device events / wall clock) into self.stats
        self.stats = None
        self.last_width = None          # floats per row of the last forward's hops and messages
        row = row.to(device=self.device, dtype=torch.int64)
        col = col.to(device=self.device, dtype=torch.int64)
        val = val.to(device=self.device, dtype=torch.float32)
        self.n = int(n)
        if bounds is None:
            self.bounds = balanced_row_bounds(row, self.n, self.world, row_multiple)
        else:
            self.bounds = torch.as_tensor(bounds, dtype=torch.int64).to(self.device)
            assert self.bounds.numel() == self.world + 1 and int(self.bounds[0]) == 0 and int(self.bounds[-1]) == self.n
        b = self.bounds.tolist()
        if any(b[i + 1] <= b[i] for i in range(self.world)):
            # every rank sees the same bounds, so every rank raises: no rank is left waiting in a collective
            raise ValueError("vertex sharding over %d ranks leaves a rank without rows (bounds %s): fewer shards, or pass bounds=" % (self.world, b))
        self.lo, self.hi = b[self.rank], b[self.rank + 1]
        mine = (row >= self.lo) & (row < self.hi)
        self._exchange_arg = exchange
        self._setup(row[mine], col[mine], val[mine], exchange)

    @classmethod
    def _from_owned(cls, n, bounds, row, col, val, group, device, exchange, ops):
        return cls(n, row, col, val, group=group, device=device, exchange=exchange, bounds=bounds, ops=ops)

    # ------------------------------------------------------------------ construction
    def _a2a(self, send, send_counts, dtype):
        """all_to_all_single of a 1-D tensor split by `send_counts` (list, one per rank of the group) -> (received tensor, receive counts)"""
        cd = self.comm_device
        cnt_out = torch.tensor(send_counts, dtype=torch.int64, device=cd)
        cnt_in = torch.empty(self.world, dtype=torch.int64, device=cd)
        dist.all_to_all_single(cnt_in, cnt_out, group=self.group)
        recv_counts = [int(c) for c in cnt_in.tolist()]
        recv = torch.empty(sum(recv_counts), dtype=dtype, device=cd)
        dist.all_to_all_single(recv, send.to(device=cd, dtype=dtype).contiguous(), output_split_sizes=recv_counts, input_split_sizes=list(send_counts),
                               group=self.group)
        return recv.to(self.device), recv_counts

    def _setup(self, r_g, c_g, v, exchange):
        """r_g, c_g, v: the stored entries of the owned rows, global ids"""
        group, ops = self.group, self.ops
        b = self.bounds.tolist()
        self._entries = (r_g, c_g, v)             # kept for transpose(): the entries whose COLUMN another rank owns travel there
        self.owned = self.hi - self.lo
        self.n_max = max(b[i + 1] - b[i] for i in range(self.world))
        r, c = r_g - self.lo, c_g
        remote = (c < self.lo) | (c >= self.hi)
        halo_ids = torch.unique(c[remote])                       # sorted global ids
        self.halo = int(halo_ids.numel())
        n_remote = self.n - self.owned
        if exchange == "auto":
            frac = torch.tensor([self.halo / max(n_remote, 1)], dtype=torch.float64, device=self.comm_device)
            dist.all_reduce(frac, op=dist.ReduceOp.MAX, group=group)
            exchange = "allgather" if frac.item() > 0.5 else "halo"
        self.exchange = exchange
        if exchange == "halo":
            c_local = torch.where(remote, self.owned + torch.searchsorted(halo_ids, c), c - self.lo)
            self.n_ext = self.owned + self.halo
            owner = torch.searchsorted(self.bounds[1:].contiguous(), halo_ids, right=True)
            self.recv_counts = torch.bincount(owner, minlength=self.world).tolist()
            # what I need from each peer, as row ids local to the peer: halo_ids is sorted, so it is already grouped by owner.  One count
            # exchange + one all_to_all_single of the id tensor (no pickled Python lists: millions of ids per rank at cfg5 scale)
            want = halo_ids - self.bounds[owner]
            ids_in, send_counts = self._a2a(want, self.recv_counts, torch.int64)
            self.send_idx = [t.contiguous() for t in torch.split(ids_in, send_counts)]
        else:
            # position of global vertex g in the gathered operand: owner(g) * n_max + (g - bounds[owner])
            owner = torch.searchsorted(self.bounds[1:].contiguous(), c, right=True)
            c_local = owner * self.n_max + (c - self.bounds[owner])
            self.n_ext = self.world * self.n_max
        # ---- overlapped form: owned rows in the order [interior | boundary] -- interior rows have no remote
        # column, so their part of a hop runs while the exchange of the previous hop's cut rows is in flight
        if exchange == "halo":
            has_remote = torch.zeros(self.owned, dtype=torch.bool, device=self.device)
            has_remote[r[remote]] = True
            order = torch.argsort(has_remote.to(torch.int8), stable=True)          # interior rows first, original order inside each class
            self.n_int = int((~has_remote).sum().item())
        else:
            order = torch.arange(self.owned, device=self.device)
            self.n_int = 0                                                           # all-gather: the halo is (almost) everything
        self.local_perm = order                      # shard row i holds the caller's owned row local_perm[i]
        inv = torch.empty_like(order)
        inv[order] = torch.arange(self.owned, device=self.device)
        self.local_inv = inv
        # (a shard whose rows are all interior or all boundary -- world size 1, a rank inside one connected block -- keeps the caller's order)
        self._permuted = exchange == "halo" and 0 < self.n_int < self.owned
        self.perm32 = order.to(torch.int32) if self._permuted else None           # row maps of the projections (None: identity)
        self.inv32 = inv.to(torch.int32) if self._permuted else None
        r2 = inv[r]                                                                  # row in the [interior | boundary] order
        if exchange == "halo":
            c2 = torch.where(c_local < self.owned, inv[c_local.clamp(max=max(self.owned - 1, 0))], c_local)    # owned columns move with their rows
            self.send_idx_l = [inv[idx] if idx.numel() else idx for idx in self.send_idx]
            is_int = r2 < self.n_int
            self.op_int = ops.operand(self.n_int, self.n_ext, r2[is_int], c2[is_int], v[is_int], self.device) if self.n_int else None
            nb = self.owned - self.n_int
            self.op_bnd = ops.operand(nb, self.n_ext, r2[~is_int] - self.n_int, c2[~is_int], v[~is_int], self.device) if nb else None
            # the same rows as one operand (non-overlapped form): identical labels, so both forms sum every row in the same order
            self.op = self.op_all = ops.operand(self.owned, self.n_ext, r2, c2, v, self.device)
        else:                       # all-gather: every row waits for the gathered operand, one operand serves both forms
            self.op = ops.operand(self.owned, self.n_ext, r, c_local, v, self.device)
            self.op_int, self.op_bnd, self.op_all = None, self.op, self.op

    def transpose(self):
        """The same partition of L-hat^T (COLLECTIVE: every rank of the group calls it; built once).  Entry (i, j, v) of an owned row i is
        entry (j, i, v) of row j of the transpose, which the owner of j holds: the entries travel there by all_to_all (three tensors: row,
        column, value), and the reverse halo lists follow from the ordinary constructor on what arrives.  The input gradient of the layer is
        the layer on this object (dX = sum_k T_k(L^T) g W_k^T)."""
        if self._T is None:
            r_g, c_g, v = self._entries
            owner = torch.searchsorted(self.bounds[1:].contiguous(), c_g, right=True)
            order = torch.argsort(owner, stable=True)                     # grouped by destination, given order inside (duplicates keep theirs)
            counts = torch.bincount(owner, minlength=self.world).tolist()
            rT, _ = self._a2a(c_g[order], counts, torch.int64)
            cT, _ = self._a2a(r_g[order], counts, torch.int64)
            vT, _ = self._a2a(v[order], counts, torch.float32)
            T = VertexShardedCheb._from_owned(self.n, self.bounds, rT, cT, vT, self.group, self.device, self._exchange_arg, self.ops)
            T._T = self
            self._T = T
            self._entries = T._entries = None          # both directions exist now: the global-id copies of the owned entries (20 bytes each) can go
        return self._T

    # ------------------------------------------------------------------ buffers
    def _buf(self, name, shape):
        """buffers of the exchange path live as long as the object (no allocation per hop)"""
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            # zeroed once: padding rows of the all-gather blocks ([owned:n_max]) are never written, and they go on the wire
            t = self._bufs[name] = torch.zeros(shape, dtype=torch.float32, device=self.device)
        return t

    def _rows(self, src, idx, out=None):
        """src[idx] for a (rows, C) tensor through the row-packing kernel (tgcn_pack_rows_f32; the injected stand-in in CPU rehearsals)"""
        if out is None:
            out = torch.empty((idx.numel(), src.shape[1]), dtype=torch.float32, device=src.device)
        if idx.numel():
            self.ops.pack(src, idx, out)
        return out

    def _to_shard_order(self, src, out):
        """owned rows in the caller's order -> the shard's [interior | boundary] order"""
        return self._rows(src, self.local_perm, out) if self._permuted else out.copy_(src)

    def _to_caller_order(self, src, out):
        return self._rows(src, self.local_inv, out) if self._permuted else out.copy_(src)

    # ------------------------------------------------------------------ diagnostics
    def describe(self, width=None):
        """what this rank exchanges per hop and time step: enough to tell a slow link from a wrong partition in one record.  width: floats
        per exchanged row (default: the last forward's -- C_out on the project-first path, C_in H otherwise)"""
        width = self.last_width if width is None else width
        d = dict(rank=self.rank, exchange=self.exchange, owned_rows=self.owned, interior_rows=self.n_int, halo_rows=self.halo,
                 ext_rows=self.n_ext, nnz=self.op.nnz if hasattr(self.op, "nnz") else None, row_floats=width)
        row_bytes = None if width is None else 4 * int(width)
        if self.exchange == "halo":
            d["recv_rows_per_peer"] = list(self.recv_counts)
            d["send_rows_per_peer"] = [int(i.numel()) for i in self.send_idx]
            rows_in = sum(self.recv_counts)
            if row_bytes is not None:
                d["message_bytes_per_peer_in"] = [c * row_bytes for c in self.recv_counts]         # rows x row_floats x 4
        else:
            d["allgather_block_rows"] = self.n_max
            rows_in = self.n_max * (self.world - 1)
        d["rows_in_per_hop"] = rows_in
        d["bytes_per_channel_in"] = 4 * rows_in
        d["bytes_in_per_hop_and_time_step"] = None if row_bytes is None else rows_in * row_bytes
        return d

    def _mark(self, marks, name):
        """one time stamp of the phase log: a device event on the current stream (GPU) or the wall clock (CPU rehearsals)"""
        if marks is None:
            return
        if self.device.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))
        else:
            import time
            marks.append((name, time.perf_counter()))

    def _close_stats(self, marks):
        if marks is None:
            return
        tot = {}
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
            ms = a.elapsed_time(b) if self.device.type == "cuda" else (b - a) * 1e3
            tot[name] = tot.get(name, 0.0) + ms
        self.stats = {k: round(v, 3) for k, v in tot.items()}

    # ------------------------------------------------------------------ the exchange of one hop
    # ------------------------------------------------------------------ host transports (gloo): staged through pinned host memory
    # RCCL collectives are ordered on the device's streams: the exchange is enqueued behind the kernels that produce its data, wait() is a
    # stream dependency, receives land IN PLACE.  A host transport must never be handed a device pointer: torch's gloo send / recv take
    # `tensor.data_ptr()` as HOST memory -- on this platform the CPU can indeed address device memory, so nothing fails, but the bytes move
    # with no ordering against the stream (a pack kernel still queued, a hop still reading the buffer) and behind the GPU's L2 (a boundary
    # hop launched back to back with the interior one keeps lines of the receive region it had cached).  Seen as: cfg4 on two gloo ranks off
    # by 2.5 % and different run to run (fixed first by draining the stream before posting), then the full-size R-MAT failing one run in
    # five.  So on anything but nccl the messages are staged: device -> pinned host (stream-ordered copy, then a host wait), gloo on the HOST
    # buffers, pinned host -> device on the compute stream when the work is waited for.  Rehearsal transports only; RCCL never comes here.
    def _host_staged(self):
        return self.device.type == "cuda" and self.comm_device.type != "cuda"

    def _hbuf(self, name, shape):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._bufs[name] = torch.zeros(shape, dtype=torch.float32).pin_memory()
        return t

    class _StagedWork:
        """a posted host-side operation + the copies that bring what it received back to the device when it is waited for"""

        def __init__(self, works, copies):
            self.works, self.copies = works, copies

        def wait(self):
            for w in self.works:
                w.wait()
            for dst, src in self.copies:
                dst.copy_(src, non_blocking=True)         # pinned host -> device, ordered on the compute stream like the kernels that read it
            return True

    def _post(self, slot, k, src_ext, mine=None, async_op=True):
        """Start fetching the remote rows of a hop tensor; returns the work handles.
        halo: `src_ext` (1, n_ext, w) with its owned part valid -- messages are packed by the pack kernel into per-peer send buffers and
        received IN PLACE into its own halo region.  all-gather: `mine` (n_max, w) block of this rank -> `src_ext` (1, world * n_max, w)."""
        w = src_ext.shape[2]
        staged = self._host_staged()
        if self.exchange != "halo":
            if staged:
                mine_h = self._hbuf(("mine_h", w, slot), (self.n_max, w))
                ext_h = self._hbuf(("gath_h", w, slot), (self.world * self.n_max, w))
                mine_h.copy_(mine)                                                   # (blocking: the producing kernels have finished)
                h = dist.all_gather_into_tensor(ext_h, mine_h, group=self.group, async_op=async_op)
                work = self._StagedWork([h] if h is not None else [], [(src_ext.view(self.world * self.n_max, w), ext_h)])
                if not async_op:
                    work.wait()
                    return []
                return [work]
            h = dist.all_gather_into_tensor(src_ext.view(self.world * self.n_max, w), mine, group=self.group, async_op=async_op)
            return [h] if h is not None else []
        ops, copies = [], []
        off = self.owned
        for p in range(self.world):
            cnt = self.recv_counts[p]
            if p != self.rank and cnt:
                dst = src_ext[0, off: off + cnt]
                if staged:
                    rb = self._hbuf(("recv_h", w, slot, k & 1, p), (cnt, w))
                    copies.append((dst, rb))
                    dst = rb
                ops.append(dist.P2POp(dist.irecv, dst, self.peer[p], group=self.group))
            off += cnt
            idx = self.send_idx_l[p]
            if p != self.rank and idx.numel():
                sb = self._buf(("send", w, slot, k & 1, p), (idx.numel(), w))
                self.ops.pack(src_ext[0], idx, sb)
                if staged:
                    sh_ = self._hbuf(("send_h", w, slot, k & 1, p), (idx.numel(), w))
                    sh_.copy_(sb)                                                    # stream-ordered behind the pack kernel, host waits for it
                    sb = sh_
                ops.append(dist.P2POp(dist.isend, sb, self.peer[p], group=self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        return [self._StagedWork(works, copies)] if staged and works else works

    def _all_reduce(self, t, group):
        """sum over the group; staged through the host for host transports"""
        if self._host_staged():
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t

    def _hop(self, op, src, z, alpha, beta, z2, gamma, out):
        if z is None and z2 is not None:
            z, beta, z2, gamma = z2, gamma, None, 0.0
        if z is None:
            beta = 0.0
        self.ops.hop(op, src, z, alpha, beta, out, z2=z2, gamma=gamma if z2 is not None else 0.0)

    def _run_chain(self, q, width, prog, init, finish, overlap, depth, marks):
        """The sharded recursion: per time step a chain of len(prog) hops over len(prog) + 1 buffers of `width` floats per row (shard row
        order), each hop behind the exchange of its source's cut rows.
          init(s, dst)       fills the owned rows (owned, width) of chain buffer 0 of time step s;
          prog[i-1]          hop i:  buf[i] = alpha * L buf[i-1] + beta * z + gamma * z2 with (alpha, zsel, beta, z2sel, gamma); a selector is
                             None, ("buf", j) -- an earlier chain buffer -- or ("view", f) with f(s, lo, hi) -> a (1, hi - lo, width) view;
          finish(s, bufs)    consumes the owned rows of the chain buffers of time step s.
        overlap: the exchange of hop i runs under the interior rows of the same hop and under the hops of the other time steps of a group of
        `depth`; otherwise one blocking exchange, then the hop on all owned rows -- the same arithmetic row by row (bit for bit)."""
        owned, ni = self.owned, self.n_int
        allg = self.exchange != "halo"
        nh = len(prog)
        self.last_width = width
        group_size = max(1, depth) if overlap else 1
        for t0 in range(0, q, group_size):
            steps = range(t0, min(q, t0 + group_size))
            if allg:
                mine = {s: [self._buf(("mine", width, s - t0, k), (self.n_max, width)) for k in range(nh + 1)] for s in steps}
                gath = {s: self._buf(("gath", width, s - t0), (1, self.n_ext, width)) for s in steps}
                own = lambda s, k: mine[s][k][:owned].unsqueeze(0)                     # noqa: E731
                src = lambda s, k: gath[s]                                             # noqa: E731
                post = lambda s, i, a: self._post(s - t0, i, gath[s], mine[s][i - 1], async_op=a)       # noqa: E731
            else:
                exts = {s: [self._buf(("ext", width, s - t0, k), (1, self.n_ext, width)) for k in range(nh + 1)] for s in steps}
                own = lambda s, k: exts[s][k][:, :owned]                               # noqa: E731
                src = lambda s, k: exts[s][k]                                          # noqa: E731
                post = lambda s, i, a: self._post(s - t0, i, exts[s][i - 1])           # noqa: E731

            def pick(sel, s, lo, hi):
                if sel is None:
                    return None
                return own(s, sel[1])[:, lo:hi] if sel[0] == "buf" else sel[1](s, lo, hi)
            for s in steps:
                init(s, own(s, 0)[0])
            for i in range(1, nh + 1):
                alpha, zsel, beta, z2sel, gamma = prog[i - 1]
                if overlap:
                    works = {s: post(s, i, True) for s in steps}                       # 1. start every exchange of this hop level
                    self._mark(marks, "pack_and_post_ms")
                    if not allg and ni:
                        for s in steps:                                                # 2. interior rows: no remote column, no wait
                            self._hop(self.op_int, src(s, i - 1), pick(zsel, s, 0, ni), alpha, beta, pick(z2sel, s, 0, ni), gamma, own(s, i)[:, :ni])
                        self._mark(marks, "interior_hops_ms")
                    for s in steps:                                                    # 3. boundary rows as their halo arrives
                        for w in works[s]:
                            w.wait()
                        self._mark(marks, "exchange_wait_ms")
                        opb = self.op if allg else self.op_bnd
                        if opb is not None:
                            self._hop(opb, src(s, i - 1), pick(zsel, s, ni, owned), alpha, beta, pick(z2sel, s, ni, owned), gamma, own(s, i)[:, ni:])
                        self._mark(marks, "boundary_hops_ms")
                else:
                    for s in steps:
                        for w in post(s, i, False):
                            w.wait()
                        self._mark(marks, "exchange_ms")
                        self._hop(self.op_all, src(s, i - 1), pick(zsel, s, 0, owned), alpha, beta, pick(z2sel, s, 0, owned), gamma, own(s, i))
                        self._mark(marks, "hops_ms")
            for s in steps:
                finish(s, [own(s, k)[0] for k in range(nh + 1)])
            self._mark(marks, "projection_ms")

    @staticmethod
    def _basis_prog(K, mode):
        """hops-first: T_k = L T_{k-1} (mode 0: monomials, the basis of the folded weight) or T_1 = L T_0, T_k = 2 L T_{k-1} - T_{k-2} (mode 1)"""
        return [(1.0, None, 0.0, None, 0.0) if (mode == 0 or i == 1) else (2.0, ("buf", i - 2), -1.0, None, 0.0) for i in range(1, K)]

    def use_project_first(self, C, N, K):
        return PROJECT_FIRST and K > 1 and 2 * N <= C

    # ------------------------------------------------------------------ layer (working basis)
    def forward(self, x_local, W, bias_local, bias_kind, mode, overlap=True, depth=2, project_first=None):
        """x_local: (q, owned, C) rows of this shard in the caller's order; W: (K, C, N) in the WORKING basis (monomial-folded for mode 0:
        `layer` takes the reference's); bias_local: per channel [N] or this shard's rows [owned, N].  Returns out_local (q, owned, N).
        project_first (default: 2 N <= C): Z = x [W_0 | ... | W_{K-1}] locally, then Horner (mode 0) / Clenshaw (mode 1) on N-wide rows --
        hops and halo messages move N floats per row; otherwise K-1 hops on C-wide rows, then the projection.
        overlap=True: the exchange of hop k runs under the interior rows of the same hop and under the hops of the other
        time steps of a group of `depth` (per-time-step pipelining); same arithmetic row by row as overlap=False."""
        q, owned, C = x_local.shape
        assert owned == self.owned
        K, Cw, N = W.shape
        assert Cw == C and 1 <= K <= 32
        ops = self.ops
        pf = self.use_project_first(C, N, K) if project_first is None else (project_first and K > 1)
        out = torch.empty((q, owned, N), dtype=torch.float32, device=x_local.device)
        marks = [] if self.collect_stats else None
        self._mark(marks, "start")
        bk = bias_kind if bias_local is not None else 0
        if pf:
            bias_s = bias_local
            if bk == 2 and self._permuted:                    # the projection adds the bias at its OUTPUT row, which is in shard order
                bias_s = self._rows(bias_local.reshape(owned, N), self.local_perm)
            Z = ops.project_first(x_local.contiguous(), ops.weight_layout(W, 0), bias_s, bk, K, N, rowmap=self.inv32)      # (q, owned, K*N), shard order
            self._mark(marks, "projection_ms")
            zv = lambda j: ("view", lambda s, lo, hi: Z[s: s + 1, lo:hi, j * N:(j + 1) * N])      # noqa: E731
            if mode == 0:       # Horner: Y_j = Z_j + L Y_{j+1}
                prog = [(1.0, zv(K - 1 - i), 1.0, None, 0.0) for i in range(1, K)]
            else:               # Clenshaw: b_k = Z_k + 2 L b_{k+1} - b_{k+2};  out = Z_0 + L b_1 - b_2
                prog = [(1.0 if i == K - 1 else 2.0, ("buf", i - 2) if i >= 2 else None, -1.0, zv(K - 1 - i), 1.0) for i in range(1, K)]
            self._run_chain(q, N, prog, lambda s, dst: dst.copy_(Z[s, :, (K - 1) * N:]),
                            lambda s, bufs: self._to_caller_order(bufs[K - 1], out[s]), overlap, depth, marks)
        else:
            def finish(s, bufs):
                ops.project(bufs, W, bias_local, bk, owned, rowmap=self.perm32, out=out[s])
            self._run_chain(q, C, self._basis_prog(K, mode), lambda s, dst: self._to_shard_order(x_local[s], dst), finish, overlap, depth, marks)
        self._close_stats(marks)
        return out

    def forward_simple(self, x_local, W, bias_local, bias_kind, mode, project_first=None):
        """The same layer without overlap: blocking exchange, then the hop on all owned rows (what the overlapped form is
        tested against, bit for bit: same operand labels, same order of every row's sum)."""
        return self.forward(x_local, W, bias_local, bias_kind, mode, overlap=False, project_first=project_first)

    def basis(self, x_local, K, mode, overlap=True, depth=2):
        """The K terms of the working basis on the owned rows, caller's row order: L^k x (mode 0) or T_k(L) x (mode 1) -> (K, q, owned, C)"""
        q, owned, C = x_local.shape
        assert owned == self.owned
        B = torch.empty((K, q, owned, C), dtype=torch.float32, device=x_local.device)

        def finish(s, bufs):
            for k in range(K):
                self._to_caller_order(bufs[k], B[k, s])
        self._run_chain(q, C, self._basis_prog(K, mode), lambda s, dst: self._to_shard_order(x_local[s], dst), finish, overlap, depth, None)
        return B

    # ------------------------------------------------------------------ layer and its gradients (reference basis)
    def layer(self, x_local, W_ref, bias_local, bias_kind, mode, **kw):
        """forward() with the weight in the REFERENCE's basis (tgcn/nn/gcn.py:39,113,194): the monomial fold of the dense-L classes' recursion
        (mode 0) happens here, inside, as in the single-GPU layer function"""
        return self.forward(x_local, self.ops.fold(W_ref.contiguous()) if mode == 0 else W_ref.contiguous(), bias_local, bias_kind, mode, **kw)

    def layer_backward(self, x_local, W_ref, g_local, bias_kind, mode, needs=(True, True, True), grad_group=None, overlap=True, depth=2):
        """Gradients of `layer` w.r.t. (x_local, W_ref, bias_local); COLLECTIVE (every rank of the group, same `needs`).
        dX = sum_k T_k(L^T) g W_k^T is the layer on the transposed shard (reverse halo exchange); dW_k = (T_k(L) x)^T g = x^T (T_k(L^T) g)
        is contracted on the side with the narrower rows -- where the forward projected first (2 N <= C) the adjoint terms G_k = T_k(L^T) g
        (N-wide hops on L^T) give BOTH gradients: dX = sum_k G_k W_k^T and dW_k = x^T G_k; otherwise the basis T_k(L) x is recomputed.
        dW is summed over the ranks of `grad_group` (default: the shard's group); so is a per-channel dbias; a per-vertex dbias is this
        rank's rows."""
        ops = self.ops
        K, C, N = W_ref.shape
        q, owned, _ = x_local.shape
        Wt = ops.fold(W_ref.contiguous()) if mode == 0 else W_ref.contiguous()
        g = g_local.contiguous()
        g2d = g.reshape(q * owned, N)
        x2d = x_local.contiguous().reshape(q * owned, C)
        grad_group = self.group if grad_group is None else grad_group
        gx = gW = gb = None
        T = self.transpose() if (needs[0] or (needs[1] and self.use_project_first(C, N, K))) else None
        if self.use_project_first(C, N, K):
            if needs[0] or needs[1]:
                G = T.basis(g, K, mode, overlap=overlap, depth=depth)                       # (K, q, owned, N)
                terms = [G[k].reshape(q * owned, N) for k in range(K)]
                if needs[0]:
                    gx = ops.project(terms, ops.weight_layout(Wt, 1), None, 0, owned).reshape(q, owned, C)
                if needs[1]:
                    gW = ops.weight_layout(ops.wgrad(terms, x2d), 1)                          # (K, N, C) = G_k^T x  ->  (K, C, N)
        else:
            if needs[1]:
                B = self.basis(x_local.contiguous(), K, mode, overlap=overlap, depth=depth)
                gW = ops.wgrad([B[k].reshape(q * owned, C) for k in range(K)], g2d)
            if needs[0]:
                gx = T.forward(g, ops.weight_layout(Wt, 1), None, 0, mode, overlap=overlap, depth=depth)
        if needs[1]:
            gW = self._all_reduce(gW.contiguous(), grad_group)
            if mode == 0:
                gW = ops.fold(gW, transpose=True)
        if needs[2] and bias_kind:
            if bias_kind == 1:
                gb = self._all_reduce(g.sum(dim=(0, 1)), grad_group)
            else:
                gb = g.sum(dim=0)
        return gx, gW, gb


# ------------------------------------------------------------------------------------ module surface
class ShardedChebFn(torch.autograd.Function):
    """out_local = layer(x_local) on a vertex shard as a differentiable op (collective in both directions)."""

    @staticmethod
    def forward(ctx, x3, W, bias_local, sh, mode, bias_kind, grad_group, overlap=True, depth=2):
        x3 = x3.contiguous()
        ctx.save_for_backward(x3, W)
        ctx.sh, ctx.mode, ctx.bias_kind, ctx.grad_group, ctx.overlap, ctx.depth = sh, mode, bias_kind, grad_group, overlap, depth
        ctx.bias_shape = None if bias_local is None else bias_local.shape
        return sh.layer(x3, W, bias_local, bias_kind if bias_local is not None else 0, mode, overlap=overlap, depth=depth)

    @staticmethod
    def backward(ctx, g):
        x3, W = ctx.saved_tensors
        needs = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.bias_shape is not None and ctx.needs_input_grad[2])
        gx, gW, gb = ctx.sh.layer_backward(x3, W, g, ctx.bias_kind, ctx.mode, needs=needs, grad_group=ctx.grad_group, overlap=ctx.overlap, depth=ctx.depth)
        if gb is not None:
            gb = gb.reshape(ctx.bias_shape)
        return gx, gW, gb, None, None, None, None, None, None


class _OwnedRowsFn(torch.autograd.Function):
    """rows [lo, hi) of a per-vertex parameter of GLOBAL shape (n, g).  Backward: the gradient of the whole parameter -- this rank's rows
    filled in, and (sync) summed over the group so that every rank holds the same full gradient and replicated optimizers stay in step."""

    @staticmethod
    def forward(ctx, full, lo, hi, group, sync):
        ctx.lo, ctx.hi, ctx.n, ctx.group, ctx.sync = lo, hi, full.shape[0], group, sync
        return full[lo:hi].contiguous()

    @staticmethod
    def backward(ctx, g):
        out = torch.zeros((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        out[ctx.lo: ctx.hi] = g
        if ctx.sync:
            if out.is_cuda and dist.get_backend(ctx.group) != "nccl":       # host transport: through host memory (VertexShardedCheb._post)
                h = out.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=ctx.group)
                out.copy_(h)
            else:
                dist.all_reduce(out, op=dist.ReduceOp.SUM, group=ctx.group)
        return out, None, None, None, None


class _ShardedMixin:
    """Vertex sharding behind the reference's module surface (examples/pytorch_based/pytorch_hcp_tgcn.py:270-273 wraps the model and the
    caller stays unchanged): same constructor arguments + the process group, `weight` / `bias` with their GLOBAL shapes (replicated on
    every rank: a reference state_dict loads), forward(x_local) -> out_local on the rows this rank owns (`owned_rows`).  Every rank
    constructs the module with the same global L; the shard -- partition, halo lists, operands -- is built on the first forward (the
    device of x), collectively, and with it the parameters are broadcast from the group's first rank (as DataParallel replicates device
    0's, pytorch_hcp_tgcn.py:271).  world size 1: the single-GPU module's own path (compact plans, one-launch kernels, ...)."""

    def _init_sharding(self, group, exchange, ops, sync_bias_grad, grad_group, sync_init, bounds=None, row_multiple=1):
        self._group, self._exchange, self._shard_ops = group, exchange, ops
        self._bounds, self._row_multiple = bounds, row_multiple
        self._grad_group = grad_group
        self._sync_bias_grad, self._sync_init = sync_bias_grad, sync_init
        self._shards = {}
        self.overlap, self.depth = True, 2        # exchange of a hop under the interior rows and the other time steps of a group of `depth`
        self.force_sharded = False                # True: the exchange path also at world size 1 (first contact with the backend on one rank)

    def shard(self, device):
        device = torch.device(device)
        sh = self._shards.get(str(device))
        if sh is None:
            n, row, col, val = _coo_of(self.L, device)
            sh = self._shards[str(device)] = VertexShardedCheb(n, row, col, val, group=self._group, device=device, exchange=self._exchange,
                                                               ops=self._shard_ops, bounds=self._bounds, row_multiple=self._row_multiple)
            self._broadcast_parameters(sh)
        return sh

    def _broadcast_parameters(self, sh):
        """once per module: every rank takes the parameters of the group's first rank (nn.DataParallel replicates device 0's, pytorch_hcp_tgcn.py:271)"""
        if self._sync_init and sh.world > 1 and not getattr(self, "_params_synced", False):
            src = sh.peer[0]
            for p in self.parameters():
                t = p.data if p.data.device == sh.comm_device else p.data.to(sh.comm_device)
                dist.broadcast(t, src=src, group=self._group)
                if t is not p.data:
                    p.data.copy_(t)
        self._params_synced = True

    def owned_rows(self, device):
        sh = self.shard(device)
        return sh.lo, sh.hi

    def _single_gpu(self):
        return self._shard_ops is None and not self.force_sharded and (not dist.is_initialized() or dist.get_world_size(self._group) == 1)

    def _sharded_layer(self, x3, W_kcn, bias_kind, mode=0, sh=None):
        sh = self.shard(x3.device) if sh is None else sh
        bias_local = None
        if self.bias is not None:
            if bias_kind == 2:
                bias_local = _OwnedRowsFn.apply(self.bias.reshape(sh.n, -1), sh.lo, sh.hi, self._grad_group or self._group, self._sync_bias_grad)
            else:
                bias_local = self.bias.reshape(-1)
        return ShardedChebFn.apply(x3, W_kcn, bias_local, sh, mode, bias_kind if self.bias is not None else 0, self._grad_group, self.overlap, self.depth)


class _Sharded(_ShardedMixin):
    """constructor of the single-GPU class (the reference's arguments) + the sharding keywords; forward on the owned rows"""

    def __init__(self, *args, group=None, exchange="auto", ops=None, sync_bias_grad=True, grad_group=None, sync_init=True, bounds=None,
                 row_multiple=1, **kw):
        super().__init__(*args, **kw)
        self._init_sharding(group, exchange, ops, sync_bias_grad, grad_group, sync_init, bounds, row_multiple)



class ShardedTGCNCheb(_Sharded, _nn.TGCNCheb):
    """tgcn/nn/gcn.py:8-79 vertex-sharded: x_local (q, owned, f) -> (q, owned, g); weight (K, f, g), bias (1, n, g) with their GLOBAL shapes."""

    def forward(self, x_local):
        if self._single_gpu():
            return _nn.TGCNCheb.forward(self, x_local)
        return self._sharded_layer(x_local.float(), self.weight, 2)


class ShardedTGCNCheb_H(_Sharded, _nn.TGCNCheb_H):
    """tgcn/nn/gcn.py:82-154 vertex-sharded: x_local (q, owned, h[, f]) -> (q, owned, g); weight (K, H, f, g), bias (1, n, g) global.  With a long
    horizon and few output channels (examples/pytorch_based/pytorch_hcp_tgcn.py:103-104) the shard projects first: hops and halo messages on g-wide rows."""

    def forward(self, x_local):
        if self._single_gpu():
            return _nn.TGCNCheb_H.forward(self, x_local)
        if x_local.dim() == 3:
            x_local = x_local.unsqueeze(3)
        q, rows, h, f = x_local.shape
        W = self.weight.reshape(self.weight.shape[0], h * f, self.out_channels)
        return self._sharded_layer(x_local.float().reshape(q, rows, h * f), W, 2)


class ShardedGCNCheb(_Sharded, _nn.GCNCheb):
    """tgcn/nn/gcn.py:158-237 vertex-sharded: x_local (q, owned[, f]) -> (q, owned, g); weight (K, f, g), bias (1, 1, g)."""

    def forward(self, x_local):
        if self._single_gpu():
            return _nn.GCNCheb.forward(self, x_local)
        if x_local.dim() == 2:
            x_local = x_local.unsqueeze(2)
        return self._sharded_layer(x_local.float(), self.weight, 1)


# ---- the edge-list classes (tgcn/nn/gcn.py:348-538): the graph arrives with every forward
class _ShardedEdge(_ShardedMixin):
    """ChebConv / ChebTimeConv vertex-sharded (examples/pytorch_geo_based/pygeo_hcp.py:85,129 call them as conv(x, edge_index); :475-477 wraps the
    model for several GPUs).  Every rank passes the same GLOBAL edge_index (and edge_weight); the shard of a graph -- operand values by the
    library's edge normalisation, partition, halo lists -- is built collectively the first time that edge_index is seen and kept (least recently
    used of 4: pygeo_hcp.py:284 swaps the graph per subject).  The vertex count is not in x_local: ask `owned_rows(device, edge_index,
    num_vertices)` first (it builds the shard), slice x, then call forward.  A learnable edge_weight is refused (its gradient lives in the
    single-GPU layer function)."""

    MAX_SHARDS = 4

    def __init__(self, *args, group=None, exchange="auto", ops=None, grad_group=None, sync_init=True, row_multiple=1, **kw):
        super().__init__(*args, **kw)
        self._init_sharding(group, exchange, ops, True, grad_group, sync_init, None, row_multiple)
        import collections
        self._edge_shards = collections.OrderedDict()

    @staticmethod
    def _key(t):
        return None if t is None else (t.data_ptr(), t._version, tuple(t.shape), str(t.device))

    def _edge_shard(self, device, edge_index, edge_weight, num_vertices):
        if edge_weight is not None and edge_weight.requires_grad:
            raise ValueError("sharded edge-list layers take fixed edge weights (detach() them); the gradient w.r.t. edge_weight is the single-GPU modules'")
        device = torch.device(device)
        key = (self._key(edge_index), self._key(edge_weight), str(device))
        hit = self._edge_shards.get(key)
        if hit is not None:
            self._edge_shards.move_to_end(key)
            return hit[0]
        if num_vertices is None:
            raise ValueError("first use of this edge_index: pass num_vertices (the x of a shard holds the owned rows only) -- owned_rows(device, edge_index, n)")
        ops = _HIP_OPS if self._shard_ops is None else self._shard_ops
        row, col, val = ops.edge_coo(edge_index, edge_weight, int(num_vertices), device)
        sh = VertexShardedCheb(int(num_vertices), row, col, val, group=self._group, device=device, exchange=self._exchange, ops=self._shard_ops,
                               row_multiple=self._row_multiple)
        self._broadcast_parameters(sh)
        self._edge_shards[key] = (sh, edge_index, edge_weight)          # the sources stay alive: their addresses cannot be handed to another graph
        while len(self._edge_shards) > self.MAX_SHARDS:
            self._edge_shards.popitem(last=False)
        return sh

    def owned_rows(self, device, edge_index, num_vertices, edge_weight=None):
        sh = self._edge_shard(device, edge_index, edge_weight, num_vertices)
        return sh.lo, sh.hi


class ShardedChebConv(_ShardedEdge, _nn.ChebConv):
    """tgcn/nn/gcn.py:348-442 vertex-sharded: forward(x_local (q, owned[, f]), edge_index (2, E) GLOBAL, edge_weight=None) -> (q, owned, g)."""

    def forward(self, x_local, edge_index, edge_weight=None, num_vertices=None):
        if self._single_gpu():
            return _nn.ChebConv.forward(self, x_local, edge_index, edge_weight)
        sh = self._edge_shard(x_local.device, edge_index, edge_weight, num_vertices)
        if x_local.dim() < 3:
            x_local = x_local.unsqueeze(-1)
        return self._sharded_layer(x_local.float(), self.weight, 1, mode=1, sh=sh)


class ShardedChebTimeConv(_ShardedEdge, _nn.ChebTimeConv):
    """tgcn/nn/gcn.py:445-538 vertex-sharded: forward(x_local (q, owned, h[, f]), edge_index, edge_weight=None) -> (q, owned, g)."""

    def forward(self, x_local, edge_index, edge_weight=None, num_vertices=None):
        if self._single_gpu():
            return _nn.ChebTimeConv.forward(self, x_local, edge_index, edge_weight)
        sh = self._edge_shard(x_local.device, edge_index, edge_weight, num_vertices)
        if x_local.dim() < 4:
            x_local = x_local.unsqueeze(-1)
        q, rows, h, f = x_local.shape
        W = self.weight.reshape(self.weight.shape[0], h * f, self.out_channels)
        return self._sharded_layer(x_local.float().reshape(q, rows, h * f), W, 1, mode=1, sh=sh)
