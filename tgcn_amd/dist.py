"""Multi-GPU sharding of the Chebyshev layer: one process per GPU, torch.distributed ("nccl" is RCCL on ROCm).

The reference's only multi-GPU mechanism is nn.DataParallel's batch split (examples/pytorch_based/
pytorch_hcp_tgcn.py:270-273).  The path shards two ways:

  * by sample / time step: every column of X is filtered independently and the projection contracts inside a
    sample, so ranks that hold the CSR need NO data-path communication (`shard_time_steps`; bench.py --gpus N);
  * by vertices: 1-D row partition of L-hat, X and out; one exchange of the previous hop's cut-edge neighbour
    rows per hop (`VertexShardedCheb`): point-to-point halo rows when the cut is small, an RCCL all-gather of the
    owned row blocks when the halo is most of the graph (R-MAT);
  * both at once (`hybrid_groups`): vertex shards inside a group of ranks, time steps across groups.

Vertex sharding overlaps communication with compute: a shard's rows are kept in the order
[interior | boundary]; the interior rows of hop k (no remote column) run while the halo rows of hop k-1 are in flight,
messages are packed by a HIP kernel and received in place, time steps are pipelined in groups of `depth` (the exchange
of one under the hops of the others), and every buffer lives as long as the object.

The compute callables are injectable so the communication logic is testable with gloo on CPU against the oracle
(tests/test_dist_gloo.py: overlapped against non-overlapped form bit for bit); the defaults are the HIP entry points.
"""
import torch
import torch.distributed as dist


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


def balanced_row_bounds(row, n, world):
    """world+1 row boundaries with ~equal stored entries per shard (power-law rows => not equal row counts)."""
    counts = torch.bincount(row, minlength=n)
    cum = torch.cumsum(counts, 0)
    total = int(cum[-1].item()) if n else 0
    marks = (torch.arange(1, world, device=row.device, dtype=torch.int64) * total) // world
    inner = (torch.searchsorted(cum, marks) + 1).clamp_(max=n)
    b = torch.cat([torch.zeros(1, dtype=torch.int64, device=row.device), inner,
                   torch.full((1,), n, dtype=torch.int64, device=row.device)])
    return torch.cummax(b, 0)[0]


def _default_operand(n_rows, n_cols, row, col, val, device):
    from .graph import GraphOperand
    return GraphOperand.from_coo(n_rows, row, col, val, device, n_cols=n_cols)


def _default_hop(op, x, z, alpha, beta, out):
    from . import functional as F
    return F.csr_hop(op, x, z=z, alpha=alpha, beta=beta, out=out)


def _default_project(terms, W, bias, bias_kind, n_vertices):
    from . import functional as F
    return F.cheb_project(terms, W, bias, bias_kind, n_vertices)


def _default_pack(src, idx, out):
    """out[i, :] = src[idx[i], :] -- the rows a peer asked for, packed for one message (HIP kernel tgcn_pack_rows_f32)"""
    from . import functional as F
    return F.pack_rows(src, idx, out)


class VertexShardedCheb:
    """Vertex-sharded layer forward.  Every rank passes the same global COO of L-hat (or at least its own rows);
    rank r owns rows [bounds[r], bounds[r+1]).

    exchange = "halo": ext operand = [owned rows ; halo rows], halo rows arrive by point-to-point messages from
                       their owners (index lists agreed once at construction);
               "allgather": ext operand = all shards' row blocks padded to the largest, one all_gather per hop;
               "auto": allgather when the halo is more than half of the remote vertices.
    """

    def __init__(self, n, row, col, val, group=None, device=None, exchange="auto", make_operand=_default_operand,
                 hop_fn=_default_hop, project_fn=_default_project, pack_fn=_default_pack):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # point-to-point peers are named by GLOBAL rank; inside a sub-group (hybrid layout) translate the group index
        self.peer = [dist.get_global_rank(group, p) if group is not None else p for p in range(self.world)]
        self.device = torch.device(device) if device is not None else row.device
        self.hop_fn, self.project_fn, self.pack_fn = hop_fn, project_fn, pack_fn
        self._bufs = {}
        self.collect_stats = False      # True: forward() times its phases on this rank (device events / wall clock) into self.stats
        self.stats = None
        row, col, val = row.to(self.device), col.to(self.device), val.to(self.device)
        self.n = n
        self.bounds = balanced_row_bounds(row, n, self.world)
        b = self.bounds.tolist()
        self.lo, self.hi = b[self.rank], b[self.rank + 1]
        self.owned = self.hi - self.lo
        self.n_max = max(b[i + 1] - b[i] for i in range(self.world))
        mine = (row >= self.lo) & (row < self.hi)
        r, c, v = row[mine] - self.lo, col[mine], val[mine]
        remote = (c < self.lo) | (c >= self.hi)
        halo_ids = torch.unique(c[remote])                       # sorted global ids
        self.halo = int(halo_ids.numel())
        n_remote = n - self.owned
        if exchange == "auto":
            frac = torch.tensor([self.halo / max(n_remote, 1)], dtype=torch.float64, device=self.device)
            dist.all_reduce(frac, op=dist.ReduceOp.MAX, group=group)
            exchange = "allgather" if frac.item() > 0.5 else "halo"
        self.exchange = exchange
        if exchange == "halo":
            c_local = torch.where(remote, self.owned + torch.searchsorted(halo_ids, c), c - self.lo)
            self.n_ext = self.owned + self.halo
            owner = torch.searchsorted(self.bounds[1:].contiguous(), halo_ids, right=True)
            self.recv_counts = torch.bincount(owner, minlength=self.world).tolist()
            want = [None] * self.world       # what I need from each peer, as row ids local to the peer
            off = 0
            for p in range(self.world):
                ids = halo_ids[off: off + self.recv_counts[p]] - b[p]
                want[p] = ids.cpu()
                off += self.recv_counts[p]
            gathered = [None] * self.world
            dist.all_gather_object(gathered, want, group=group)
            self.send_idx = [gathered[p][self.rank].to(self.device) for p in range(self.world)]
        else:
            # position of global vertex g in the gathered operand: owner(g) * n_max + (g - bounds[owner])
            owner = torch.searchsorted(self.bounds[1:].contiguous(), c, right=True)
            c_local = owner * self.n_max + (c - self.bounds[owner])
            self.n_ext = self.world * self.n_max
        self.op = make_operand(self.owned, self.n_ext, r, c_local, v, self.device)
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
        self.local_perm = order
        inv = torch.empty_like(order)
        inv[order] = torch.arange(self.owned, device=self.device)
        self.local_inv = inv
        r2 = inv[r]                                                                  # row in the [interior | boundary] order
        if exchange == "halo":
            c2 = torch.where(c_local < self.owned, inv[c_local.clamp(max=max(self.owned - 1, 0))], c_local)    # owned columns move with their rows
            self.send_idx_l = [inv[idx] if idx.numel() else idx for idx in self.send_idx]
        else:
            c2 = c_local
        if exchange == "halo":
            is_int = r2 < self.n_int
            self.op_int = make_operand(self.n_int, self.n_ext, r2[is_int], c2[is_int], v[is_int], self.device) if self.n_int else None
            nb = self.owned - self.n_int
            self.op_bnd = make_operand(nb, self.n_ext, r2[~is_int] - self.n_int, c2[~is_int], v[~is_int], self.device) if nb else None
            # the same rows as one operand (non-overlapped form): identical labels, so both forms sum every row in the same order
            self.op_all = make_operand(self.owned, self.n_ext, r2, c2, v, self.device)
        else:                       # all-gather: every row waits for the gathered operand, one operand serves both forms
            self.op_int, self.op_bnd, self.op_all = None, self.op, self.op

    # ------------------------------------------------------------------ layer, overlapped
    def _buf(self, name, shape):
        """buffers of the overlapped path live as long as the object (no allocation per hop)"""
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
            self.pack_fn(src, idx, out)
        return out

    # ------------------------------------------------------------------ diagnostics
    def describe(self):
        """what this rank exchanges per hop and time step: enough to tell a slow link from a wrong partition in one record"""
        C4 = 4
        d = dict(rank=self.rank, exchange=self.exchange, owned_rows=self.owned, interior_rows=self.n_int, halo_rows=self.halo,
                 ext_rows=self.n_ext, nnz=self.op.nnz if hasattr(self.op, "nnz") else None)
        if self.exchange == "halo":
            d["recv_rows_per_peer"] = list(self.recv_counts)
            d["send_rows_per_peer"] = [int(i.numel()) for i in self.send_idx]
            d["bytes_per_channel_in"] = C4 * sum(self.recv_counts)
        else:
            d["allgather_block_rows"] = self.n_max
            d["bytes_per_channel_in"] = C4 * self.n_max * (self.world - 1)
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

    def _start_exchange(self, slot, k, src_ext):
        """Start fetching the remote rows of hop tensor `src_ext` ((1, n_ext, C), owned part valid) into its own halo region;
        returns the work handles.  Messages are packed by the pack kernel into per-peer send buffers and received IN PLACE."""
        C = src_ext.shape[2]
        if self.exchange == "halo":
            ops = []
            off = self.owned
            for p in range(self.world):
                cnt = self.recv_counts[p]
                if p != self.rank and cnt:
                    ops.append(dist.P2POp(dist.irecv, src_ext[0, off: off + cnt], self.peer[p], group=self.group))
                off += cnt
                idx = self.send_idx_l[p]
                if p != self.rank and idx.numel():
                    sb = self._buf(("send", slot, k & 1, p), (idx.numel(), C))
                    self.pack_fn(src_ext[0], idx, sb)
                    ops.append(dist.P2POp(dist.isend, sb, self.peer[p], group=self.group))
            return dist.batch_isend_irecv(ops) if ops else []
        raise AssertionError

    def forward(self, x_local, W, bias_local, bias_kind, mode, overlap=True, depth=2):
        """x_local: (q, owned, C) rows of this shard; W: (K, C, N) (already monomial-folded for mode 0);
        bias_local: per channel [N] or this shard's rows [owned, N].  Returns out_local (q, owned, N).
        overlap=True: the exchange of hop k runs under the interior rows of the same hop and under the hops of the other
        time steps of a group of `depth` (per-time-step pipelining); same arithmetic row by row as overlap=False."""
        if not overlap:
            return self.forward_simple(x_local, W, bias_local, bias_kind, mode)
        q, owned, C = x_local.shape
        assert owned == self.owned
        K, _, N = W.shape
        ni = self.n_int
        out = torch.empty((q, owned, N), dtype=torch.float32, device=x_local.device)
        allg = self.exchange != "halo"
        bias_l = bias_local
        if bias_kind == 2 and bias_local is not None and not allg:
            bias_l = self._rows(bias_local, self.local_perm)
        marks = [] if self.collect_stats else None
        self._mark(marks, "start")
        for t0 in range(0, q, depth):
            steps = range(t0, min(q, t0 + depth))
            # hop tensors of a time step: K buffers of (1, n_ext, C) [halo] or (n_max, C) + one gathered copy [all-gather]
            if allg:
                mine = {s: [self._buf(("mine", s - t0, k), (self.n_max, C)) for k in range(K)] for s in steps}
                ext = {s: self._buf(("ext", s - t0), (1, self.n_ext, C)) for s in steps}
                for s in steps:
                    mine[s][0][: owned].copy_(x_local[s])
                    mine[s][0][owned:].zero_()
                own = lambda s, k: mine[s][k][: owned].unsqueeze(0)
            else:
                exts = {s: [self._buf(("ext", s - t0, k), (1, self.n_ext, C)) for k in range(K)] for s in steps}
                for s in steps:
                    self._rows(x_local[s], self.local_perm, exts[s][0][0, : owned])
                own = lambda s, k: exts[s][k][:, : owned]
            for k in range(1, K):
                works = {}
                for s in steps:                                       # 1. start every exchange of this hop level
                    if allg:
                        works[s] = [dist.all_gather_into_tensor(ext[s].view(self.world * self.n_max, C), mine[s][k - 1], group=self.group, async_op=True)]
                    else:
                        works[s] = self._start_exchange(s - t0, k, exts[s][k - 1])
                self._mark(marks, "pack_and_post_ms")
                if not allg and ni:
                    for s in steps:                                   # 2. interior rows: no remote column, no wait
                        z = own(s, k - 2)[:, : ni] if (mode != 0 and k >= 2) else None
                        self.hop_fn(self.op_int, exts[s][k - 1], z, 2.0 if z is not None else 1.0, -1.0 if z is not None else 0.0, own(s, k)[:, : ni])
                    self._mark(marks, "interior_hops_ms")
                for s in steps:                                       # 3. boundary rows as their halo arrives
                    for w in works[s]:
                        w.wait()
                    self._mark(marks, "exchange_wait_ms")
                    src = ext[s] if allg else exts[s][k - 1]
                    opb = self.op if allg else self.op_bnd
                    if opb is not None:
                        z = own(s, k - 2)[:, ni:] if (mode != 0 and k >= 2) else None
                        self.hop_fn(opb, src, z, 2.0 if z is not None else 1.0, -1.0 if z is not None else 0.0, own(s, k)[:, ni:])
                    self._mark(marks, "boundary_hops_ms")
            for s in steps:
                o = self.project_fn([own(s, k).reshape(owned, C) for k in range(K)], W, bias_l, bias_kind, owned)
                o = o.reshape(owned, N)
                if allg:
                    out[s] = o
                else:
                    self._rows(o, self.local_inv, out[s])
            self._mark(marks, "projection_ms")
        self._close_stats(marks)
        return out

    # ------------------------------------------------------------------ layer, one exchange after the other
    def forward_simple(self, x_local, W, bias_local, bias_kind, mode):
        """The same layer without overlap: blocking exchange, then the hop on all owned rows (what the overlapped form is
        tested against, bit for bit: same operand labels, same order of every row's sum)."""
        q, owned, C = x_local.shape
        assert owned == self.owned
        K, _, N = W.shape
        halo = self.exchange == "halo"
        out = torch.empty((q, owned, N), dtype=torch.float32, device=x_local.device)
        bias_l = bias_local
        if halo and bias_kind == 2 and bias_local is not None:
            bias_l = self._rows(bias_local, self.local_perm)
        marks = [] if self.collect_stats else None
        self._mark(marks, "start")
        for s in range(q):
            xs = self._rows(x_local[s], self.local_perm) if halo else x_local[s]
            terms = [xs.unsqueeze(0)]
            for k in range(1, K):
                ext = torch.empty((1, self.n_ext, C), dtype=torch.float32, device=xs.device)
                if halo:
                    ext[0, : owned] = terms[k - 1][0]
                    for w in self._start_exchange(0, k, ext):
                        w.wait()
                else:
                    mine = torch.zeros((self.n_max, C), dtype=torch.float32, device=xs.device)
                    mine[: owned] = terms[k - 1][0]
                    dist.all_gather_into_tensor(ext.view(self.world * self.n_max, C), mine, group=self.group)
                self._mark(marks, "exchange_ms")
                y = torch.empty((1, owned, C), dtype=torch.float32, device=xs.device)
                z = terms[k - 2] if (mode != 0 and k >= 2) else None
                self.hop_fn(self.op_all, ext, z, 2.0 if z is not None else 1.0, -1.0 if z is not None else 0.0, y)
                terms.append(y)
                self._mark(marks, "hops_ms")
            o = self.project_fn([t.reshape(owned, C) for t in terms], W, bias_l, bias_kind, owned).reshape(owned, N)
            if halo:
                self._rows(o, self.local_inv, out[s])
            else:
                out[s] = o
            self._mark(marks, "projection_ms")
        self._close_stats(marks)
        return out
