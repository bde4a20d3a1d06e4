"""Multi-GPU sharding of the Chebyshev layer: one process per GPU, torch.distributed ("nccl" is RCCL on ROCm).

The reference's only multi-GPU mechanism is nn.DataParallel's batch split (examples/pytorch_based/
pytorch_hcp_tgcn.py:270-273).  The path shards two ways:

  * by sample / time step: every column of X is filtered independently and the projection contracts inside a
    sample, so ranks that hold the CSR need NO data-path communication (`shard_time_steps`; bench.py --gpus N);
  * by vertices: 1-D row partition of L-hat, X and out; one exchange of the previous hop's cut-edge neighbour
    rows per hop (`VertexShardedCheb`): point-to-point halo rows when the cut is small, an RCCL all-gather of the
    owned row blocks when the halo is most of the graph (R-MAT);
  * both at once (`hybrid_groups`): vertex shards inside a group of ranks, time steps across groups.

The compute callables are injectable so the communication logic is testable with gloo on CPU against the oracle
(tests/test_dist_gloo.py); the defaults are the HIP entry points.
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


class VertexShardedCheb:
    """Vertex-sharded layer forward.  Every rank passes the same global COO of L-hat (or at least its own rows);
    rank r owns rows [bounds[r], bounds[r+1]).

    exchange = "halo": ext operand = [owned rows ; halo rows], halo rows arrive by point-to-point messages from
                       their owners (index lists agreed once at construction);
               "allgather": ext operand = all shards' row blocks padded to the largest, one all_gather per hop;
               "auto": allgather when the halo is more than half of the remote vertices.
    """

    def __init__(self, n, row, col, val, group=None, device=None, exchange="auto", make_operand=_default_operand,
                 hop_fn=_default_hop, project_fn=_default_project):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # point-to-point peers are named by GLOBAL rank; inside a sub-group (hybrid layout) translate the group index
        self.peer = [dist.get_global_rank(group, p) if group is not None else p for p in range(self.world)]
        self.device = torch.device(device) if device is not None else row.device
        self.hop_fn, self.project_fn = hop_fn, project_fn
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

    # ------------------------------------------------------------------ exchange
    def _fill_ext(self, ext, p_owned):
        """ext: (q, n_ext, C) buffer whose owned part already holds the previous hop; fetch the remote rows."""
        q, _, C = ext.shape
        if self.exchange == "halo":
            sends = [ext[:, idx, :].contiguous() if idx.numel() else None for idx in self.send_idx]
            recvs = [torch.empty((q, cnt, C), dtype=ext.dtype, device=ext.device) if cnt else None for cnt in self.recv_counts]
            ops = []
            for p in range(self.world):
                if p == self.rank:
                    continue
                if recvs[p] is not None:
                    ops.append(dist.P2POp(dist.irecv, recvs[p], self.peer[p], group=self.group))
                if sends[p] is not None:
                    ops.append(dist.P2POp(dist.isend, sends[p], self.peer[p], group=self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            off = self.owned
            for p in range(self.world):
                if recvs[p] is not None:
                    ext[:, off: off + self.recv_counts[p], :] = recvs[p]
                    off += self.recv_counts[p]
        else:
            assert q == 1, "all-gather exchange runs one sample per pass"
            mine = torch.zeros((self.n_max, C), dtype=ext.dtype, device=ext.device)
            mine[: self.owned] = p_owned[0]
            dist.all_gather_into_tensor(ext.view(self.world * self.n_max, C), mine, group=self.group)

    def _owned_view(self, ext):
        if self.exchange == "halo":
            return ext[:, : self.owned, :]
        return ext[:, self.rank * self.n_max: self.rank * self.n_max + self.owned, :]

    # ------------------------------------------------------------------ layer
    def forward(self, x_local, W, bias_local, bias_kind, mode):
        """x_local: (q, owned, C) rows of this shard; W: (K, C, N) (already monomial-folded for mode 0);
        bias_local: per channel [N] or this shard's rows [owned, N].  Returns out_local (q, owned, N)."""
        q, owned, C = x_local.shape
        assert owned == self.owned
        K, _, N = W.shape
        out = torch.empty((q, owned, N), dtype=torch.float32, device=x_local.device)
        step = q if self.exchange == "halo" else 1
        for q0 in range(0, q, step):
            xs = x_local[q0: q0 + step].contiguous()
            terms = [xs]
            for k in range(1, K):
                ext = torch.empty((step, self.n_ext, C), dtype=torch.float32, device=xs.device)
                if self.exchange == "halo":
                    ext[:, : self.owned] = terms[k - 1]
                self._fill_ext(ext, terms[k - 1])
                y = torch.empty((step, owned, C), dtype=torch.float32, device=xs.device)
                if mode == 0 or k == 1:
                    self.hop_fn(self.op, ext, None, 1.0, 0.0, y)
                else:
                    self.hop_fn(self.op, ext, terms[k - 2], 2.0, -1.0, y)
                terms.append(y)
            o = self.project_fn([t.reshape(step * owned, C) for t in terms], W, bias_local, bias_kind, owned)
            out[q0: q0 + step] = o.reshape(step, owned, N)
        return out
