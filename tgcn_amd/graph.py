"""Graph operand of the Chebyshev layers: L-hat as device CSR + nnz-balanced row-block schedules.

Built ONCE per operand (the reference re-does `L.to(device)` and the degree normalisation on every forward:
tgcn/nn/gcn.py:141,223 and :408-413,505-510) and cached by the modules.  The preparation below is index
plumbing on torch tensors (sort / cumsum / searchsorted); every flop of the layer runs in libtgcn_hip.so.
"""

import ctypes as C
import threading

import torch

from . import _lib

ROW_THRESH = 32         # rows with more stored entries are processed as column-ordered segments
SEG_LEN = 32            # stored entries per segment (measured on cfg5: 32 beats 16/64/128/256, tools/hop_bench.py)
SEG_MODE = 0            # 0: one lane group per segment (shipped); 1: one WAVE per segment of SEG_LEN * (64 / lanes) entries, pieces folded in
                        # the wave -- 4.7x fewer partial rows but the column-ordered sweep loses its locality: cfg5 hop 3.82 -> 4.44 ms (docs/EXPERIMENTS.md A.3)
SEG_KEY = "first"       # column of the segment used as its place in the processing order
WAVE_ROWS = True        # 16-lane schedules: rows with ROW_THRESH < entries <= 128 are whole-row segments of one WAVE each (its lane groups fold inside the
                        # wave: no partial rows, no fix-up for them; cfg5: fix-up 0.216 -> 0.155 ms per launch, hop unchanged); False: lane-group segments
HUGE_SLOTS = 64         # long rows with more segments than this get a whole workgroup in the fix-up
ROW_COST = 4            # per-row overhead of the balance model, in entry equivalents
DENSE_MAX_N = 256       # operands up to this size that store >= 1/4 of their entries also keep a dense copy
MAX_BLOCKS_HINT = 2048  # below this many blocks, make blocks smaller rather than leave CUs idle
COMPACT_MIN_ROWS = 1 << 16     # operands at least this large ...
COMPACT_MIN_EMPTY = 0.125      # ... with at least this share of structurally empty rows keep compact hop tensors (CompactPlan)
BUILDER = "library"     # operands / schedules of device tensors: "library" = the HIP kernels of libtgcn_hip.so (csrc/device_build.h: own stable
                        # radix sort, prefix sums, marks); "torch" = the torch index ops below (always for CPU tensors; the cross-check)
BLOCK_ROWS_MAX = None   # cost of a row block at most this many entry-equivalents per lane group (None: 64 for rows up to 64 floats, else 256)


def _as_i32(t):
    return t.to(torch.int32).contiguous()


def _check_range(row, col, n, ncol):
    """One-off range check: an out-of-range vertex index would be an out-of-bounds read inside the kernels."""
    if row.numel() and (int(row.min()) < 0 or int(row.max()) >= n or int(col.min()) < 0 or int(col.max()) >= ncol):
        raise _lib.TgcnError("graph operand: vertex index outside [0, %d) x [0, %d)" % (n, ncol))


class Schedule:
    """Work schedule for one lane-group width (include/tgcn_hip.h: tgcn_csr_sched)."""

    def __init__(self, rowptr, n, lanes_per_row, edges=None, row_thresh=None, seg_len=None, seg_mode=None, n_cols=None, builder=None):
        defaults = (row_thresh is None and seg_len is None and seg_mode is None and ROW_THRESH == 32 and SEG_LEN == 32 and SEG_MODE == 0
                    and SEG_KEY == "first" and WAVE_ROWS and BLOCK_ROWS_MAX is None and ROW_COST == 4 and HUGE_SLOTS == 64 and MAX_BLOCKS_HINT == 2048)
        builder = BUILDER if builder is None else builder
        if builder == "library" and rowptr.is_cuda and edges is not None and defaults:
            self._build_library(rowptr, n, lanes_per_row, edges, n if n_cols is None else n_cols)
            return
        row_thresh = ROW_THRESH if row_thresh is None else row_thresh
        seg_len = max(SEG_LEN if seg_len is None else seg_len, 1)
        seg_mode = (SEG_MODE if seg_mode is None else seg_mode) if lanes_per_row < 64 else 0
        if seg_mode == 1:
            seg_len *= 64 // lanes_per_row          # a wave's lane groups share the segment
        dev = rowptr.device
        gpb = 256 // lanes_per_row
        deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
        is_seg = deg > row_thresh
        # ---- short rows: nnz-balanced row blocks
        cost = torch.where(is_seg, torch.zeros_like(deg), deg) + ROW_COST
        cum = torch.cumsum(cost, 0)
        total = int(cum[-1].item())
        # Large operands: small blocks for narrow rows.  The workgroups resident on an XCD work on consecutive row blocks, so the
        # rows they gather from -- for a graph with locality -- span (workgroups in flight) x (rows per block); with 256-cost blocks
        # that window was 13 MB at C = 64 and thrashed the 4 MB L2 (banded 10 M-vertex graph: 30 GB fetched, 4.31 ms; with 64-cost
        # blocks 3.02 ms).  No effect on graphs without locality (R-MAT: 4.35 vs 4.34 ms); rows wider than 64 floats keep the
        # larger blocks (sheet mesh at C = 1200: 0.380 vs 0.398 ms).  tgcn_sched_build (csrc/graph_build.h) applies the same rule.
        cap = BLOCK_ROWS_MAX if BLOCK_ROWS_MAX else (64 if lanes_per_row <= 16 else 256)
        target = max(gpb * 16, min(gpb * cap, -(-total // MAX_BLOCKS_HINT)))
        nblk = max(1, -(-total // target))
        if nblk > 1:
            marks = torch.arange(1, nblk, device=dev, dtype=torch.int64) * target
            inner = (torch.searchsorted(cum, marks) + 1).clamp_(max=n)
        else:
            inner = torch.zeros(0, dtype=torch.int64, device=dev)
        self.blk_row = _as_i32(torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), inner,
                                          torch.full((1,), n, dtype=torch.int64, device=dev)]))
        self.nblk = nblk
        # ---- longer rows: segments, long rows (several segments) first by decreasing segment count
        self.nwseg = 0
        wave_rows = None
        # 16-lane groups only: with 4-lane groups (cfg5n) a wave is 16 groups and a 40-entry row leaves most of them idle (hop +8 %)
        wave_max = 32 * (64 // lanes_per_row) if lanes_per_row == 16 else 0
        if WAVE_ROWS and wave_max > row_thresh and seg_mode == 0 and edges is not None:
            wave_rows = (is_seg & (deg <= wave_max)).nonzero().flatten()
            is_seg_long = is_seg & (deg > wave_max)
        else:
            is_seg_long = is_seg
        seg_rows = is_seg_long.nonzero().flatten()
        z1 = torch.zeros(1, dtype=torch.int32, device=dev)
        self.nseg = self.nlong = self.nhuge = self.npartial = 0
        self.seg_row = self.seg_e0 = self.seg_e1 = self.seg_slot = self.long_row = z1
        self.long_slot = torch.zeros(2, dtype=torch.int32, device=dev)
        if seg_rows.numel():
            nsegs = (deg[seg_rows] + seg_len - 1) // seg_len
            order = torch.argsort(nsegs, descending=True, stable=True)
            seg_rows, nsegs = seg_rows[order], nsegs[order]
            self.nlong = int((nsegs > 1).sum().item())
            self.nhuge = int((nsegs > HUGE_SLOTS).sum().item())
            self.nseg = int(nsegs.sum().item())
            first = torch.cumsum(nsegs, 0) - nsegs
            row_of = torch.repeat_interleave(seg_rows, nsegs)
            within = torch.arange(self.nseg, device=dev, dtype=torch.int64) - torch.repeat_interleave(first, nsegs)
            e0 = rowptr[row_of].to(torch.int64) + within * seg_len
            e1 = torch.minimum(e0 + seg_len, rowptr[row_of + 1].to(torch.int64))
            # slots: segments of long rows are numbered consecutively per row (long rows come first in `seg_rows`)
            self.npartial = int(nsegs[: self.nlong].sum().item())
            slot = torch.arange(self.nseg, device=dev, dtype=torch.int64)
            slot = torch.where(slot < self.npartial, slot, torch.full_like(slot, -1))
            if self.nlong:
                self.long_row = _as_i32(seg_rows[: self.nlong])
                self.long_slot = _as_i32(torch.cat([first[: self.nlong], first[self.nlong - 1: self.nlong] + nsegs[self.nlong - 1: self.nlong]]))
            # processing order: by first column
            if edges is not None:
                if SEG_KEY == "middle":
                    key = edges[(e0 + e1) // 2, 0].to(torch.int64)
                elif SEG_KEY == "last":
                    key = edges[e1 - 1, 0].to(torch.int64)
                elif SEG_KEY == "first_then_last":
                    key = edges[e0, 0].to(torch.int64) * (n + 1) + edges[e1 - 1, 0].to(torch.int64)
                else:
                    key = edges[e0, 0].to(torch.int64)
                perm = torch.argsort(key, stable=True)
                row_of, e0, e1, slot = row_of[perm], e0[perm], e1[perm], slot[perm]
            self.seg_row, self.seg_e0, self.seg_e1, self.seg_slot = _as_i32(row_of), _as_i32(e0), _as_i32(e1), _as_i32(slot)
        if wave_rows is not None and wave_rows.numel():
            # whole rows of one wave each, in order of their first column, in FRONT of the lane-group segments
            w0 = rowptr[wave_rows].to(torch.int64)
            w1 = rowptr[wave_rows + 1].to(torch.int64)
            wperm = torch.argsort(edges[w0, 0].to(torch.int64), stable=True)
            wr, w0, w1 = wave_rows[wperm], w0[wperm], w1[wperm]
            self.nwseg = int(wr.numel())
            had = self.nseg
            cat = lambda a, b: torch.cat([_as_i32(a), b[:had]]) if had else _as_i32(a)
            self.seg_row, self.seg_e0, self.seg_e1 = cat(wr, self.seg_row), cat(w0, self.seg_e0), cat(w1, self.seg_e1)
            self.seg_slot = cat(torch.full_like(wr, -1), self.seg_slot)
            self.nseg = had + self.nwseg
        self.lanes_per_row = lanes_per_row
        self.row_thresh = row_thresh
        self.seg_len = seg_len
        self.seg_mode = seg_mode
        # >= 1/8 structurally empty rows: their row blocks (mostly streaming zero writes) are dealt among the segment blocks
        self.row_mix = int(int((deg == 0).sum().item()) * 8 >= n)
        self.struct = _lib.SchedStruct(lanes_per_row, row_thresh, self.nblk, self.nseg, self.nlong, self.nhuge, self.npartial, seg_mode, self.row_mix, self.nwseg,
                                       self.blk_row.data_ptr(), self.seg_row.data_ptr(), self.seg_e0.data_ptr(),
                                       self.seg_e1.data_ptr(), self.seg_slot.data_ptr(), self.long_row.data_ptr(),
                                       self.long_slot.data_ptr())


def _sched_build_library(self, rowptr, n, lanes_per_row, edges, n_cols):
    """Schedule through tgcn_sched_build_csr (device kernels of the library), copied into torch-owned arrays"""
    import ctypes as C
    L = _lib.lib()
    nnz = int(rowptr[-1].item())
    csr = _lib.CsrStruct(int(n), nnz, rowptr.data_ptr(), edges.data_ptr(), None)
    # any row length with the right lane-group width: lanes -> a C that maps back to it (aligned rows of 4 floats per lane)
    C_row = lanes_per_row * 4 if lanes_per_row < 64 else 256
    assert L.tgcn_hop_lanes_per_row(C_row, 1) == lanes_per_row, (C_row, lanes_per_row)
    h = C.c_void_p()
    with torch.cuda.device(rowptr.device):
        torch.cuda.current_stream().synchronize()            # the library builds on the default stream
        _lib.check(L.tgcn_sched_build_csr(C.byref(csr), int(n_cols), C_row, 1, C.byref(h)))
        try:
            st = L.tgcn_sched_get(h).contents
            dev = rowptr.device
            i32 = lambda k: torch.empty(max(int(k), 1), dtype=torch.int32, device=dev)
            self.nblk, self.nseg, self.nlong, self.nhuge, self.npartial = st.nblk, st.nseg, st.nlong, st.nhuge, st.npartial
            self.blk_row = i32(st.nblk + 1)
            self.seg_row, self.seg_e0, self.seg_e1, self.seg_slot = i32(st.nseg), i32(st.nseg), i32(st.nseg), i32(st.nseg)
            self.long_row = i32(st.nlong)
            self.long_slot = torch.zeros(max(st.nlong + 1, 2), dtype=torch.int32, device=dev)
            _lib.check(L.tgcn_sched_copy(h, _lib.stream_ptr(), _lib.ptr(self.blk_row), _lib.ptr(self.seg_row), _lib.ptr(self.seg_e0),
                                         _lib.ptr(self.seg_e1), _lib.ptr(self.seg_slot), _lib.ptr(self.long_row), _lib.ptr(self.long_slot)))
            self.lanes_per_row, self.row_thresh, self.seg_len, self.seg_mode, self.row_mix, self.nwseg = st.lanes_per_row, st.row_thresh, 32, st.seg_mode, st.row_mix, st.nwseg
        finally:
            L.tgcn_sched_destroy(h)
    self.struct = _lib.SchedStruct(self.lanes_per_row, self.row_thresh, self.nblk, self.nseg, self.nlong, self.nhuge, self.npartial, self.seg_mode,
                                   self.row_mix, self.nwseg, self.blk_row.data_ptr(), self.seg_row.data_ptr(), self.seg_e0.data_ptr(), self.seg_e1.data_ptr(),
                                   self.seg_slot.data_ptr(), self.long_row.data_ptr(), self.long_slot.data_ptr())


Schedule._build_library = _sched_build_library


def _spread_hot_block(perm, n_live):
    """Hub-first order without the hub pile-up: the first H = 2^floor(log2(n_live / 8)) positions of a degree-sorted order (the hot block: on the
    10 M / 160 M R-MAT 524,288 vertices that take 84 % of the gathers) are visited in BIT-REVERSED order, the rest stays sorted.  A plain
    degree-sorted order makes the column-ordered sweep of hop_kernel walk the columns in order of popularity, so that every workgroup in
    flight asks for the same few hub lines at the same time (docs/EXPERIMENTS.md A.4: same L2 hit rate as random labels, 4x the L2 tag stalls,
    +19 % fabric read latency, 3.96 vs 3.69 ms per launch); bit reversal keeps the hot rows together in one block (cache-resident) but makes
    neighbours in popularity neither neighbours in memory nor in the sweep."""
    if n_live < 16:
        return perm
    bits = (n_live // 8).bit_length() - 1
    if bits < 1:
        return perm
    H = 1 << bits
    p = torch.arange(H, device=perm.device, dtype=torch.int64)
    r = torch.zeros_like(p)
    for i in range(bits):
        r |= ((p >> i) & 1) << (bits - 1 - i)
    out = perm.clone()
    out[:H] = perm[r]
    return out


class CompactPlan:
    """Operand with structurally empty rows (R-MAT: 5.27 M of 10 M vertices; the isolated fake vertices the reference's
    coarsening pads with, gcn/coarsening.py:167-217) prepared for tgcn_cheb_forward_compact_f32: hop tensors exist only for
    the n_c vertices that have stored entries.  `first` = the n_c non-empty rows with columns in the caller's labels (hop 1
    gathers from x), `rest` = the same rows and entry order with columns in compact ids -- an entry whose column is an
    empty vertex points at the zero row n_c -- for hops 2..K-1; `rows` / `empty` = caller's label of every compact / empty
    row, ascending.  Both operands share one schedule (same row pointers, same entry order)."""

    def __init__(self, op, keep, kind="rows"):
        dev = op.device
        self.device = dev
        self.kind = kind                            # "rows": keep = rows with entries; "closed": also every vertex an entry points at
        self.n = op.n
        rows = keep.nonzero().flatten()
        self.n_c = int(rows.numel())
        self.rows = _as_i32(rows)
        self.empty = _as_i32((~keep).nonzero().flatten())
        self.n_empty = int(self.empty.numel())
        rowptr_c = torch.cat([op.rowptr[:-1][keep], op.rowptr[-1:]]).contiguous()
        cid = torch.full((op.n,), self.n_c, dtype=torch.int32, device=dev)
        cid[rows] = torch.arange(self.n_c, dtype=torch.int32, device=dev)
        edges_c = op.edges.clone()
        if op.nnz:
            edges_c[:, 0] = cid[op.edges[:, 0].long()]
        self.cid = cid                              # compact id of every vertex, n_c for the empty ones
        self.first = GraphOperand._from_packed(self.n_c, rowptr_c, op.edges, op.nnz, n_cols=op.n)
        self.rest = GraphOperand._from_packed(self.n_c, rowptr_c, edges_c, op.nnz, n_cols=self.n_c + 1)
        self.q_chunk_cache = {}                     # functional.cheb_forward_compact: time steps per pass, chosen once per shape
        self.first._sched = self.rest._sched       # one schedule: built from `rest`, valid for both
        self.first._lock = self.rest._lock

    def schedule_for(self, C_row, aligned16=True):
        return self.rest.schedule_for(C_row, aligned16)

    def rows64(self):
        """`rows` as int64 (the row-packing kernel's index type), made once"""
        r = getattr(self, "_rows64", None)
        if r is None:
            r = self._rows64 = self.rows.long()
        return r


class GraphOperand:
    """L-hat (n x n) in CSR with 8-byte packed {int32 col, float val} entries, on one device."""

    def __init__(self, n, rowptr, col, val, n_cols=None):
        assert rowptr.dtype == torch.int32 and col.dtype == torch.int32 and val.dtype == torch.float32
        nnz = int(col.numel())
        edges = torch.empty((max(nnz, 1), 2), dtype=torch.int32, device=rowptr.device)
        if nnz:
            edges[:, 0] = col
            edges[:, 1] = val.view(torch.int32)
        self._init_packed(n, rowptr, edges, nnz, n_cols)

    @staticmethod
    def _from_packed(n, rowptr, edges, nnz, n_cols=None):
        """Operand over an existing packed entry array (shared, not copied)."""
        op = GraphOperand.__new__(GraphOperand)
        op._init_packed(n, rowptr, edges, nnz, n_cols)
        return op

    def _init_packed(self, n, rowptr, edges, nnz, n_cols):
        assert rowptr.dtype == torch.int32 and edges.dtype == torch.int32 and edges.shape[1] == 2
        self.n = int(n)
        self.n_cols = int(n if n_cols is None else n_cols)   # > n for a vertex shard: owned rows x (owned + halo) columns
        self.nnz = int(nnz)
        if self.nnz >= 2 ** 31 - 1 or self.n >= 2 ** 31 - 1:
            raise _lib.TgcnError("graph operand outside the int32 index range (n=%d nnz=%d)" % (self.n, self.nnz))
        self.device = rowptr.device
        self.rowptr = rowptr.contiguous()
        self.edges = edges
        # small dense operands (the 148-parcel DTI graph of load/res): a dense copy for the matrix-pipe kernels
        self.dense = None
        if self.n_cols == self.n and 16 <= self.n <= DENSE_MAX_N and self.nnz * 4 >= self.n * self.n:
            counts = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
            rows = torch.repeat_interleave(torch.arange(self.n, device=self.device), counts)
            col, val = edges[: self.nnz, 0].to(torch.int64), edges[: self.nnz, 1].contiguous().view(torch.float32)
            self.dense = torch.sparse_coo_tensor(torch.stack([rows, col]), val, (self.n, self.n)).coalesce().to_dense().contiguous()
        self.struct = _lib.CsrStruct(self.n, self.nnz, self.rowptr.data_ptr(), self.edges.data_ptr(),
                                     self.dense.data_ptr() if self.dense is not None else None)
        self._sched = {}
        self._lock = threading.RLock()
        self._transpose = None
        self.values_epoch = 0     # bumped by update_values: which packing of the values a recorded backward belongs to
        self._compact = False     # CompactPlan, None when the operand does not qualify, False until asked
        self.perm = None          # reordered(): internal row i holds the caller's vertex perm[i]
        self.inv_perm = None

    def compact_plan(self, kind="rows"):
        """CompactPlan when enough vertices can be left out of the hop tensors for that to pay, else None (built once per kind).
        kind "rows" (reference_power recursion): the kept vertices are the rows with stored entries -- P_k[i] = 0, k >= 1, for the others, and
        an entry pointing at one of them reads a zero row.  kind "closed" (true Chebyshev recursion, where T_k of an empty row is +-x, not 0):
        additionally every vertex some entry points at, so that the left-out vertices are ISOLATED (no entries, never referenced) and
        T_k[i] = x[i], 0, -x[i], 0, ... holds for them in closed form.  Symmetric patterns give the same set for both: one plan is shared."""
        assert kind in ("rows", "closed")
        with self._lock:
            if self._compact is False:
                self._compact = {}
            if kind not in self._compact:
                plan = None
                if self.n == self.n_cols and self.n >= COMPACT_MIN_ROWS and self.nnz > 0:
                    keep = self.rowptr[1:] > self.rowptr[:-1]
                    shared = False
                    if kind == "closed":
                        ref = torch.zeros(self.n, dtype=torch.bool, device=self.device)
                        ref[self.edges[: self.nnz, 0].long()] = True
                        if bool((ref & ~keep).any().item()):
                            keep = keep | ref
                        else:                                 # same vertex set: the "rows" plan serves both recursions
                            plan, shared = self.compact_plan("rows"), True
                    if not shared:
                        n_c = int(keep.sum().item())
                        if 0 < n_c and (self.n - n_c) >= COMPACT_MIN_EMPTY * self.n:
                            plan = CompactPlan(self, keep, kind)
                self._compact[kind] = plan
            return self._compact[kind]

    def update_values(self, vals, _from_transpose=False):
        """New VALUES on the same pattern, in place (vals: (nnz,) fp32 in this operand's CSR order): learnable edge weights change every
        optimizer step while edge_index does not, and everything expensive -- the COO -> CSR sort, the schedules, the compact plans' row maps --
        depends on the pattern only.  The packed entries, the dense copy, the compact plans' second entry array and the cached transpose (through
        a once-computed entry map) are refreshed; device pointers do not move, so the ctypes structs stay valid."""
        self.values_epoch += 1          # a backward that was recorded against the previous values notices (functional._values_guard)
        if self.nnz == 0:
            return
        vals = vals.detach().to(device=self.device, dtype=torch.float32).reshape(-1)
        assert vals.numel() == self.nnz
        bits = vals.contiguous().view(torch.int32)
        self.edges[: self.nnz, 1] = bits
        if self.dense is not None:
            counts = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
            rows = torch.repeat_interleave(torch.arange(self.n, device=self.device), counts)
            # (coalesce sums duplicates in a fixed order, like the first build; index_put_(accumulate=True) would use float atomics)
            self.dense.copy_(torch.sparse_coo_tensor(torch.stack([rows, self.edges[: self.nnz, 0].to(torch.int64)]), vals, (self.n, self.n)).coalesce().to_dense())
        with self._lock:
            plans = [pl for pl in (self._compact or {}).values() if pl is not None] if self._compact is not False else []
            for pl in plans:
                if pl.rest.edges.data_ptr() != self.edges.data_ptr():
                    pl.rest.edges[: self.nnz, 1] = bits
            T = self._transpose
            if T is not None and not _from_transpose:
                order = getattr(self, "_t_order", None)
                if order is None:        # entry j of the transpose is entry order[j] of this operand: the builder's stable (row', col') = (col, row) sort
                    row, col, _ = self.coo()
                    order = self._t_order = torch.argsort(col * max(self.n, self.n_cols) + row, stable=True)
                T.update_values(vals[order], _from_transpose=True)

    def __deepcopy__(self, memo):
        """An operand is immutable once built (device arrays + ctypes structs that point into them): copies of a module share it."""
        return self

    # ------------------------------------------------------------------ vertex reordering
    def reordered(self, kind):
        """The same operator with its vertices relabelled -- P L P^T plus the permutation, which the layer functions apply to x / bias on
        the way in and to the result on the way out (functional.cheb_layer), so callers keep their own labels.  Square operands only.
          "rcm"        reverse Cuthill-McKee of the symmetrised pattern (scipy, on the host, one-off): the bandwidth-reducing order the
                       reference's coarsened graphs get from their construction (gcn/coarsening.py:167-217 orders vertices by cluster).
                       THE order to use for graphs that have locality under some labelling (meshes, banded, k-NN grids with scrambled ids).
          "hub_first"  decreasing number of stored entries, the hot block visited in bit-reversed order (_spread_hot_block).  NOT a speed-up on
                       graphs without locality: on the 10 M / 160 M R-MAT the hop takes 3.99 ms per launch against 3.73 ms with the caller's own
                       (random) labels -- hop_kernel's column-ordered sweep already has the hub locality, and ANY hub-first order makes every
                       workgroup in flight ask for the same hot block at the same time (profiles/r05_exp_reorder_degree.log; plain degree
                       order: 4.18 ms; docs/EXPERIMENTS.md A.4).  Kept for experiments; warns.  ("degree": old name, same thing, warns too.)
          "degree_sorted"  plain decreasing-degree order (round 1-4's "degree"): the slowest of the three on power-law graphs; warns."""
        assert self.n == self.n_cols, "reordered(): square operands only"
        row, col, val = self.coo()
        if kind in ("hub_first", "degree", "degree_sorted"):
            import warnings
            warnings.warn("GraphOperand.reordered(%r): a hub-first vertex order is measured 7-12 %% SLOWER than the caller's own labels on power-law graphs "
                          "without locality (R-MAT 10 M / 160 M: 3.99 / 4.18 ms per hop launch against 3.73 ms; profiles/r05_exp_reorder_degree.log) -- "
                          "use reordered('rcm') for graphs that have locality, or no reordering%s"
                          % (kind, "; 'degree' was renamed 'hub_first'" if kind == "degree" else ""), stacklevel=2)
            deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
            perm = torch.argsort(deg, descending=True, stable=True)
            if kind != "degree_sorted":
                perm = _spread_hot_block(perm, int((deg > 0).sum().item()))
        elif kind == "rcm":
            import scipy.sparse as sp
            from scipy.sparse.csgraph import reverse_cuthill_mckee
            r, c = row.cpu().numpy(), col.cpu().numpy()
            pat = sp.coo_matrix((torch.ones(r.shape[0]).numpy(), (r, c)), shape=(self.n, self.n)).tocsr()
            perm = torch.as_tensor(reverse_cuthill_mckee((pat + pat.T).tocsr(), symmetric_mode=True).astype("int64"), device=self.device)
        else:
            raise _lib.TgcnError("reordered(): unknown order %r (rcm | hub_first | degree_sorted)" % (kind,))
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(self.n, device=self.device)
        op = GraphOperand.from_coo(self.n, inv[row], inv[col], val, self.device)
        op.perm, op.inv_perm = perm, inv
        return op

    @property
    def shape(self):
        return (self.n, self.n_cols)

    # ------------------------------------------------------------------ constructors
    @staticmethod
    def from_coo(n, row, col, val, device=None, n_cols=None):
        """Entries may come in any order; duplicates stay separate entries (their sum is what
        scatter_add computes, tgcn/nn/gcn.py:308,343)."""
        device = row.device if device is None else torch.device(device)
        row = row.to(device=device, dtype=torch.int64)
        col = col.to(device=device, dtype=torch.int64)
        val = val.to(device=device, dtype=torch.float32)
        if not (BUILDER == "library" and device.type == "cuda"):         # the library's builder checks the range itself
            _check_range(row, col, n, n if n_cols is None else n_cols)
        if not (row.numel() == col.numel() == val.numel()):
            raise _lib.TgcnError("graph operand: row / col / val lengths differ")
        if BUILDER == "library" and device.type == "cuda":
            return GraphOperand._from_coo_library(n, row.contiguous(), col.contiguous(), val.contiguous(), device, n if n_cols is None else n_cols)
        # stable: duplicates of one (row, col) keep their given order, so two operands built from the same list sum them alike
        order = torch.argsort(row * (n if n_cols is None else max(n, n_cols)) + col, stable=True)
        counts = torch.bincount(row, minlength=n)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        return GraphOperand(n, _as_i32(rowptr), _as_i32(col[order]), val[order].contiguous(), n_cols)

    @staticmethod
    def _from_coo_library(n, row, col, val, device, n_cols):
        """COO -> CSR by the HIP kernels of the library (tgcn_csr_build_f32: two stable radix sorts + a prefix sum) into torch-owned arrays"""
        L = _lib.lib()
        nnz = int(row.numel())
        if nnz >= 2 ** 31 - 1 or n >= 2 ** 31 - 1 or n_cols >= 2 ** 31 - 1:
            raise _lib.TgcnError("graph operand outside the int32 index range (n=%d nnz=%d)" % (n, nnz))
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=device)
        edges = torch.empty((max(nnz, 1), 2), dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            ws = torch.empty(L.tgcn_csr_build_workspace_bytes(n, nnz), dtype=torch.uint8, device=device)
            _lib.check(L.tgcn_csr_build_f32(_lib.stream_ptr(), n, n_cols, nnz, _lib.ptr(row), _lib.ptr(col), _lib.ptr(val), _lib.ptr(rowptr),
                                            _lib.ptr(edges), _lib.ptr(ws), ws.numel()))
        return GraphOperand._from_packed(n, rowptr, edges, nnz, n_cols)

    @staticmethod
    def from_dense(L, device=None):
        """Dense (n, n) tensor / ndarray as passed to TGCNCheb / TGCNCheb_H / GCNCheb (gcn.py:18,92,168)."""
        L = torch.as_tensor(L)
        device = L.device if device is None else torch.device(device)
        L = L.to(device=device, dtype=torch.float32)
        idx = L.nonzero()
        return GraphOperand.from_coo(L.shape[0], idx[:, 0], idx[:, 1], L[idx[:, 0], idx[:, 1]], device)

    @staticmethod
    def from_any(L, device):
        """Dense tensor, torch sparse COO/CSR (gcn_matmul accepts those, gcn_matmul.py:154), scipy sparse or ndarray."""
        if isinstance(L, GraphOperand):
            return L if L.device == torch.device(device) else L.to(device)
        if isinstance(L, torch.Tensor):
            if L.layout == torch.sparse_coo:
                Lc = L.coalesce()
                i = Lc.indices()
                return GraphOperand.from_coo(L.shape[0], i[0], i[1], Lc.values(), device)
            if L.layout == torch.sparse_csr:
                Lc = L.to_sparse_coo().coalesce()
                i = Lc.indices()
                return GraphOperand.from_coo(L.shape[0], i[0], i[1], Lc.values(), device)
            return GraphOperand.from_dense(L, device)
        if hasattr(L, "tocoo"):                                        # scipy.sparse
            coo = L.tocoo()
            return GraphOperand.from_coo(coo.shape[0], torch.as_tensor(coo.row), torch.as_tensor(coo.col),
                                         torch.as_tensor(coo.data), device)
        return GraphOperand.from_dense(torch.as_tensor(L), device)

    @staticmethod
    def from_edge_index(edge_index, edge_weight, n, device=None):
        """ChebConv / ChebTimeConv operand (tgcn/nn/gcn.py:398-413, :495-510): self loops removed,
        deg = number of edges per SOURCE vertex (unweighted), lap_e = -deg^-1/2[row] * w_e * deg^-1/2[col],
        deg^-1/2 = 0 for isolated vertices."""
        device = edge_index.device if device is None else torch.device(device)
        if edge_weight is not None and edge_weight.requires_grad:
            raise _lib.TgcnError("edge_weight.requires_grad: a GraphOperand is packed outside autograd, so the weight would silently get no gradient "
                                 "from it -- pass the weight to ChebConv / ChebTimeConv instead (their forward carries d loss / d edge_weight, "
                                 "lap = -deg[row] * edge_weight * deg[col], tgcn/nn/gcn.py:413,510), or detach() it here")
        if BUILDER == "library" and device.type == "cuda":
            return GraphOperand._from_edge_index_library(edge_index, edge_weight, int(n), device)
        # cross-check form (and CPU rehearsals): the same steps with torch index ops
        row, col = edge_index[0].to(device), edge_index[1].to(device)
        _check_range(row, col, n, n)
        keep = row != col
        row, col = row[keep], col[keep]
        if edge_weight is None:
            w = torch.ones(row.numel(), dtype=torch.float32, device=device)
        else:
            w = edge_weight.detach().reshape(-1).to(device=device, dtype=torch.float32)[keep]
        deg = torch.bincount(row, minlength=n).to(torch.float32)
        dis = deg.pow(-0.5)
        dis[torch.isinf(dis)] = 0
        lap = -dis[row] * w * dis[col]
        return GraphOperand.from_coo(n, row, col, lap, device)

    @staticmethod
    def _from_edge_index_library(edge_index, edge_weight, n, device):
        """tgcn_edge_normalise_f32 (keep mask, order-preserving compaction, integer source degrees, -d^-1/2 w d^-1/2) on torch-owned arrays,
        then the library's COO -> CSR: the product path runs no arithmetic of its own on the edge list"""
        L = _lib.lib()
        ei = edge_index.to(device=device, dtype=torch.int64).contiguous()
        if ei.dim() != 2 or ei.shape[0] != 2:
            raise _lib.TgcnError("edge_index must be (2, E)")
        E = int(ei.shape[1])
        w = None if edge_weight is None else edge_weight.detach().reshape(-1).to(device=device, dtype=torch.float32).contiguous()
        if w is not None and w.numel() != E:
            raise _lib.TgcnError("edge_weight has %d entries for %d edges" % (w.numel(), E))
        row = torch.empty(max(E, 1), dtype=torch.int64, device=device)
        col = torch.empty(max(E, 1), dtype=torch.int64, device=device)
        val = torch.empty(max(E, 1), dtype=torch.float32, device=device)
        kept = C.c_int64(0)
        with torch.cuda.device(device):
            ws = torch.empty(max(L.tgcn_edge_normalise_workspace_bytes(n, E), 256), dtype=torch.uint8, device=device)
            _lib.check(L.tgcn_edge_normalise_f32(_lib.stream_ptr(), n, E, _lib.ptr(ei), _lib.ptr(w), _lib.ptr(row), _lib.ptr(col), _lib.ptr(val),
                                                 C.byref(kept), _lib.ptr(ws), ws.numel()))
        k = int(kept.value)
        return GraphOperand._from_coo_library(n, row[:k].contiguous(), col[:k].contiguous(), val[:k].contiguous(), device, n)

    @staticmethod
    def from_adjacency(n, row, col, weight, lmax=2.0, device=None):
        """The operand the callers build on the host before every layer -- rescale_L(laplacian(W, normalized=True),
        lmax) (gcn/graph.py:117-136, 232-238; examples/gcn_mnist.py:131 re-does it every forward) -- from the COO of
        the weight matrix W, on the device:  d = colsum(W) + eps,  L = I - D^-1/2 W D^-1/2,  L-hat = L * (2/lmax) - I.
        With lmax = 2 the diagonal cancels and L-hat = -D^-1/2 W D^-1/2."""
        device = row.device if device is None else torch.device(device)
        row = row.to(device=device, dtype=torch.int64).contiguous()
        col = col.to(device=device, dtype=torch.int64).contiguous()
        w = weight.detach().to(device=device, dtype=torch.float32).contiguous()
        if not (row.numel() == col.numel() == w.numel()):
            raise _lib.TgcnError("adjacency: row / col / weight lengths differ")
        if BUILDER == "library" and device.type == "cuda":
            # tgcn_adjacency_normalise_f32: column sums by a stable sort + one wave per column (fixed order, no float atomics), values, diagonal
            L = _lib.lib()
            m = int(row.numel())
            ro = torch.empty(m + n, dtype=torch.int64, device=device)
            co = torch.empty(m + n, dtype=torch.int64, device=device)
            vo = torch.empty(m + n, dtype=torch.float32, device=device)
            count = C.c_int64(0)
            with torch.cuda.device(device):
                ws = torch.empty(max(L.tgcn_adjacency_normalise_workspace_bytes(n, m), 256), dtype=torch.uint8, device=device)
                _lib.check(L.tgcn_adjacency_normalise_f32(_lib.stream_ptr(), n, m, _lib.ptr(row), _lib.ptr(col), _lib.ptr(w), float(lmax), _lib.ptr(ro),
                                                          _lib.ptr(co), _lib.ptr(vo), C.byref(count), _lib.ptr(ws), ws.numel()))
            k = int(count.value)
            return GraphOperand._from_coo_library(n, ro[:k].contiguous(), co[:k].contiguous(), vo[:k].contiguous(), device, n)
        # cross-check form (and CPU rehearsals): the same steps with torch index ops
        _check_range(row, col, n, n)
        d = torch.zeros(n, dtype=torch.float32, device=device).index_add_(0, col, w)      # W.sum(axis=0)
        d = d + 1.401298464324817e-45                                                      # np.spacing(float32(0))
        dis = 1.0 / torch.sqrt(d)
        scale = 2.0 / float(lmax)
        val = -scale * dis[row] * w * dis[col]
        if scale != 1.0:                                                                   # (2/lmax - 1) on the diagonal
            diag = torch.arange(n, device=device)
            row = torch.cat([row, diag])
            col = torch.cat([col, diag])
            val = torch.cat([val, torch.full((n,), scale - 1.0, dtype=torch.float32, device=device)])
        return GraphOperand.from_coo(n, row, col, val, device)

    # ------------------------------------------------------------------ derived operands
    def coo(self):
        counts = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
        row = torch.repeat_interleave(torch.arange(self.n, device=self.device), counts)
        e = self.edges[: self.nnz]
        return row, e[:, 0].to(torch.int64), e[:, 1].contiguous().view(torch.float32)

    def transpose(self):
        """L-hat^T, for the input gradient (L-hat is symmetric for the dense-L classes' usual operand, but
        ChebConv normalises by the source degree only, so the general case is kept)."""
        with self._lock:
            return self._transpose_locked()

    def _transpose_locked(self):
        if self._transpose is None:
            row, col, val = self.coo()
            # a rectangular operand (vertex shard: n owned rows x n_cols owned + halo columns) transposes to n_cols x n
            self._transpose = GraphOperand.from_coo(self.n_cols, col, row, val, self.device, n_cols=self.n)
            self._transpose._transpose = self
        return self._transpose

    def to(self, device):
        row, col, val = self.coo()
        op = GraphOperand.from_coo(self.n, row, col, val, device, n_cols=self.n_cols)
        if self.perm is not None:           # a reordered operand stays one on the other device
            op.perm, op.inv_perm = self.perm.to(device), self.inv_perm.to(device)
        return op

    def schedule(self, lanes_per_row):
        with self._lock:                  # replicas of one module may share an operand (nn.DataParallel threads)
            s = self._sched.get(lanes_per_row)
            if s is None:
                s = self._sched[lanes_per_row] = Schedule(self.rowptr, self.n, lanes_per_row, edges=self.edges, n_cols=self.n_cols)
        return s

    def schedule_for(self, C_row, aligned16=True):
        return self.schedule(_lib.lib().tgcn_hop_lanes_per_row(int(C_row), 1 if aligned16 else 0))

    def to_scipy(self):
        import scipy.sparse as sp
        row, col, val = self.coo()
        return sp.coo_matrix((val.cpu().numpy(), (row.cpu().numpy(), col.cpu().numpy())), shape=(self.n, self.n)).tocsr()
