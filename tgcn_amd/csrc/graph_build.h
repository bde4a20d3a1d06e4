// graph_build.h -- construction of the sparse operand and its work schedule inside the library, so that the C ABI is usable
// without the Python package (include/tgcn_hip.h: tgcn_graph_*, tgcn_sched_*, tgcn_csr_build_f32).  The work is done ON THE DEVICE
// by the kernels of device_build.h (stable radix sort, prefix sums, binary-search marks); this file is their host-side
// orchestration.  tgcn_graph_create_* / tgcn_sched_build* are the only entry points that allocate device memory and synchronise
// (a handful of 8-byte read-backs: totals that size the next allocation); they run once per operand, never on the forward path.
// The schedule is the one tgcn_amd/graph.py::Schedule builds with torch index ops (row blocks + column-ordered segments), which
// stays as the cross-check: tests/test_c_abi_graph.py and tests/test_device_build.py compare the two array by array.
// Part of the single translation unit tgcn_hip.hip (included once, inside its anonymous namespace).
#pragma once

struct DeviceBuf {
  void* p = nullptr;
  ~DeviceBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) {
    if (p) { (void)hipFree(p); p = nullptr; }
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { p = nullptr; return -1; }
    return 0;
  }
  int zero(size_t bytes) { return (alloc(bytes) || hipMemset(p, 0, bytes ? bytes : 16) != hipSuccess) ? -1 : 0; }
};

}  // namespace (the opaque handle types are named by the header)

struct tgcn_graph {
  tgcn_csr csr;
  int64_t n_cols;
  DeviceBuf rowptr, edges;
};

struct tgcn_sched {
  tgcn_csr_sched s;
  DeviceBuf blk_row, seg_row, seg_e0, seg_e1, seg_slot, long_row, long_slot;
};

namespace {

template <typename T>
int read_back(T* dst, const T* dev) {      // one value of a device array (sizes the next step)
  return hipMemcpy(dst, dev, sizeof(T), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

inline bool graph_sizes_ok(int64_t n, int64_t n_cols, int64_t nnz) {
  return n > 0 && n_cols > 0 && nnz >= 0 && n < (int64_t)INT32_MAX && n_cols < (int64_t)INT32_MAX && nnz < (int64_t)INT32_MAX - 1;
}

// rows sorted by (row, col), duplicates kept as separate entries in their given order (their sum is what scatter_add computes,
// tgcn/nn/gcn.py:308,343): csr_build_device into arrays the handle owns
int graph_from_device_coo(int64_t n, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col, const float* val, tgcn_graph** out) {
  tgcn_graph* g = new (std::nothrow) tgcn_graph();
  if (!g) TGCN_FAIL(TGCN_ERR_LAUNCH, "graph: out of host memory");
  DeviceBuf ws;
  if (g->rowptr.alloc((size_t)(n + 1) * sizeof(int32_t)) || g->edges.alloc((size_t)(nnz ? nnz : 1) * sizeof(tgcn_edge)) ||
      ws.alloc(csr_build_ws_bytes(n, nnz))) {
    delete g;
    TGCN_FAIL(TGCN_ERR_LAUNCH, "graph: device allocation failed");
  }
  const int rc = csr_build_device((hipStream_t)0, n, n_cols, nnz, row, col, val, (int32_t*)g->rowptr.p, (tgcn_edge*)g->edges.p, (char*)ws.p);
  if (rc != TGCN_OK || hipStreamSynchronize((hipStream_t)0) != hipSuccess) {
    delete g;
    if (rc == TGCN_OK) TGCN_FAIL(TGCN_ERR_LAUNCH, "graph: build failed on the device");
    return rc;
  }
  g->csr.n = n; g->csr.nnz = nnz; g->csr.rowptr = (const int32_t*)g->rowptr.p; g->csr.edges = (const tgcn_edge*)g->edges.p;
  g->csr.dense = nullptr;
  g->n_cols = n_cols;
  *out = g;
  return TGCN_OK;
}

// rows[e] = row of entry e of a CSR given by int64 row pointers (last rp[i] <= e)
__global__ void expand_rows_kernel(const int64_t* __restrict__ rp, int64_t n, int64_t nnz, int64_t* __restrict__ rows, int* __restrict__ bad) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n;                         // first i with rp[i + 1] > e
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (rp[mid + 1] <= e) lo = mid + 1; else hi = mid; }
    if (lo >= n) { *bad = 1; lo = n - 1; }
    rows[e] = lo;
  }
}

__global__ void rowptr_check_kernel(const int64_t* __restrict__ rp, int64_t n, int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (rp[i + 1] < rp[i] || (i == 0 && rp[0] != 0)) *bad = 1;
}

__global__ void i32_to_i64_kernel(const int32_t* __restrict__ in, int64_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i];
}

}  // namespace

extern "C" {

size_t tgcn_csr_build_workspace_bytes(int64_t n, int64_t nnz) { return (n > 0 && nnz >= 0) ? csr_build_ws_bytes(n, nnz) : 0; }

int tgcn_csr_build_f32(void* stream, int64_t n, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col, const float* val,
                       int32_t* rowptr, tgcn_edge* edges, void* workspace, size_t workspace_bytes) {
  if (!graph_sizes_ok(n, n_cols, nnz) || !rowptr || (nnz > 0 && (!row || !col || !val || !edges))) TGCN_FAIL(TGCN_ERR_INVALID, "csr_build: bad argument");
  if (!workspace || workspace_bytes < csr_build_ws_bytes(n, nnz) || ((uintptr_t)workspace & 255)) TGCN_FAIL(TGCN_ERR_WORKSPACE, "csr_build: workspace %zu < %zu (256-byte aligned)", workspace_bytes, csr_build_ws_bytes(n, nnz));
  return csr_build_device((hipStream_t)stream, n, n_cols, nnz, row, col, val, rowptr, edges, (char*)workspace);
}

int tgcn_graph_create_from_coo(int64_t n, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col, const float* val,
                               tgcn_graph** out) {
  if (!out || !graph_sizes_ok(n, n_cols, nnz) || (nnz > 0 && (!row || !col || !val))) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_coo: bad argument");
  return graph_from_device_coo(n, n_cols, nnz, row, col, val, out);
}

int tgcn_graph_create_from_csr(int64_t n, int64_t n_cols, const int64_t* rowptr, const int32_t* col, const float* val, tgcn_graph** out) {
  if (!out || !rowptr || n <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_csr: bad argument");
  int64_t nnz = 0, first = 0;
  if (read_back(&nnz, rowptr + n) || read_back(&first, rowptr)) TGCN_FAIL(TGCN_ERR_LAUNCH, "graph_create_from_csr: device read failed");
  if (!graph_sizes_ok(n, n_cols, nnz) || first != 0 || (nnz > 0 && (!col || !val))) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_csr: bad rowptr");
  DeviceBuf rows, cols, bad;
  if (rows.alloc((size_t)(nnz ? nnz : 1) * 8) || cols.alloc((size_t)(nnz ? nnz : 1) * 8) || bad.zero(sizeof(int))) TGCN_FAIL(TGCN_ERR_LAUNCH, "graph_create_from_csr: device allocation failed");
  hipLaunchKernelGGL(rowptr_check_kernel, dim3(grid_1d(n)), dim3(kBlock), 0, (hipStream_t)0, rowptr, n, (int*)bad.p);
  if (nnz > 0) {
    hipLaunchKernelGGL(expand_rows_kernel, dim3(grid_1d(nnz)), dim3(kBlock), 0, (hipStream_t)0, rowptr, n, nnz, (int64_t*)rows.p, (int*)bad.p);
    hipLaunchKernelGGL(i32_to_i64_kernel, dim3(grid_1d(nnz)), dim3(kBlock), 0, (hipStream_t)0, col, (int64_t*)cols.p, nnz);
  }
  int h_bad = 0;
  if (read_back(&h_bad, (const int*)bad.p)) TGCN_FAIL(TGCN_ERR_LAUNCH, "graph_create_from_csr: device read failed");
  if (h_bad) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_csr: rowptr is not a non-decreasing sequence from 0 to nnz");
  return graph_from_device_coo(n, n_cols, nnz, (const int64_t*)rows.p, (const int64_t*)cols.p, val, out);
}

/* ChebConv / ChebTimeConv operand from the caller's edge list (tgcn/nn/gcn.py:398-413 == :495-510): self loops removed,
 * deg = number of edges per SOURCE vertex (unweighted), lap_e = -deg^-1/2[row] * w_e * deg^-1/2[col], deg^-1/2 = 0 for
 * vertices without outgoing edges. */
int tgcn_graph_create_from_edge_index(int64_t n, int64_t E, const int64_t* edge_index, const float* edge_weight, tgcn_graph** out) {
  if (!out || !graph_sizes_ok(n, n, E) || (E > 0 && !edge_index)) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_edge_index: bad argument");
  DeviceBuf ws, r, c, v;
  const size_t m1 = (size_t)(E ? E : 1);
  if (ws.alloc(edge_norm_ws_bytes(n, E)) || r.alloc(m1 * 8) || c.alloc(m1 * 8) || v.alloc(m1 * 4))
    TGCN_FAIL(TGCN_ERR_LAUNCH, "graph_create_from_edge_index: device allocation failed");
  int64_t kept = 0;
  const int rc = edge_normalise_device((hipStream_t)0, n, E, edge_index, edge_weight, (int64_t*)r.p, (int64_t*)c.p, (float*)v.p, &kept, (char*)ws.p);
  if (rc != TGCN_OK) return rc;
  return graph_from_device_coo(n, n, kept, (const int64_t*)r.p, (const int64_t*)c.p, (const float*)v.p, out);
}

/* The same normalisation on CALLER memory (ABI v5): the host side that owns its arrays (torch tensors: tgcn_amd/graph.py::from_edge_index)
 * runs the library's kernels instead of its own arithmetic.  row / col / val: E slots each; *kept (host) = entries written, in the
 * given order.  Synchronises the stream once (range flag + count). */
size_t tgcn_edge_normalise_workspace_bytes(int64_t n, int64_t E) { return (n > 0 && E >= 0) ? edge_norm_ws_bytes(n, E) : 0; }

int tgcn_edge_normalise_f32(void* stream, int64_t n, int64_t E, const int64_t* edge_index, const float* edge_weight, int64_t* row, int64_t* col,
                            float* val, int64_t* kept, void* workspace, size_t workspace_bytes) {
  if (!kept || !graph_sizes_ok(n, n, E) || (E > 0 && (!edge_index || !row || !col || !val))) TGCN_FAIL(TGCN_ERR_INVALID, "edge_normalise: bad argument");
  if (!workspace || workspace_bytes < edge_norm_ws_bytes(n, E) || ((uintptr_t)workspace & 255))
    TGCN_FAIL(TGCN_ERR_WORKSPACE, "edge_normalise: workspace %zu < %zu (256-byte aligned)", workspace_bytes, edge_norm_ws_bytes(n, E));
  return edge_normalise_device((hipStream_t)stream, n, E, edge_index, edge_weight, row, col, val, kept, (char*)workspace);
}

/* COO of the weight matrix W -> COO of rescale_L(laplacian(W, normalized=True), lmax) (gcn/graph.py:117-136, 232-238) on caller memory:
 * d = colsum(W) + eps (one wave per column, fixed summation order), L-hat = (2/lmax) (I - D^-1/2 W D^-1/2) - I.  row_out / col_out /
 * val_out: m + n slots (the n diagonal entries exist only for lmax != 2); *count (host) = entries written. */
size_t tgcn_adjacency_normalise_workspace_bytes(int64_t n, int64_t m) { return (n > 0 && m >= 0) ? adjacency_norm_ws_bytes(n, m) : 0; }

int tgcn_adjacency_normalise_f32(void* stream, int64_t n, int64_t m, const int64_t* row, const int64_t* col, const float* weight, float lmax,
                                 int64_t* row_out, int64_t* col_out, float* val_out, int64_t* count, void* workspace, size_t workspace_bytes) {
  if (!count || !graph_sizes_ok(n, n, m + n) || !(lmax > 0.f) || !row_out || !col_out || !val_out || (m > 0 && (!row || !col || !weight)))
    TGCN_FAIL(TGCN_ERR_INVALID, "adjacency_normalise: bad argument");
  if (!workspace || workspace_bytes < adjacency_norm_ws_bytes(n, m) || ((uintptr_t)workspace & 255))
    TGCN_FAIL(TGCN_ERR_WORKSPACE, "adjacency_normalise: workspace %zu < %zu (256-byte aligned)", workspace_bytes, adjacency_norm_ws_bytes(n, m));
  return adjacency_normalise_device((hipStream_t)stream, n, m, row, col, weight, lmax, row_out, col_out, val_out, count, (char*)workspace);
}

const tgcn_csr* tgcn_graph_csr(const tgcn_graph* g) { return g ? &g->csr : nullptr; }
int64_t tgcn_graph_n_cols(const tgcn_graph* g) { return g ? g->n_cols : 0; }
void tgcn_graph_destroy(tgcn_graph* g) { delete g; }

/* Row-block + column-ordered-segment schedule of a CSR operand for rows of C floats (the arrays documented at tgcn_csr_sched),
 * built on the device. */
int tgcn_sched_build_csr(const tgcn_csr* A, int64_t n_cols, int32_t C, int aligned16, tgcn_sched** out) {
  if (!A || !out || C <= 0 || A->n <= 0 || !A->rowptr || (A->nnz > 0 && !A->edges) || n_cols <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "sched_build: bad argument");
  const int lanes = hop_geom(C, aligned16).lpr;
  const int gpb = kBlock / lanes;
  const int64_t n = A->n;
  const int32_t seg_mode = 0;                             // lane-group segments, as tgcn_amd/graph.py::SEG_MODE (wave segments measured slower on cfg5)
  const int32_t row_thresh = 32, seg_len = 32, huge_slots = 64, row_cost = 4;
  const int64_t max_blocks_hint = 2048;
  hipStream_t st = (hipStream_t)0;
  tgcn_sched* sc = new (std::nothrow) tgcn_sched();
  if (!sc) TGCN_FAIL(TGCN_ERR_LAUNCH, "sched_build: out of host memory");
#define TGCN_SCHED_FAIL(...) do { delete sc; TGCN_FAIL(TGCN_ERR_LAUNCH, __VA_ARGS__); } while (0)
  // ---- short rows: nnz-balanced row blocks (cost = entries + 4 per row; rows that become segments cost 4)
  // row_thresh < entries <= wave_max: whole-row wave segments (nwseg).  16-lane groups only: with 4-lane groups a wave is 16 groups and a
  // 40-entry row leaves most of them idle (cfg5n: hop +8 %); other widths unmeasured
  const int32_t wave_max = lanes == 16 ? 32 * (64 / lanes) : row_thresh;
  DeviceBuf cost, is_seg, is_wave, pos, posw, scanws, empties;
  if (empties.zero(8) || cost.alloc((size_t)n * 8) || is_seg.zero((size_t)(n + 1) * 8) || is_wave.zero((size_t)(n + 1) * 8) || pos.alloc((size_t)(n + 1) * 8) ||
      posw.alloc((size_t)(n + 1) * 8) || scanws.alloc(scan_ws_elems(n + 1) * 8))
    TGCN_SCHED_FAIL("sched_build: device allocation failed");
  hipLaunchKernelGGL(sched_cost_kernel, dim3(grid_1d(n)), dim3(kBlock), 0, st, A->rowptr, n, (int)row_thresh, (int)wave_max, (int)row_cost, (int64_t*)cost.p,
                     (int64_t*)is_seg.p, (int64_t*)is_wave.p, (unsigned long long*)empties.p);
  scan_i64(st, (const int64_t*)cost.p, (int64_t*)cost.p, n, 1, (int64_t*)scanws.p);                   // cum = inclusive prefix sum
  scan_i64(st, (const int64_t*)is_seg.p, (int64_t*)pos.p, n + 1, 0, (int64_t*)scanws.p);              // pos[n] = rows cut into lane-group segments
  scan_i64(st, (const int64_t*)is_wave.p, (int64_t*)posw.p, n + 1, 0, (int64_t*)scanws.p);            // posw[n] = whole-row wave segments
  int64_t total = 0, m = 0, mw = 0, n_empty = 0;
  if (read_back(&total, (const int64_t*)cost.p + (n - 1)) || read_back(&m, (const int64_t*)pos.p + n) || read_back(&mw, (const int64_t*)posw.p + n) ||
      read_back(&n_empty, (const int64_t*)empties.p))
    TGCN_SCHED_FAIL("sched_build: device read failed");
  const int64_t cap = lanes <= 16 ? 64 : 256;        // narrow rows: small blocks keep an XCD's gather window inside its L2 (tgcn_amd/graph.py)
  const int64_t target = std::max<int64_t>(gpb * 16, std::min<int64_t>(gpb * cap, (total + max_blocks_hint - 1) / max_blocks_hint));
  const int64_t nblk = std::max<int64_t>(1, (total + target - 1) / target);
  if (sc->blk_row.alloc((size_t)(nblk + 1) * 4)) TGCN_SCHED_FAIL("sched_build: device allocation failed");
  hipLaunchKernelGGL(sched_marks_kernel, dim3(grid_1d(nblk + 1)), dim3(kBlock), 0, st, (const int64_t*)cost.p, n, target, nblk, (int32_t*)sc->blk_row.p);
  // ---- longest rows: lane-group segments; rows with several segments ("long") first, by decreasing segment count (stable)
  int64_t nlong = 0, nhuge = 0, npartial = 0, nsegl = 0;
  DeviceBuf seg_rows, key, sortws, nsegs, first, cnt2;
  if (m > 0) {
    const uint32_t max_key = (uint32_t)((A->nnz + seg_len - 1) / seg_len + 1);
    if (seg_rows.alloc((size_t)m * 4) || key.alloc((size_t)m * 4) || sortws.alloc(sort_ws_bytes(m)) || nsegs.zero((size_t)(m + 1) * 8) ||
        first.alloc((size_t)(m + 1) * 8) || cnt2.alloc(16))
      TGCN_SCHED_FAIL("sched_build: device allocation failed");
    hipLaunchKernelGGL(sched_longrows_kernel, dim3(grid_1d(n)), dim3(kBlock), 0, st, A->rowptr, n, (const int64_t*)is_seg.p, (const int64_t*)pos.p,
                       (int)seg_len, max_key, (uint32_t*)seg_rows.p, (uint32_t*)key.p);
    radix_sort_pairs(st, (uint32_t*)key.p, (uint32_t*)seg_rows.p, m, bits_for((int64_t)max_key + 1), (char*)sortws.p);
    hipLaunchKernelGGL(sched_nsegs_kernel, dim3(grid_1d(m)), dim3(kBlock), 0, st, (const uint32_t*)key.p, max_key, m, (int64_t*)nsegs.p);
    hipLaunchKernelGGL(sched_counts_kernel, dim3(1), dim3(64), 0, st, (const int64_t*)nsegs.p, m, (int)huge_slots, (int64_t*)cnt2.p);
    scan_i64(st, (const int64_t*)nsegs.p, (int64_t*)first.p, m + 1, 0, (int64_t*)scanws.p);             // first[m] = number of segments
    int64_t h2[2] = {0, 0};
    if (hipMemcpy(h2, cnt2.p, 16, hipMemcpyDeviceToHost) != hipSuccess || read_back(&nsegl, (const int64_t*)first.p + m)) TGCN_SCHED_FAIL("sched_build: device read failed");
    nlong = h2[0]; nhuge = h2[1];
    if (nlong > 0 && read_back(&npartial, (const int64_t*)first.p + nlong)) TGCN_SCHED_FAIL("sched_build: device read failed");
  }
  const int64_t nseg = mw + nsegl;               // whole-row wave segments first, then the lane-group segments
  if (nseg >= (int64_t)INT32_MAX) TGCN_SCHED_FAIL("sched_build: too many segments");
  const size_t sball = (size_t)(nseg > 0 ? nseg : 1) * 4;
  if (sc->seg_row.zero(sball) || sc->seg_e0.zero(sball) || sc->seg_e1.zero(sball) || sc->seg_slot.zero(sball) || sc->long_row.zero((size_t)(nlong ? nlong : 1) * 4) ||
      sc->long_slot.zero((size_t)(nlong + 2) * 4))
    TGCN_SCHED_FAIL("sched_build: device allocation failed");
  DeviceBuf wrows, wkey, wsort, u_row, u_e0, u_e1, u_slot, key2, ident, sortws2;
  if (mw > 0) {                                  // in order of their first column (stable: ties in row order)
    if (wrows.alloc((size_t)mw * 4) || wkey.alloc((size_t)mw * 4) || wsort.alloc(sort_ws_bytes(mw))) TGCN_SCHED_FAIL("sched_build: device allocation failed");
    hipLaunchKernelGGL(sched_waverows_kernel, dim3(grid_1d(n)), dim3(kBlock), 0, st, A->rowptr, A->edges, n, (const int64_t*)is_wave.p, (const int64_t*)posw.p,
                       (uint32_t*)wrows.p, (uint32_t*)wkey.p);
    radix_sort_pairs(st, (uint32_t*)wkey.p, (uint32_t*)wrows.p, mw, bits_for(n_cols), (char*)wsort.p);
    hipLaunchKernelGGL(sched_wavesegs_kernel, dim3(grid_1d(mw)), dim3(kBlock), 0, st, A->rowptr, (const uint32_t*)wrows.p, mw, (int32_t*)sc->seg_row.p,
                       (int32_t*)sc->seg_e0.p, (int32_t*)sc->seg_e1.p, (int32_t*)sc->seg_slot.p);
  }
  if (nsegl > 0) {
    const size_t sb = (size_t)nsegl * 4;
    if (u_row.alloc(sb) || u_e0.alloc(sb) || u_e1.alloc(sb) || u_slot.alloc(sb) || key2.alloc(sb) || ident.alloc(sb) || sortws2.alloc(sort_ws_bytes(nsegl)))
      TGCN_SCHED_FAIL("sched_build: device allocation failed");
    hipLaunchKernelGGL(sched_segments_kernel, dim3(grid_1d(nsegl)), dim3(kBlock), 0, st, A->rowptr, A->edges, (const uint32_t*)seg_rows.p, (const int64_t*)first.p, m,
                       nsegl, npartial, (int)seg_len, (int32_t*)u_row.p, (int32_t*)u_e0.p, (int32_t*)u_e1.p, (int32_t*)u_slot.p, (uint32_t*)key2.p, (uint32_t*)ident.p);
    // processing order: by first column (stable)
    radix_sort_pairs(st, (uint32_t*)key2.p, (uint32_t*)ident.p, nsegl, bits_for(n_cols), (char*)sortws2.p);
    const unsigned gs = grid_1d(nsegl);
    hipLaunchKernelGGL(permute_i32_kernel, dim3(gs), dim3(kBlock), 0, st, (const int32_t*)u_row.p, (const uint32_t*)ident.p, (int32_t*)sc->seg_row.p + mw, nsegl);
    hipLaunchKernelGGL(permute_i32_kernel, dim3(gs), dim3(kBlock), 0, st, (const int32_t*)u_e0.p, (const uint32_t*)ident.p, (int32_t*)sc->seg_e0.p + mw, nsegl);
    hipLaunchKernelGGL(permute_i32_kernel, dim3(gs), dim3(kBlock), 0, st, (const int32_t*)u_e1.p, (const uint32_t*)ident.p, (int32_t*)sc->seg_e1.p + mw, nsegl);
    hipLaunchKernelGGL(permute_i32_kernel, dim3(gs), dim3(kBlock), 0, st, (const int32_t*)u_slot.p, (const uint32_t*)ident.p, (int32_t*)sc->seg_slot.p + mw, nsegl);
    if (nlong > 0)
      hipLaunchKernelGGL(sched_long_kernel, dim3(grid_1d(nlong + 1)), dim3(kBlock), 0, st, (const uint32_t*)seg_rows.p, (const int64_t*)first.p, nlong, npartial,
                         (int32_t*)sc->long_row.p, (int32_t*)sc->long_slot.p);
  }
  if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) TGCN_SCHED_FAIL("sched_build: kernels failed");
#undef TGCN_SCHED_FAIL
  memset(&sc->s, 0, sizeof(sc->s));
  sc->s.lanes_per_row = lanes; sc->s.row_thresh = row_thresh; sc->s.nblk = (int32_t)nblk; sc->s.nseg = (int32_t)nseg;
  sc->s.nlong = (int32_t)nlong; sc->s.nhuge = (int32_t)nhuge; sc->s.npartial = (int32_t)npartial; sc->s.seg_mode = seg_mode;
  sc->s.nwseg = (int32_t)mw;
  sc->s.row_mix = (n_empty * 8 >= n) ? 1 : 0;            // many empty rows: their blocks among the segment blocks (tgcn_csr_sched.row_mix)
  sc->s.blk_row = (const int32_t*)sc->blk_row.p; sc->s.seg_row = (const int32_t*)sc->seg_row.p; sc->s.seg_e0 = (const int32_t*)sc->seg_e0.p;
  sc->s.seg_e1 = (const int32_t*)sc->seg_e1.p; sc->s.seg_slot = (const int32_t*)sc->seg_slot.p; sc->s.long_row = (const int32_t*)sc->long_row.p;
  sc->s.long_slot = (const int32_t*)sc->long_slot.p;
  *out = sc;
  return TGCN_OK;
}

int tgcn_sched_build(const tgcn_graph* g, int32_t C, int aligned16, tgcn_sched** out) {
  if (!g) TGCN_FAIL(TGCN_ERR_INVALID, "sched_build: bad argument");
  return tgcn_sched_build_csr(&g->csr, g->n_cols, C, aligned16, out);
}

/* The arrays of a library-built schedule copied (device to device, on `stream`) into arrays the caller owns, sized from the counts of
 * tgcn_sched_get: blk_row [nblk+1], seg_* [max(nseg,1)], long_row [max(nlong,1)], long_slot [nlong+1, at least 2]. */
int tgcn_sched_copy(const tgcn_sched* s, void* stream, int32_t* blk_row, int32_t* seg_row, int32_t* seg_e0, int32_t* seg_e1, int32_t* seg_slot,
                    int32_t* long_row, int32_t* long_slot) {
  if (!s || !blk_row || !seg_row || !seg_e0 || !seg_e1 || !seg_slot || !long_row || !long_slot) TGCN_FAIL(TGCN_ERR_INVALID, "sched_copy: null array");
  hipStream_t st = (hipStream_t)stream;
  const size_t ns = (size_t)(s->s.nseg > 0 ? s->s.nseg : 1) * 4, nl = (size_t)(s->s.nlong > 0 ? s->s.nlong : 1) * 4;
  const size_t nls = (size_t)(s->s.nlong > 0 ? s->s.nlong + 1 : 2) * 4;
  if (hipMemcpyAsync(blk_row, s->blk_row.p, (size_t)(s->s.nblk + 1) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemcpyAsync(seg_row, s->seg_row.p, ns, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemcpyAsync(seg_e0, s->seg_e0.p, ns, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemcpyAsync(seg_e1, s->seg_e1.p, ns, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemcpyAsync(seg_slot, s->seg_slot.p, ns, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemcpyAsync(long_row, s->long_row.p, nl, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemcpyAsync(long_slot, s->long_slot.p, nls, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    TGCN_FAIL(TGCN_ERR_LAUNCH, "sched_copy: copy failed");
  return TGCN_OK;
}

const tgcn_csr_sched* tgcn_sched_get(const tgcn_sched* s) { return s ? &s->s : nullptr; }
void tgcn_sched_destroy(tgcn_sched* s) { delete s; }

}  // extern "C"

namespace {

// ---- Graclus / METIS-style greedy matching of one coarsening level (gcn/coarsening.py:119-165 is a Python loop over vertices and
// their entries).  HOST arrays in, host array out: coarsening is one-off preprocessing of the caller's graph, long before the
// forward path.  Entries (rr, cc, vv) sorted by row (any order inside a row: the FIRST best neighbour in that order wins, as in
// the reference); `order` = the sequence in which vertices are visited; `weight` = the vertex weights (degrees).  An unmatched
// vertex v pairs with its unmatched neighbour u maximising vv * (1/weight[v] + 1/weight[u]), in the arithmetic of T.
// Row extents are derived the way the reference derives them (coarsening.py:132-141), bit for bit, because every coarser graph
// depends on the result: rows are numbered in order of appearance (equal to the vertex id when no vertex is isolated) and, with
// a_k entries in the k-th appearing row, the loop reads a_0 + 1 entries for the first row (one entry of the second row too),
// a_k for the rows in between and a_k - 1 for the last row.
template <typename T>
int graclus_match(int64_t nnz, const int64_t* rr, const int64_t* cc, const T* vv, int64_t n, const int64_t* order, const T* weight,
                  int32_t* cluster) {
  if (n <= 0 || nnz <= 0 || !rr || !cc || !vv || !order || !weight || !cluster) TGCN_FAIL(TGCN_ERR_INVALID, "graclus_match: bad argument");
  if (n != rr[nnz - 1] + 1) TGCN_FAIL(TGCN_ERR_INVALID, "graclus_match: n must be the last row + 1 (coarsening.py:121)");
  std::vector<int64_t> first((size_t)n + 1, 0), extent((size_t)n + 1, 0);
  {
    int64_t seen = 0, current = rr[0];
    for (int64_t e = 0; e < nnz; ++e) {
      if (rr[e] < 0 || rr[e] >= n || cc[e] < 0 || cc[e] >= n || (e > 0 && rr[e] < rr[e - 1])) TGCN_FAIL(TGCN_ERR_INVALID, "graclus_match: entries must be sorted by row, inside [0, n)");
      extent[seen] += 1;
      if (rr[e] > current) { current = rr[e]; first[seen + 1] = e; ++seen; }
    }
  }
  std::vector<char> taken((size_t)n, 0);
  int32_t next = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t v = order[i];
    if (v < 0 || v >= n) TGCN_FAIL(TGCN_ERR_INVALID, "graclus_match: visiting order outside [0, n)");
    if (taken[v]) continue;
    taken[v] = 1;
    int64_t best = -1;
    T best_w = (T)0;
    for (int64_t e = first[v]; e < first[v] + extent[v] && e < nnz; ++e) {
      const int64_t u = cc[e];
      const T w = taken[u] ? (T)0 : vv[e] * ((T)1 / weight[v] + (T)1 / weight[u]);
      if (w > best_w) { best_w = w; best = u; }
    }
    cluster[v] = next;
    if (best >= 0) { cluster[best] = next; taken[best] = 1; }
    ++next;
  }
  return TGCN_OK;
}

}  // namespace

extern "C" {
int tgcn_graclus_match_f32(int64_t nnz, const int64_t* rr, const int64_t* cc, const float* vv, int64_t n, const int64_t* order,
                           const float* weight, int32_t* cluster) {
  return graclus_match<float>(nnz, rr, cc, vv, n, order, weight, cluster);
}
int tgcn_graclus_match_f64(int64_t nnz, const int64_t* rr, const int64_t* cc, const double* vv, int64_t n, const int64_t* order,
                           const double* weight, int32_t* cluster) {
  return graclus_match<double>(nnz, rr, cc, vv, n, order, weight, cluster);
}
}  // extern "C"

namespace {
