// graph_build.h -- one-off construction of the sparse operand and its work schedule inside the library, so that the C ABI
// is usable without the Python package (include/tgcn_hip.h: tgcn_graph_*, tgcn_sched_*).  Host code: the arrays are copied
// from the device, sorted / scanned here and uploaded again -- the only entry points that allocate and synchronise; they
// run once per operand, never on the forward path.  The schedule is the one tgcn_amd/graph.py::Schedule builds (row blocks +
// column-ordered segments; tests/test_c_abi_graph.py compares the two array by array).
// Part of the single translation unit tgcn_hip.hip (included once, inside its anonymous namespace).
#pragma once

struct DeviceBuf {
  void* p = nullptr;
  ~DeviceBuf() { if (p) (void)hipFree(p); }
  int upload(const void* host, size_t bytes) {
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { p = nullptr; return -1; }
    if (bytes && hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return -1;
    return 0;
  }
};

}  // namespace (the opaque handle types are named by the header)

struct tgcn_graph {
  tgcn_csr csr;
  int64_t n_cols;
  DeviceBuf rowptr, edges;
  std::vector<int32_t> h_rowptr;        // host copies: schedules are built from them
  std::vector<tgcn_edge> h_edges;
};

struct tgcn_sched {
  tgcn_csr_sched s;
  DeviceBuf blk_row, seg_row, seg_e0, seg_e1, seg_slot, long_row, long_slot;
};

namespace {

template <typename T>
int fetch(std::vector<T>& dst, const T* dev, size_t count) {
  dst.resize(count);
  if (count && hipMemcpy(dst.data(), dev, count * sizeof(T), hipMemcpyDefault) != hipSuccess) return -1;   // device (the convention) or host memory
  return 0;
}

// rows sorted by (row, col), duplicates kept as separate entries in their given order (their sum is what scatter_add
// computes, tgcn/nn/gcn.py:308,343)
int graph_from_host_coo(int64_t n, int64_t n_cols, const std::vector<int64_t>& row, const std::vector<int64_t>& col,
                        const std::vector<float>& val, tgcn_graph** out) {
  const size_t nnz = row.size();
  for (size_t e = 0; e < nnz; ++e)
    if (row[e] < 0 || row[e] >= n || col[e] < 0 || col[e] >= n_cols)
      TGCN_FAIL(TGCN_ERR_INVALID, "graph: vertex index outside [0, %lld) x [0, %lld)", (long long)n, (long long)n_cols);
  std::vector<int64_t> order(nnz);
  std::iota(order.begin(), order.end(), (int64_t)0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return row[a] != row[b] ? row[a] < row[b] : col[a] < col[b]; });
  tgcn_graph* g = new (std::nothrow) tgcn_graph();
  if (!g) TGCN_FAIL(TGCN_ERR_LAUNCH, "graph: out of host memory");
  g->h_rowptr.assign(n + 1, 0);
  for (size_t e = 0; e < nnz; ++e) g->h_rowptr[row[e] + 1]++;
  for (int64_t i = 0; i < n; ++i) g->h_rowptr[i + 1] += g->h_rowptr[i];
  g->h_edges.resize(nnz ? nnz : 1);
  for (size_t e = 0; e < nnz; ++e) { g->h_edges[e].col = (int32_t)col[order[e]]; g->h_edges[e].val = val[order[e]]; }
  if (g->rowptr.upload(g->h_rowptr.data(), (n + 1) * sizeof(int32_t)) || g->edges.upload(g->h_edges.data(), g->h_edges.size() * sizeof(tgcn_edge))) {
    delete g;
    TGCN_FAIL(TGCN_ERR_LAUNCH, "graph: device allocation / upload failed");
  }
  g->csr.n = n; g->csr.nnz = (int64_t)nnz; g->csr.rowptr = (const int32_t*)g->rowptr.p; g->csr.edges = (const tgcn_edge*)g->edges.p;
  g->csr.dense = nullptr;
  g->n_cols = n_cols;
  *out = g;
  return TGCN_OK;
}

inline bool graph_sizes_ok(int64_t n, int64_t n_cols, int64_t nnz) {
  return n > 0 && n_cols > 0 && nnz >= 0 && n < (int64_t)INT32_MAX && n_cols < (int64_t)INT32_MAX && nnz < (int64_t)INT32_MAX - 1;
}

}  // namespace

extern "C" {

int tgcn_graph_create_from_coo(int64_t n, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col, const float* val,
                               tgcn_graph** out) {
  if (!out || !graph_sizes_ok(n, n_cols, nnz) || (nnz > 0 && (!row || !col || !val))) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_coo: bad argument");
  std::vector<int64_t> r, c;
  std::vector<float> v;
  if (fetch(r, row, (size_t)nnz) || fetch(c, col, (size_t)nnz) || fetch(v, val, (size_t)nnz)) TGCN_FAIL(TGCN_ERR_LAUNCH, "graph_create_from_coo: device read failed");
  return graph_from_host_coo(n, n_cols, r, c, v, out);
}

int tgcn_graph_create_from_csr(int64_t n, int64_t n_cols, const int64_t* rowptr, const int32_t* col, const float* val, tgcn_graph** out) {
  if (!out || !rowptr || n <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_csr: bad argument");
  std::vector<int64_t> rp;
  if (fetch(rp, rowptr, (size_t)n + 1)) TGCN_FAIL(TGCN_ERR_LAUNCH, "graph_create_from_csr: device read failed");
  const int64_t nnz = rp[n];
  if (!graph_sizes_ok(n, n_cols, nnz) || rp[0] != 0 || (nnz > 0 && (!col || !val))) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_csr: bad rowptr");
  std::vector<int32_t> c32;
  std::vector<float> v;
  if (fetch(c32, col, (size_t)nnz) || fetch(v, val, (size_t)nnz)) TGCN_FAIL(TGCN_ERR_LAUNCH, "graph_create_from_csr: device read failed");
  std::vector<int64_t> r((size_t)nnz), c((size_t)nnz);
  for (int64_t i = 0; i < n; ++i) {
    if (rp[i + 1] < rp[i]) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_csr: rowptr decreases at row %lld", (long long)i);
    for (int64_t e = rp[i]; e < rp[i + 1]; ++e) { r[e] = i; c[e] = c32[e]; }
  }
  return graph_from_host_coo(n, n_cols, r, c, v, out);
}

/* ChebConv / ChebTimeConv operand from the caller's edge list (tgcn/nn/gcn.py:398-413 == :495-510): self loops removed,
 * deg = number of edges per SOURCE vertex (unweighted), lap_e = -deg^-1/2[row] * w_e * deg^-1/2[col], deg^-1/2 = 0 for
 * vertices without outgoing edges. */
int tgcn_graph_create_from_edge_index(int64_t n, int64_t E, const int64_t* edge_index, const float* edge_weight, tgcn_graph** out) {
  if (!out || !graph_sizes_ok(n, n, E) || (E > 0 && !edge_index)) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_edge_index: bad argument");
  std::vector<int64_t> ei;
  std::vector<float> w;
  if (fetch(ei, edge_index, (size_t)(2 * E)) || (edge_weight && fetch(w, edge_weight, (size_t)E))) TGCN_FAIL(TGCN_ERR_LAUNCH, "graph_create_from_edge_index: device read failed");
  std::vector<int64_t> r, c;
  std::vector<float> v;
  std::vector<float> deg((size_t)n, 0.f);
  for (int64_t e = 0; e < E; ++e) {
    const int64_t a = ei[e], b = ei[E + e];
    if (a < 0 || a >= n || b < 0 || b >= n) TGCN_FAIL(TGCN_ERR_INVALID, "graph_create_from_edge_index: vertex index outside [0, %lld)", (long long)n);
    if (a == b) continue;
    r.push_back(a); c.push_back(b); v.push_back(edge_weight ? w[e] : 1.f);
    deg[a] += 1.f;
  }
  std::vector<float> dis((size_t)n);
  for (int64_t i = 0; i < n; ++i) dis[i] = deg[i] > 0.f ? 1.0f / sqrtf(deg[i]) : 0.f;
  for (size_t e = 0; e < r.size(); ++e) v[e] = -dis[r[e]] * v[e] * dis[c[e]];
  return graph_from_host_coo(n, n, r, c, v, out);
}

const tgcn_csr* tgcn_graph_csr(const tgcn_graph* g) { return g ? &g->csr : nullptr; }
int64_t tgcn_graph_n_cols(const tgcn_graph* g) { return g ? g->n_cols : 0; }
void tgcn_graph_destroy(tgcn_graph* g) { delete g; }

/* Row-block + column-ordered-segment schedule of `g` for rows of C floats (the arrays documented at tgcn_csr_sched). */
int tgcn_sched_build(const tgcn_graph* g, int32_t C, int aligned16, tgcn_sched** out) {
  if (!g || !out || C <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "sched_build: bad argument");
  const int lanes = hop_geom(C, aligned16).lpr;
  const int gpb = kBlock / lanes;
  const int64_t n = g->csr.n;
  const int32_t seg_mode = 0;                             // lane-group segments, as tgcn_amd/graph.py::SEG_MODE (wave segments measured slower on cfg5)
  const int32_t row_thresh = 32, seg_len = seg_mode == 1 ? 32 * (64 / lanes) : 32, huge_slots = 64, row_cost = 4;
  const int64_t max_blocks_hint = 2048;
  const std::vector<int32_t>& rp = g->h_rowptr;
  // ---- short rows: nnz-balanced row blocks (cost = entries + 4 per row; long rows cost 4)
  std::vector<int64_t> cum((size_t)n);
  int64_t total = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t d = rp[i + 1] - rp[i];
    total += (d > row_thresh ? 0 : d) + row_cost;
    cum[i] = total;
  }
  const int64_t cap = lanes <= 16 ? 64 : 256;        // narrow rows: small blocks keep an XCD's gather window inside its L2 (tgcn_amd/graph.py)
  const int64_t target = std::max<int64_t>(gpb * 16, std::min<int64_t>(gpb * cap, (total + max_blocks_hint - 1) / max_blocks_hint));
  const int64_t nblk = std::max<int64_t>(1, (total + target - 1) / target);
  std::vector<int32_t> blk_row;
  blk_row.push_back(0);
  for (int64_t b = 1; b < nblk; ++b) {
    const int64_t mark = b * target;
    const int64_t pos = std::lower_bound(cum.begin(), cum.end(), mark) - cum.begin();      // searchsorted(cum, mark)
    blk_row.push_back((int32_t)std::min<int64_t>(pos + 1, n));
  }
  blk_row.push_back((int32_t)n);
  // ---- longer rows: segments; rows with several segments ("long") first, by decreasing segment count (stable)
  std::vector<int64_t> seg_rows;
  for (int64_t i = 0; i < n; ++i) if (rp[i + 1] - rp[i] > row_thresh) seg_rows.push_back(i);
  std::vector<int64_t> nsegs(seg_rows.size());
  for (size_t i = 0; i < seg_rows.size(); ++i) nsegs[i] = (rp[seg_rows[i] + 1] - rp[seg_rows[i]] + seg_len - 1) / seg_len;
  std::vector<size_t> ord(seg_rows.size());
  std::iota(ord.begin(), ord.end(), (size_t)0);
  std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return nsegs[a] > nsegs[b]; });
  std::vector<int32_t> seg_row, seg_e0, seg_e1, seg_slot, long_row, long_slot;
  int64_t nlong = 0, nhuge = 0, npartial = 0, nseg = 0;
  for (size_t i : ord) { if (nsegs[i] > 1) { ++nlong; npartial += nsegs[i]; } if (nsegs[i] > huge_slots) ++nhuge; nseg += nsegs[i]; }
  std::vector<int64_t> key;
  {
    int64_t slot = 0;
    for (size_t i : ord) {
      const int64_t r = seg_rows[i];
      if (nsegs[i] > 1) { long_row.push_back((int32_t)r); long_slot.push_back((int32_t)slot); }
      for (int64_t s = 0; s < nsegs[i]; ++s, ++slot) {
        const int64_t e0 = rp[r] + s * seg_len, e1 = std::min<int64_t>(e0 + seg_len, rp[r + 1]);
        seg_row.push_back((int32_t)r); seg_e0.push_back((int32_t)e0); seg_e1.push_back((int32_t)e1);
        seg_slot.push_back(slot < npartial ? (int32_t)slot : -1);
        key.push_back(g->h_edges[e0].col);
      }
    }
    if (nlong) long_slot.push_back((int32_t)npartial);
  }
  // processing order: by first column (stable)
  std::vector<size_t> perm((size_t)nseg);
  std::iota(perm.begin(), perm.end(), (size_t)0);
  std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
  auto permute = [&](std::vector<int32_t>& v) { std::vector<int32_t> t(v.size()); for (size_t i = 0; i < v.size(); ++i) t[i] = v[perm[i]]; v.swap(t); };
  permute(seg_row); permute(seg_e0); permute(seg_e1); permute(seg_slot);
  if (seg_row.empty()) { seg_row.push_back(0); seg_e0.push_back(0); seg_e1.push_back(0); seg_slot.push_back(0); }
  if (long_row.empty()) long_row.push_back(0);
  if (long_slot.size() < 2) long_slot.assign(2, 0);
  tgcn_sched* sc = new (std::nothrow) tgcn_sched();
  if (!sc) TGCN_FAIL(TGCN_ERR_LAUNCH, "sched_build: out of host memory");
  auto up = [&](DeviceBuf& b, const std::vector<int32_t>& v) { return b.upload(v.data(), v.size() * sizeof(int32_t)); };
  if (up(sc->blk_row, blk_row) || up(sc->seg_row, seg_row) || up(sc->seg_e0, seg_e0) || up(sc->seg_e1, seg_e1) || up(sc->seg_slot, seg_slot) ||
      up(sc->long_row, long_row) || up(sc->long_slot, long_slot)) {
    delete sc;
    TGCN_FAIL(TGCN_ERR_LAUNCH, "sched_build: device allocation / upload failed");
  }
  memset(&sc->s, 0, sizeof(sc->s));
  sc->s.lanes_per_row = lanes; sc->s.row_thresh = row_thresh; sc->s.nblk = (int32_t)nblk; sc->s.nseg = (int32_t)nseg;
  sc->s.nlong = (int32_t)nlong; sc->s.nhuge = (int32_t)nhuge; sc->s.npartial = (int32_t)npartial; sc->s.seg_mode = seg_mode;
  sc->s.blk_row = (const int32_t*)sc->blk_row.p; sc->s.seg_row = (const int32_t*)sc->seg_row.p; sc->s.seg_e0 = (const int32_t*)sc->seg_e0.p;
  sc->s.seg_e1 = (const int32_t*)sc->seg_e1.p; sc->s.seg_slot = (const int32_t*)sc->seg_slot.p; sc->s.long_row = (const int32_t*)sc->long_row.p;
  sc->s.long_slot = (const int32_t*)sc->long_slot.p;
  *out = sc;
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
