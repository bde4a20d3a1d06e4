// device_build.h -- operand and schedule construction ON THE DEVICE: the index plumbing the reference does on the
// host before every layer (tgcn/nn/gcn.py:398-413: self-loop removal, source-degree normalisation, scatter order;
// gcn/graph.py:117-136: rescaled Laplacian) and the work schedule of hop_kernel, as HIP kernels of this library:
//   scan_i64            exclusive / inclusive prefix sum (tile scan + recursive scan of the tile sums + add-back)
//   radix_sort_pairs    stable LSD radix sort of (uint32 key, uint32 payload), 8 bits per pass: per-block digit histogram, one scan
//                       over the (digit, block) table, stable scatter with a wave-level multisplit (ballot per key bit)
//   csr_build           COO in any order -> rows sorted by (row, col), duplicates kept in their given order (two stable sorts:
//                       by column, then by row), packed {col, val} entries, int32 row pointers
//   sched_build         row-block marks (binary search in the cost prefix sum), long rows by decreasing segment count (stable),
//                       segment cut, column-ordered segment sweep (stable sort by first column)
// Everything is integer work with a fixed order: results are bit-identical to the torch builders of tgcn_amd/graph.py, which stay
// as the cross-check (tests/test_c_abi_graph.py, tests/test_device_build.py).
// Part of the single translation unit tgcn_hip.hip (included once, inside its anonymous namespace).
#pragma once

// --------------------------------------------------------------------------------------------------
// prefix sums
// --------------------------------------------------------------------------------------------------
constexpr int kScanPer = 8;                       // elements per thread
constexpr int kScanTile = kBlock * kScanPer;      // elements per workgroup

// out[i] = sum of in[0..i) (exclusive) or in[0..i] (inclusive) within the tile; tile_sums[tile] = the tile's total
// in and out may be the SAME array (scan_i64 is called in place for the radix histogram, counts -> rowptr and the cost prefix): no
// __restrict__ on them; every thread loads its elements into registers before its first store.
__global__ __launch_bounds__(kBlock) void scan_tile_kernel(const int64_t* in, int64_t* out,
                                                           int64_t* __restrict__ tile_sums, int64_t n, int inclusive) {
  __shared__ int64_t wave_tot[kBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)tid * kScanPer;
  int64_t v[kScanPer];
  int64_t run = 0;
#pragma unroll
  for (int j = 0; j < kScanPer; ++j) {
    v[j] = (base + j < n) ? in[base + j] : 0;
    run += v[j];
  }
  // inclusive scan of the per-thread totals inside the wave
  int64_t inc = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int64_t wave_base = 0;
  for (int w = 0; w < wave; ++w) wave_base += wave_tot[w];
  int64_t acc = wave_base + inc - run;            // sum of everything before this thread's elements
#pragma unroll
  for (int j = 0; j < kScanPer; ++j) {
    if (base + j < n) out[base + j] = inclusive ? acc + v[j] : acc;
    acc += v[j];
  }
  if (tid == kBlock - 1 && tile_sums) tile_sums[blockIdx.x] = acc;
}

__global__ __launch_bounds__(kBlock) void scan_add_kernel(int64_t* __restrict__ out, const int64_t* __restrict__ tile_off, int64_t n) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
  const int64_t off = tile_off[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanPer; ++j)
    if (base + j < n) out[base + j] += off;
}

inline size_t scan_ws_elems(int64_t n) {       // int64 elements of scratch for scan_i64 on n elements
  size_t total = 0;
  while (n > kScanTile) {
    n = (n + kScanTile - 1) / kScanTile;
    total += (size_t)n;
  }
  return total + 1;
}

// in may equal out.  ws: scan_ws_elems(n) int64.
inline void scan_i64(hipStream_t st, const int64_t* in, int64_t* out, int64_t n, int inclusive, int64_t* ws) {
  if (n <= 0) return;
  const int64_t tiles = (n + kScanTile - 1) / kScanTile;
  if (tiles == 1) {
    hipLaunchKernelGGL(scan_tile_kernel, dim3(1), dim3(kBlock), 0, st, in, out, (int64_t*)nullptr, n, inclusive);
    return;
  }
  hipLaunchKernelGGL(scan_tile_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, st, in, out, ws, n, inclusive);
  scan_i64(st, ws, ws, tiles, 0, ws + tiles);                      // exclusive scan of the tile sums, in place
  hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, st, out, ws, n);
}

// --------------------------------------------------------------------------------------------------
// stable LSD radix sort of (key, payload) pairs
// --------------------------------------------------------------------------------------------------
constexpr int kSortChunks = 16;
constexpr int kSortTile = kBlock * kSortChunks;   // keys per workgroup and pass

__global__ __launch_bounds__(kBlock) void radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int shift,
                                                            int64_t* __restrict__ hist, int nblocks) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll 4
  for (int c = 0; c < kSortChunks; ++c) {
    const int64_t i = base + (int64_t)c * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);      // integer counts: the totals do not depend on the order
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];   // digit-major: one scan gives every (digit, block) offset
}

__global__ __launch_bounds__(kBlock) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                               uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                               int64_t n, int shift, const int64_t* __restrict__ offsets, int nblocks) {
  __shared__ unsigned int digit_base[256];            // keys of each digit this workgroup has placed so far
  __shared__ unsigned int wave_cnt[kBlock / 64][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  digit_base[tid] = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) wave_cnt[w][tid] = 0;
  __syncthreads();
  const int64_t my_off = offsets[(int64_t)tid * nblocks + blockIdx.x];   // thread t holds digit t's global offset for this block
  __shared__ int64_t block_off[256];
  block_off[tid] = my_off;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
  for (int c = 0; c < kSortChunks; ++c) {
    const int64_t i = base + (int64_t)c * kBlock + tid;
    const bool valid = i < n;
    const uint32_t k = valid ? keys_in[i] : 0u;
    const uint32_t pv = valid ? vals_in[i] : 0u;
    const unsigned d = (k >> shift) & 255u;
    // lanes of this wave holding the same digit (valid ones only), in lane order: a stable rank inside the wave
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      same &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const unsigned rank_w = __popcll(same & ((1ull << lane) - 1ull));
    if (valid && rank_w == 0) wave_cnt[wave][d] = __popcll(same);
    __syncthreads();
    if (valid) {
      unsigned pre = 0;
      for (int w = 0; w < wave; ++w) pre += wave_cnt[w][d];
      const int64_t pos = block_off[d] + digit_base[d] + pre + rank_w;
      keys_out[pos] = k;
      vals_out[pos] = pv;
    }
    __syncthreads();
    unsigned tot = 0;                                    // thread t closes digit t for this chunk
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { tot += wave_cnt[w][tid]; wave_cnt[w][tid] = 0; }
    digit_base[tid] += tot;
    __syncthreads();
  }
}

struct SortWs { uint32_t *k2, *v2; int64_t *hist, *scan; };    // carved from caller scratch by sort_ws_carve

inline size_t sort_ws_bytes(int64_t n) {
  const int64_t nblocks = (n + kSortTile - 1) / kSortTile;
  return align_up((size_t)(n > 0 ? n : 1) * 4, 256) * 2 + align_up((size_t)(256 * nblocks + 1) * 8, 256) + align_up(scan_ws_elems(256 * nblocks + 1) * 8, 256);
}

inline SortWs sort_ws_carve(char* ws, int64_t n) {
  const int64_t nblocks = (n + kSortTile - 1) / kSortTile;
  SortWs s;
  s.k2 = (uint32_t*)ws; ws += align_up((size_t)(n > 0 ? n : 1) * 4, 256);
  s.v2 = (uint32_t*)ws; ws += align_up((size_t)(n > 0 ? n : 1) * 4, 256);
  s.hist = (int64_t*)ws; ws += align_up((size_t)(256 * nblocks + 1) * 8, 256);
  s.scan = (int64_t*)ws;
  return s;
}

// Sorts keys / vals (n pairs, keys < 2^bits) in place, stable.  ws: sort_ws_bytes(n).
inline void radix_sort_pairs(hipStream_t st, uint32_t* keys, uint32_t* vals, int64_t n, int bits, char* ws) {
  if (n <= 1) return;
  const SortWs w = sort_ws_carve(ws, n);
  const int nblocks = (int)((n + kSortTile - 1) / kSortTile);
  uint32_t *ka = keys, *va = vals, *kb = w.k2, *vb = w.v2;
  int passes = (bits + 7) / 8;
  if (passes < 1) passes = 1;
  for (int p = 0; p < passes; ++p) {
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(kBlock), 0, st, ka, n, 8 * p, w.hist, nblocks);
    scan_i64(st, w.hist, w.hist, (int64_t)256 * nblocks, 0, w.scan);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblocks), dim3(kBlock), 0, st, ka, va, kb, vb, n, 8 * p, w.hist, nblocks);
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  if (ka != keys) {        // odd number of passes: the result sits in the scratch pair
    (void)hipMemcpyAsync(keys, ka, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(vals, va, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
  }
}

inline int bits_for(int64_t max_plus_one) {      // bits needed for keys in [0, max_plus_one)
  int b = 1;
  while (b < 32 && ((int64_t)1 << b) < max_plus_one) ++b;
  return b;
}

// --------------------------------------------------------------------------------------------------
// small index kernels
// --------------------------------------------------------------------------------------------------
__global__ void iota_kernel(uint32_t* __restrict__ v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}

// keys[i] = (uint32) src[idx ? idx[i] : i]; any value outside [0, limit) raises the flag (an out-of-range vertex id would be
// an out-of-bounds read inside the kernels)
__global__ void gather_key_kernel(const int64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t* __restrict__ keys,
                                  int64_t n, int64_t limit, int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = src[idx ? (int64_t)idx[i] : i];
    if (v < 0 || v >= limit) { *bad = 1; keys[i] = 0; }
    else keys[i] = (uint32_t)v;
  }
}

// entries[e] = {col[order[e]], val[order[e]]}; counts[row of e] += 1 (sorted_rows[e] is the row key of position e)
__global__ void pack_edges_kernel(const int64_t* __restrict__ col, const float* __restrict__ val, const uint32_t* __restrict__ order,
                                  const uint32_t* __restrict__ sorted_rows, tgcn_edge* __restrict__ edges, int64_t* __restrict__ counts,
                                  int64_t nnz) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = order[e];
    tgcn_edge t;
    t.col = (int32_t)col[s];
    t.val = val[s];
    edges[e] = t;
    atomicAdd((unsigned long long*)&counts[sorted_rows[e]], 1ull);     // integer counts: order-independent
  }
}

__global__ void i64_to_i32_kernel(const int64_t* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (int32_t)in[i];
}

inline size_t csr_build_ws_bytes(int64_t n, int64_t nnz) {
  const size_t m = (size_t)(nnz > 0 ? nnz : 1);
  return align_up(m * 4, 256) * 2 + sort_ws_bytes(nnz) + align_up((size_t)(n + 2) * 8, 256) + align_up(scan_ws_elems(n + 1) * 8, 256) + 256;
}

// COO (device arrays, any order) -> rowptr int32 [n+1], packed entries [nnz] sorted by (row, col), duplicates in their given order.
// Returns TGCN_ERR_INVALID for an index outside [0, n) x [0, n_cols).  Synchronises once (the range flag).
inline int csr_build_device(hipStream_t st, int64_t n, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col,
                            const float* val, int32_t* rowptr, tgcn_edge* edges, char* ws) {
  uint32_t* keys = (uint32_t*)ws; ws += align_up((size_t)(nnz > 0 ? nnz : 1) * 4, 256);
  uint32_t* order = (uint32_t*)ws; ws += align_up((size_t)(nnz > 0 ? nnz : 1) * 4, 256);
  char* sort_ws = ws; ws += sort_ws_bytes(nnz);
  int64_t* counts = (int64_t*)ws; ws += align_up((size_t)(n + 2) * 8, 256);
  int64_t* scan_ws = (int64_t*)ws; ws += align_up(scan_ws_elems(n + 1) * 8, 256);
  int* bad = (int*)ws;
  if (hipMemsetAsync(bad, 0, sizeof(int), st) != hipSuccess || hipMemsetAsync(counts, 0, (size_t)(n + 2) * 8, st) != hipSuccess)
    TGCN_FAIL(TGCN_ERR_LAUNCH, "csr_build: memset failed");
  if (nnz > 0) {
    const unsigned g = grid_1d(nnz);
    // stable sort by column, then stable sort by row = sorted by (row, col) with duplicates in their given order
    hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(kBlock), 0, st, order, nnz);
    hipLaunchKernelGGL(gather_key_kernel, dim3(g), dim3(kBlock), 0, st, col, (const uint32_t*)nullptr, keys, nnz, n_cols, bad);
    radix_sort_pairs(st, keys, order, nnz, bits_for(n_cols), sort_ws);
    hipLaunchKernelGGL(gather_key_kernel, dim3(g), dim3(kBlock), 0, st, row, (const uint32_t*)order, keys, nnz, n, bad);
    radix_sort_pairs(st, keys, order, nnz, bits_for(n), sort_ws);
    int h_bad = 0;
    if (hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      TGCN_FAIL(TGCN_ERR_LAUNCH, "csr_build: device read failed");
    if (h_bad) TGCN_FAIL(TGCN_ERR_INVALID, "graph: vertex index outside [0, %lld) x [0, %lld)", (long long)n, (long long)n_cols);
    hipLaunchKernelGGL(pack_edges_kernel, dim3(g), dim3(kBlock), 0, st, col, val, order, keys, edges, counts, nnz);
  }
  scan_i64(st, counts, counts, n + 1, 0, scan_ws);                  // counts[n] = 0: the exclusive scan over n+1 gives rowptr[n] = nnz
  hipLaunchKernelGGL(i64_to_i32_kernel, dim3(grid_1d(n + 1)), dim3(kBlock), 0, st, counts, rowptr, n + 1);
  TGCN_CHECK_LAUNCH("csr_build");
  return TGCN_OK;
}

// --------------------------------------------------------------------------------------------------
// edge list -> normalised operand (tgcn/nn/gcn.py:398-413): device kernels
// --------------------------------------------------------------------------------------------------
// keep[e] = row != col (self loops removed); also range-checks both ends
__global__ void edge_keep_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t n, int64_t* __restrict__ keep, int* __restrict__ bad) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = ei[e], b = ei[E + e];
    if (a < 0 || a >= n || b < 0 || b >= n) { *bad = 1; keep[e] = 0; }
    else keep[e] = (a != b) ? 1 : 0;
  }
}

// compaction in the given order (pos = exclusive scan of keep) + unweighted source-degree count
__global__ void edge_compact_kernel(const int64_t* __restrict__ ei, const float* __restrict__ w, int64_t E, const int64_t* __restrict__ keep,
                                    const int64_t* __restrict__ pos, int64_t* __restrict__ row, int64_t* __restrict__ col,
                                    float* __restrict__ val, unsigned int* __restrict__ deg) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    if (!keep[e]) continue;
    const int64_t o = pos[e];
    row[o] = ei[e];
    col[o] = ei[E + e];
    val[o] = w ? w[e] : 1.f;
    atomicAdd(&deg[ei[e]], 1u);
  }
}

// lap_e = -deg^-1/2[row] * w_e * deg^-1/2[col], deg^-1/2 = 0 where the degree is 0 (1 / sqrtf like the host builder)
__global__ void edge_normalise_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col, float* __restrict__ val,
                                      const unsigned int* __restrict__ deg, int64_t m) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (int64_t)gridDim.x * blockDim.x) {
    const unsigned da = deg[row[e]], db = deg[col[e]];
    const float a = da ? 1.0f / sqrtf((float)da) : 0.f, b = db ? 1.0f / sqrtf((float)db) : 0.f;
    val[e] = -a * val[e] * b;
  }
}

inline size_t edge_norm_ws_bytes(int64_t n, int64_t E) {
  const size_t m1 = (size_t)(E > 0 ? E : 1) + 1;
  return align_up(m1 * 8, 256) * 2 + align_up(scan_ws_elems(E + 1) * 8, 256) + align_up((size_t)n * 4, 256) + 256;
}

// Edge list (2, E) [+ weights] -> COO of the ChebConv / ChebTimeConv operand in caller memory (row / col / val hold up to E entries),
// in the given order; *kept_host = entries written.  Two read-backs (range flag, count).  ws: edge_norm_ws_bytes(n, E).
inline int edge_normalise_device(hipStream_t st, int64_t n, int64_t E, const int64_t* ei, const float* w, int64_t* row, int64_t* col,
                                 float* val, int64_t* kept_host, char* ws) {
  const size_t m1 = (size_t)(E > 0 ? E : 1) + 1;
  int64_t* keep = (int64_t*)ws; ws += align_up(m1 * 8, 256);
  int64_t* pos = (int64_t*)ws; ws += align_up(m1 * 8, 256);
  int64_t* scanws = (int64_t*)ws; ws += align_up(scan_ws_elems(E + 1) * 8, 256);
  unsigned int* deg = (unsigned int*)ws; ws += align_up((size_t)n * 4, 256);
  int* bad = (int*)ws;
  *kept_host = 0;
  if (E <= 0) return TGCN_OK;
  if (hipMemsetAsync(keep, 0, m1 * 8, st) != hipSuccess || hipMemsetAsync(deg, 0, (size_t)n * 4, st) != hipSuccess ||
      hipMemsetAsync(bad, 0, sizeof(int), st) != hipSuccess)
    TGCN_FAIL(TGCN_ERR_LAUNCH, "edge_normalise: memset failed");
  hipLaunchKernelGGL(edge_keep_kernel, dim3(grid_1d(E)), dim3(kBlock), 0, st, ei, E, n, keep, bad);
  scan_i64(st, keep, pos, E + 1, 0, scanws);                                    // pos[E] = number of kept edges
  int h_bad = 0;
  int64_t kept = 0;
  if (hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(&kept, pos + E, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    TGCN_FAIL(TGCN_ERR_LAUNCH, "edge_normalise: device read failed");
  if (h_bad) TGCN_FAIL(TGCN_ERR_INVALID, "edge list: vertex index outside [0, %lld)", (long long)n);
  hipLaunchKernelGGL(edge_compact_kernel, dim3(grid_1d(E)), dim3(kBlock), 0, st, ei, w, E, (const int64_t*)keep, (const int64_t*)pos, row, col, val, deg);
  if (kept > 0)
    hipLaunchKernelGGL(edge_normalise_kernel, dim3(grid_1d(kept)), dim3(kBlock), 0, st, (const int64_t*)row, (const int64_t*)col, val, (const unsigned int*)deg, kept);
  TGCN_CHECK_LAUNCH("edge_normalise");
  *kept_host = kept;
  return TGCN_OK;
}

// --------------------------------------------------------------------------------------------------
// adjacency COO -> rescale_L(laplacian(W, normalized=True), lmax) COO (gcn/graph.py:117-136, 232-238): device kernels
// --------------------------------------------------------------------------------------------------
__global__ void range_check_kernel(const int64_t* __restrict__ v, int64_t m, int64_t limit, int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x)
    if (v[i] < 0 || v[i] >= limit) *bad = 1;
}

// cptr[j] = first position of the sorted key array holding a key >= j (j = 0 .. n): column j's entries are [cptr[j], cptr[j+1])
__global__ void lower_bound_kernel(const uint32_t* __restrict__ keys, int64_t m, int64_t n, int64_t* __restrict__ cptr) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= n; j += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = m;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)keys[mid] < j) lo = mid + 1; else hi = mid; }
    cptr[j] = lo;
  }
}

// dis[j] = 1 / sqrt(sum of the weights of column j + eps): one wave per column, lanes take the column's entries (in their given order)
// with stride 64 and fold through a fixed shuffle tree -- no float atomics, the same bits every run
__global__ __launch_bounds__(kBlock) void colsum_rsqrt_kernel(const float* __restrict__ w, const uint32_t* __restrict__ order,
                                                              const int64_t* __restrict__ cptr, int64_t n, float eps, float* __restrict__ dis) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t j = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); j < n; j += nwaves) {
    float acc = 0.f;
    for (int64_t k = cptr[j] + lane; k < cptr[j + 1]; k += 64) acc += w[order[k]];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) dis[j] = 1.0f / sqrtf(acc + eps);
  }
}

// entry e < m: {row, col, -scale * dis[row] * w * dis[col]}; entry m + i (diagonal, only when scale != 1): {i, i, scale - 1}
__global__ void lap_values_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col, const float* __restrict__ w, int64_t m, int64_t n,
                                  const float* __restrict__ dis, float scale, int diag, int64_t* __restrict__ row_out, int64_t* __restrict__ col_out,
                                  float* __restrict__ val_out) {
  const int64_t total = m + (diag ? n : 0);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    if (e < m) {
      const int64_t r = row[e], c = col[e];
      row_out[e] = r; col_out[e] = c;
      val_out[e] = -scale * dis[r] * w[e] * dis[c];
    } else {
      row_out[e] = col_out[e] = e - m;
      val_out[e] = scale - 1.0f;
    }
  }
}

inline size_t adjacency_norm_ws_bytes(int64_t n, int64_t m) {
  const size_t m1 = (size_t)(m > 0 ? m : 1);
  return align_up(m1 * 4, 256) * 2 + sort_ws_bytes(m) + align_up((size_t)(n + 2) * 8, 256) + align_up((size_t)n * 4, 256) + 256;
}

// COO of the weight matrix W (m entries, any order) -> COO of L-hat = (2/lmax) (I - D^-1/2 W D^-1/2) - I, d = colsum(W) + eps, in caller
// memory (m + n slots; the n diagonal entries are written only when lmax != 2).  One read-back (range flag).  ws: adjacency_norm_ws_bytes.
inline int adjacency_normalise_device(hipStream_t st, int64_t n, int64_t m, const int64_t* row, const int64_t* col, const float* w, float lmax,
                                      int64_t* row_out, int64_t* col_out, float* val_out, int64_t* count_host, char* ws) {
  const size_t m1 = (size_t)(m > 0 ? m : 1);
  uint32_t* keys = (uint32_t*)ws; ws += align_up(m1 * 4, 256);
  uint32_t* order = (uint32_t*)ws; ws += align_up(m1 * 4, 256);
  char* sort_ws = ws; ws += sort_ws_bytes(m);
  int64_t* cptr = (int64_t*)ws; ws += align_up((size_t)(n + 2) * 8, 256);
  float* dis = (float*)ws; ws += align_up((size_t)n * 4, 256);
  int* bad = (int*)ws;
  const float scale = 2.0f / lmax;
  const int diag = scale != 1.0f;
  *count_host = 0;
  if (hipMemsetAsync(bad, 0, sizeof(int), st) != hipSuccess) TGCN_FAIL(TGCN_ERR_LAUNCH, "adjacency_normalise: memset failed");
  if (m > 0) {
    const unsigned g = grid_1d(m);
    hipLaunchKernelGGL(range_check_kernel, dim3(g), dim3(kBlock), 0, st, row, m, n, bad);
    hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(kBlock), 0, st, order, m);
    hipLaunchKernelGGL(gather_key_kernel, dim3(g), dim3(kBlock), 0, st, col, (const uint32_t*)nullptr, keys, m, n, bad);
    int h_bad = 0;
    if (hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      TGCN_FAIL(TGCN_ERR_LAUNCH, "adjacency_normalise: device read failed");
    if (h_bad) TGCN_FAIL(TGCN_ERR_INVALID, "adjacency: vertex index outside [0, %lld)", (long long)n);
    radix_sort_pairs(st, keys, order, m, bits_for(n), sort_ws);                  // stable: a column's entries keep their given order
  }
  hipLaunchKernelGGL(lower_bound_kernel, dim3(grid_1d(n + 1)), dim3(kBlock), 0, st, (const uint32_t*)keys, m, n, cptr);
  int64_t cblocks = (n + kBlock / 64 - 1) / (kBlock / 64);
  if (cblocks > 65536) cblocks = 65536;
  hipLaunchKernelGGL(colsum_rsqrt_kernel, dim3((unsigned)cblocks), dim3(kBlock), 0, st, w, (const uint32_t*)order, (const int64_t*)cptr, n,
                     1.401298464324817e-45f /* np.spacing(float32(0)) */, dis);
  const int64_t total = m + (diag ? n : 0);
  if (total > 0)
    hipLaunchKernelGGL(lap_values_kernel, dim3(grid_1d(total)), dim3(kBlock), 0, st, row, col, w, m, n, (const float*)dis, scale, diag, row_out, col_out, val_out);
  TGCN_CHECK_LAUNCH("adjacency_normalise");
  *count_host = total;
  return TGCN_OK;
}

// --------------------------------------------------------------------------------------------------
// schedule
// --------------------------------------------------------------------------------------------------
// cost[i] = (entries of row i if <= row_thresh else 0) + row_cost;  is_seg[i] = entries > wave_max (rows cut into lane-group segments);
// is_wave[i] = row_thresh < entries <= wave_max (whole-row wave segments; wave_max = row_thresh: none)
__global__ void sched_cost_kernel(const int32_t* __restrict__ rowptr, int64_t n, int row_thresh, int wave_max, int row_cost, int64_t* __restrict__ cost,
                                  int64_t* __restrict__ is_seg, int64_t* __restrict__ is_wave, unsigned long long* __restrict__ n_empty) {
  unsigned long long mine = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t d = rowptr[i + 1] - rowptr[i];
    cost[i] = (d > row_thresh ? 0 : d) + row_cost;
    is_seg[i] = d > wave_max ? 1 : 0;
    is_wave[i] = (d > row_thresh && d <= wave_max) ? 1 : 0;
    mine += (d == 0);
  }
  if (mine) atomicAdd(n_empty, mine);      // integer count: order-independent
}

// blk_row[b] = min(searchsorted(cum, b * target) + 1, n) for 1 <= b < nblk (cum = inclusive prefix sum of the costs); ends 0 and n
__global__ void sched_marks_kernel(const int64_t* __restrict__ cum, int64_t n, int64_t target, int64_t nblk, int32_t* __restrict__ blk_row) {
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= nblk; b += (int64_t)gridDim.x * blockDim.x) {
    if (b == 0) { blk_row[0] = 0; continue; }
    if (b == nblk) { blk_row[nblk] = (int32_t)n; continue; }
    const int64_t mark = b * target;
    int64_t lo = 0, hi = n;                       // first index with cum[idx] >= mark
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (cum[mid] < mark) lo = mid + 1; else hi = mid; }
    blk_row[b] = (int32_t)((lo + 1 < n) ? lo + 1 : n);
  }
}

// rows cut into segments, in row order: seg_rows[pos[i]] = i; key = max_key - nsegs (ascending key = decreasing segment count)
__global__ void sched_longrows_kernel(const int32_t* __restrict__ rowptr, int64_t n, const int64_t* __restrict__ is_seg,
                                      const int64_t* __restrict__ pos, int seg_len, uint32_t max_key, uint32_t* __restrict__ seg_rows,
                                      uint32_t* __restrict__ key) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (!is_seg[i]) continue;
    const int64_t d = rowptr[i + 1] - rowptr[i];
    const int64_t o = pos[i];
    seg_rows[o] = (uint32_t)i;
    key[o] = max_key - (uint32_t)((d + seg_len - 1) / seg_len);
  }
}

// whole-row wave segments, in row order: rows[pos[i]] = i, key = first column of the row
__global__ void sched_waverows_kernel(const int32_t* __restrict__ rowptr, const tgcn_edge* __restrict__ edges, int64_t n, const int64_t* __restrict__ is_wave,
                                      const int64_t* __restrict__ pos, uint32_t* __restrict__ rows, uint32_t* __restrict__ key) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (!is_wave[i]) continue;
    const int64_t o = pos[i];
    rows[o] = (uint32_t)i;
    key[o] = (uint32_t)edges[rowptr[i]].col;
  }
}

// the sorted wave rows as the first nw entries of the segment arrays
__global__ void sched_wavesegs_kernel(const int32_t* __restrict__ rowptr, const uint32_t* __restrict__ rows, int64_t nw, int32_t* __restrict__ seg_row,
                                      int32_t* __restrict__ seg_e0, int32_t* __restrict__ seg_e1, int32_t* __restrict__ seg_slot) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows[i];
    seg_row[i] = (int32_t)r; seg_e0[i] = rowptr[r]; seg_e1[i] = rowptr[r + 1]; seg_slot[i] = -1;
  }
}

__global__ void sched_nsegs_kernel(const uint32_t* __restrict__ key, uint32_t max_key, int64_t m, int64_t* __restrict__ nsegs) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) nsegs[i] = max_key - key[i];
}

// segment s belongs to the sorted long row j with first[j] <= s < first[j+1] (first = exclusive scan of nsegs, first[m] = nseg)
__global__ void sched_segments_kernel(const int32_t* __restrict__ rowptr, const tgcn_edge* __restrict__ edges, const uint32_t* __restrict__ seg_rows,
                                      const int64_t* __restrict__ first, int64_t m, int64_t nseg, int64_t npartial, int seg_len,
                                      int32_t* __restrict__ seg_row, int32_t* __restrict__ seg_e0, int32_t* __restrict__ seg_e1,
                                      int32_t* __restrict__ seg_slot, uint32_t* __restrict__ key, uint32_t* __restrict__ ident) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = m;                       // last j with first[j] <= s
    while (lo + 1 < hi) { const int64_t mid = (lo + hi) >> 1; if (first[mid] <= s) lo = mid; else hi = mid; }
    const int64_t r = seg_rows[lo];
    const int64_t e0 = (int64_t)rowptr[r] + (s - first[lo]) * seg_len;
    const int64_t e1 = (e0 + seg_len < (int64_t)rowptr[r + 1]) ? e0 + seg_len : (int64_t)rowptr[r + 1];
    seg_row[s] = (int32_t)r;
    seg_e0[s] = (int32_t)e0;
    seg_e1[s] = (int32_t)e1;
    seg_slot[s] = s < npartial ? (int32_t)s : -1;
    key[s] = (uint32_t)edges[e0].col;
    ident[s] = (uint32_t)s;
  }
}

__global__ void permute_i32_kernel(const int32_t* __restrict__ in, const uint32_t* __restrict__ perm, int32_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[perm[i]];
}

__global__ void sched_long_kernel(const uint32_t* __restrict__ seg_rows, const int64_t* __restrict__ first, int64_t nlong, int64_t npartial,
                                  int32_t* __restrict__ long_row, int32_t* __restrict__ long_slot) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nlong; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nlong) { long_row[i] = (int32_t)seg_rows[i]; long_slot[i] = (int32_t)first[i]; }
    else long_slot[nlong] = (int32_t)npartial;
  }
}

// counts of sorted (descending) segment counts above 1 / above huge_slots: nsegs is non-increasing, so two binary searches
__global__ void sched_counts_kernel(const int64_t* __restrict__ nsegs, int64_t m, int huge_slots, int64_t* __restrict__ out2) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int which = 0; which < 2; ++which) {
    const int64_t thr = which == 0 ? 1 : huge_slots;      // number of leading entries with nsegs > thr
    int64_t lo = 0, hi = m;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (nsegs[mid] > thr) lo = mid + 1; else hi = mid; }
    out2[which] = lo;
  }
}
