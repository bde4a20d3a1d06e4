// pool_relayout.h -- (Q,n,C) -> (n,Q,C) re-layout, gcn_pool / gcn_pool_4 and the fused relu + pool epilogue pass
// Part of the single translation unit tgcn_hip.hip (included once, inside its anonymous namespace).
#pragma once

// --------------------------------------------------------------------------------------------------
// relayout (Q,n,C) -> (n,Q,C), C <= 32
// --------------------------------------------------------------------------------------------------
constexpr int kRelT = 16;           // samples per tile
// vt vertices per tile: 16 for C >= 4, more for shorter rows so that a tile row read is >= 256 contiguous bytes
__global__ __launch_bounds__(kBlock) void relayout_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int64_t Q, int64_t n, int C, int vt) {
  __shared__ float tile[kRelT * kRelT * 32];          // vt * C <= 512 floats per sample row
  const int64_t i0 = (int64_t)blockIdx.x * vt, q0 = (int64_t)blockIdx.y * kRelT;
  const int seg = vt * C;             // floats per (sample, vt vertices)
  for (int e = threadIdx.x; e < kRelT * seg; e += kBlock) {
    const int q = e / seg, rem = e % seg;
    float v = 0.f;
    if (q0 + q < Q && i0 + rem / C < n) v = in[((q0 + q) * n + i0) * C + rem];
    tile[e] = v;
  }
  __syncthreads();
  const int oseg = kRelT * C;         // floats per (vertex, 16 samples)
  for (int e = threadIdx.x; e < vt * oseg; e += kBlock) {
    const int i = e / oseg, rem = e % oseg;
    const int q = rem / C, c = rem % C;
    if (i0 + i < n && q0 + q < Q) out[((i0 + i) * Q + q0) * C + rem] = tile[(q * vt + i) * C + c];
  }
}

inline int relayout_vertex_tile(int C) { return C >= 4 ? 16 : (C >= 2 ? 32 : 64); }

// --------------------------------------------------------------------------------------------------
// pooling
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void relu_pool_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           uint8_t* __restrict__ idx, int64_t total, int f, int p) {
  for (int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (int64_t)gridDim.x * kBlock) {
    const int64_t row = o / f;
    const int c = (int)(o % f);
    const float* src = x + row * p * f + c;
    float best = src[0];
    int bi = 0;
    for (int j = 1; j < p; ++j) {
      const float v = src[(int64_t)j * f];
      if (v > best || (v != v && best == best)) { best = v; bi = j; }
    }
    out[o] = best > 0.f ? best : (best != best ? best : 0.f);
    if (idx) idx[o] = (uint8_t)bi;
  }
}

// grad wrt the layer output of max-pool(relu(.)): the pooled gradient goes to the arg-max vertex where z > 0
__global__ __launch_bounds__(kBlock) void relu_pool_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ z,
                                                               const uint8_t* __restrict__ idx, float* __restrict__ gy,
                                                               int64_t total, int f, int p) {
  for (int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (int64_t)gridDim.x * kBlock) {
    const int64_t row = o / f;
    const int c = (int)(o % f);
    const int bi = idx[o];
    const float g = z[o] > 0.f ? gz[o] : 0.f;
    float* dst = gy + row * p * f + c;
    for (int j = 0; j < p; ++j) dst[(int64_t)j * f] = (j == bi) ? g : 0.f;
  }
}

__global__ __launch_bounds__(kBlock) void pool_max_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          int32_t* __restrict__ idx, int64_t total, int f, int p) {
  for (int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (int64_t)gridDim.x * kBlock) {
    const int64_t row = o / f;  // (q, i_out) flattened
    const int c = (int)(o % f);
    const float* src = x + row * p * f + c;
    float best = src[0];
    int bi = 0;
    for (int j = 1; j < p; ++j) {
      const float v = src[(int64_t)j * f];
      if (v > best || (v != v && best == best)) {  // NaN propagates like torch.max
        best = v;
        bi = j;
      }
    }
    out[o] = best;
    if (idx) idx[o] = bi;
  }
}

__global__ __launch_bounds__(kBlock) void pool_max_bwd_kernel(const float* __restrict__ go, const int32_t* __restrict__ idx,
                                                              float* __restrict__ gi, int64_t total, int f, int p) {
  for (int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (int64_t)gridDim.x * kBlock) {
    const int64_t row = o / f;
    const int c = (int)(o % f);
    const int bi = idx[o];
    const float g = go[o];
    float* dst = gi + row * p * f + c;
    for (int j = 0; j < p; ++j) dst[(int64_t)j * f] = (j == bi) ? g : 0.f;
  }
}

inline int grid_1d(int64_t total) {
  int64_t g = (total + kBlock - 1) / kBlock;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// out[i, :] = src[idx[i], :]: the rows a neighbouring vertex shard asked for, packed for one message (tgcn_amd/dist.py).
__global__ __launch_bounds__(kBlock) void pack_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                           float* __restrict__ out, int64_t nrows, int32_t C, int64_t ld_src) {
  const int64_t total = nrows * C;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    out[i] = src[idx[r] * ld_src + c];
  }
}

// out[r, :] = src[idx[r], :] with int32 row ids and 16-byte pieces when the rows allow it: T_0 of the compacted Chebyshev layer (the kept rows of
// x packed to compact ids, tgcn_cheb_compact_layer_f32 mode 1) and the wide-row form of tgcn_pack_rows_f32
template <int VEC>
__global__ __launch_bounds__(kBlock) void gather_rows_i32_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, float* __restrict__ out,
                                                                 int64_t nrows, int32_t C, int64_t ld_src) {
  const int per = C / VEC;
  const int64_t total = nrows * per;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / per;
    const int c = (int)(i - r * per) * VEC;
    float v[VEC];
    load_vec<VEC>(src + (int64_t)idx[r] * ld_src + c, v);
    store_vec<VEC>(out + r * C + c, v);
  }
}

// (Q, n, C) -> (n, Q, C) for rows of more than 32 floats: every row is >= 128 contiguous bytes, so a plain row copy is already coalesced
// (the LDS-tiled transpose above is for the short rows it was built for)
template <int VEC>
__global__ __launch_bounds__(kBlock) void relayout_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t Q, int64_t n, int32_t C) {
  const int per = C / VEC;
  const int64_t total = Q * n * per;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t orow = i / per;                    // output row = vertex * Q + sample
    const int c = (int)(i - orow * per) * VEC;
    const int64_t v = orow / Q, b = orow - v * Q;
    float t[VEC];
    load_vec<VEC>(in + (b * n + v) * C + c, t);
    store_vec<VEC>(out + orow * C + c, t);
  }
}

// Sampled dense-dense product over the stored pattern: the gradient of a hop w.r.t. its operand VALUES (tgcn/nn/gcn.py:296-308, 413, 510: the
// reference's gather / scale / scatter_add form is differentiable in `value` / `edge_weight`):
//   dval[e] (+)= alpha * sum_b sum_c A[b, row(e), c] * B[b, col(e), c]          for every stored entry e
// One 16-lane group per entry (balanced on power-law graphs: no row is a unit of work); the group finds its row by binary search in the row
// pointers, walks the C channels 64 at a time (one float4 per lane) over all nb samples, and folds through a fixed shuffle tree: one writer
// per entry, no atomics, the same bits every run.
template <int VEC>
__global__ __launch_bounds__(kBlock) void sddmm_kernel(int64_t n, int64_t nnz, const int32_t* __restrict__ rowptr, const tgcn_edge* __restrict__ ev,
                                                       int32_t nb, int32_t C, const float* __restrict__ A, int64_t a_bs, int64_t a_ld,
                                                       const float* __restrict__ B, int64_t b_bs, int64_t b_ld, float alpha,
                                                       float* __restrict__ dval, int accumulate) {
  const int t = threadIdx.x & 15;
  const int64_t groups = (int64_t)gridDim.x * (kBlock / 16);
  for (int64_t e = (int64_t)blockIdx.x * (kBlock / 16) + (threadIdx.x >> 4); e < nnz; e += groups) {
    int64_t lo = 0, hi = n;                               // row of entry e: the last r with rowptr[r] <= e
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)rowptr[mid] <= e) lo = mid; else hi = mid; }
    const int64_t r = lo, c = ev[e].col;
    float acc = 0.f;
    for (int32_t b = 0; b < nb; ++b) {
      const float* __restrict__ ar = A + (int64_t)b * a_bs + r * a_ld;
      const float* __restrict__ br = B + (int64_t)b * b_bs + c * b_ld;
      for (int32_t k = t * VEC; k < C; k += 16 * VEC) {
        if constexpr (VEC == 4) {
          const float4 x = *reinterpret_cast<const float4*>(ar + k), y = *reinterpret_cast<const float4*>(br + k);
          acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
        } else {
          acc = fmaf(ar[k], br[k], acc);
        }
      }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
    if (t == 0) dval[e] = accumulate ? fmaf(alpha, acc, dval[e]) : alpha * acc;
  }
}

// fp64 hop for the numpy twin gcn.graph.chebyshev with float64 operands (the reference computes in L.dtype, gcn/graph.py:247,
// 256-265): S = L X; P = S (optional); Y = alpha S + beta Z.  One thread per output element, entries in stored order
// (deterministic); these calls are small (M ~ 1e3 rows, N ~ batch columns), so no schedule.
__global__ __launch_bounds__(kBlock) void hop_f64_kernel(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const double* __restrict__ val, int64_t F, const double* __restrict__ X,
                                                         const double* __restrict__ Z, double alpha, double beta,
                                                         double* __restrict__ Y, double* __restrict__ P) {
  const int64_t c = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t r = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (r >= n || c >= F) return;
  double acc = 0.0;
  for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) acc = fma(val[e], X[(int64_t)col[e] * F + c], acc);
  if (P) P[r * F + c] = acc;
  if (Y) Y[r * F + c] = Z ? fma(alpha, acc, beta * Z[r * F + c]) : alpha * acc;
}

// out[j, :] = sum_k fold[k, j] * W[k, :]  (transpose != 0: fold[j, k]): the K x K change of basis between the reference's
// recursion and the monomials (tgcn_amd/functional.py::power_fold_matrix) applied to a (K, CN) weight or weight gradient.
__global__ __launch_bounds__(kBlock) void fold_weight_kernel(const float* __restrict__ fold, const float* __restrict__ W, float* __restrict__ out,
                                                             int K, int64_t CN, int transpose) {
  // grid-stride: grid_1d caps the grid at 8192 workgroups (2 M threads), a (K, C, N) weight can be larger (K = 25, C = H f = 1536, N = 64)
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < CN; i += (int64_t)gridDim.x * kBlock)
    for (int j = 0; j < K; ++j) {
      float a = 0.f;
      for (int k = 0; k < K; ++k) a = fmaf(transpose ? fold[j * K + k] : fold[k * K + j], W[(int64_t)k * CN + i], a);
      out[(int64_t)j * CN + i] = a;
    }
}

// The three re-layouts of a (K, C, N) layer weight that the drivers take (tiny tensors; round 4 did them with torch permutes on the forward path):
//   kind 0  (C, K*N)   out[c, k*N + n] = W[k, c, n]    column blocks W_0 | ... | W_{K-1}: the ONE projection of the project-first form
//   kind 1  (K, N, C)  out[k, n, c]    = W[k, c, n]    W_k^T: the input gradient dx = sum_k T_k(L^T) g W_k^T run as a layer on (L^T, g, W^T)
//   kind 2  (N, K*C)   out[n, k*C + c] = W[k, c, n]    G = g [W_0^T | ... | W_{K-1}^T] for all K terms in one projection
__global__ __launch_bounds__(kBlock) void weight_layout_kernel(const float* __restrict__ W, float* __restrict__ out, int K, int C, int N, int kind) {
  const int64_t total = (int64_t)K * C * N;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {      // index into W: (k, c, n); grid-stride (grid_1d caps the grid)
    const int n = (int)(i % N), c = (int)((i / N) % C), k = (int)(i / ((int64_t)N * C));
    int64_t o;
    if (kind == 0) o = ((int64_t)c * K + k) * N + n;
    else if (kind == 1) o = ((int64_t)k * N + n) * C + c;
    else o = ((int64_t)n * K + k) * C + c;
    out[o] = W[i];
  }
}

// ---- backward of the streaming time-window projection (tgcn_cheb_project_windows_f32):
//   out[(s, w), i, :] = sum_k sum_h stack[k, s, i, w + h] W[k, h, :]          w in [0, T - H], stack: (K, S, n, T)
// dgrad: G[k, s, i, t] = sum_h sum_n g[(s, t - h), i, n] W[k, h, n]   (0 <= t - h <= T - H): what the hops on L^T then fold;
// one thread per (k, s, i, t), g rows read through the cache by the K threads that share them.
__global__ __launch_bounds__(kBlock) void windows_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ W, float* __restrict__ G,
                                                               int64_t S, int64_t n, int32_t T, int32_t H, int32_t N, int32_t K) {
  const int64_t total = (int64_t)K * S * n * T;
  const int nwin = T - H + 1;
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
    const int t = (int)(idx % T);
    const int64_t i = (idx / T) % n;
    const int64_t sidx = (idx / T / n) % S;
    const int k = (int)(idx / T / n / S);
    const int h0 = t - (nwin - 1) > 0 ? t - (nwin - 1) : 0, h1 = t < H - 1 ? t : H - 1;
    float acc = 0.f;
    for (int h = h0; h <= h1; ++h) {
      const float* gr = g + ((sidx * nwin + (t - h)) * n + i) * N;
      const float* wr = W + ((int64_t)k * H + h) * N;
      for (int c = 0; c < N; ++c) acc = fmaf(gr[c], wr[c], acc);
    }
    G[idx] = acc;
  }
}

// wgrad: dW[k, h, c] = sum_{s, w, i} stack[k, s, i, w + h] g[(s, w), i, c].  Block (k*H + h, column tile of 64, chunk of the
// (s, w, i) range): 64 column lanes x 4 row lanes, partial sums folded through LDS in lane order, the chunks by a second
// launch in chunk order (deterministic).
__global__ __launch_bounds__(kBlock) void windows_wgrad_partial_kernel(const float* __restrict__ stack, const float* __restrict__ g,
                                                                       float* __restrict__ partial, int64_t S, int64_t n, int32_t T,
                                                                       int32_t H, int32_t N, int32_t nchunks) {
  __shared__ float red[4][64];
  const int kh = blockIdx.x, k = kh / H, h = kh % H;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), ml = threadIdx.x >> 6;
  const int nwin = T - H + 1;
  const int64_t M = S * nwin * n, per = (M + nchunks - 1) / nchunks;
  const int64_t m0 = (int64_t)blockIdx.z * per, m1 = m0 + per < M ? m0 + per : M;
  float acc = 0.f;
  if (c < N)
    for (int64_t m = m0 + ml; m < m1; m += 4) {
      const int64_t i = m % n, w = (m / n) % nwin, sidx = m / n / nwin;
      acc = fmaf(stack[(((int64_t)k * S + sidx) * n + i) * T + w + h], g[m * N + c], acc);
    }
  red[ml][threadIdx.x & 63] = acc;
  __syncthreads();
  if (ml == 0 && c < N)
    partial[((int64_t)blockIdx.z * gridDim.x + kh) * N + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ __launch_bounds__(kBlock) void windows_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int64_t count, int32_t nchunks) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (int64_t)gridDim.x * kBlock) {
    float acc = 0.f;
    for (int z = 0; z < nchunks; ++z) acc += partial[(int64_t)z * count + i];
    dW[i] = acc;
  }
}
