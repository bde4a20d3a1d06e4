// small_graph.h -- graphs that fit in LDS: whole layer / basis in one launch (sparse, first-layer and dense matrix-pipe forms)
// Part of the single translation unit tgcn_hip.hip (included once, inside its anonymous namespace).
#pragma once

// --------------------------------------------------------------------------------------------------
// small graphs: the whole layer in ONE launch, CSR and activations resident in LDS
// --------------------------------------------------------------------------------------------------
// Workgroup = (sample q, tile of NTC output channels).  The recursion runs on the OUTPUT side (n x NTC values in
// LDS instead of n x C x K hop tensors in HBM):
//   mode 0 (monomial-folded weight, Horner):  Y_j = X W_j + L Y_{j+1},                     out = Y_0 + bias
//   mode 1 (Chebyshev weight, Clenshaw):      b_k = X W_k + 2 L b_{k+1} - b_{k+2},         out = X W_0 + L b_1 - b_2 + bias
// Thread t owns vertex t (up to 1024 threads) and keeps its input row in registers (rows longer than 32 floats are
// re-read from global memory in 32-float pieces every step: 8 loads against 512 fmaf); X W_j is VALU fmaf,
// L . is a walk over the LDS-resident CSR reading neighbour rows of the previous buffer from LDS.
// W'_j[c][g] = sum_k fold[k][j] W[k][c][g] (reference_power -> monomial basis), k ascending; eight loads in flight per
// round trip instead of one (the weights come from L2: the serial form cost ~1 us per k and per step).
__device__ __forceinline__ float folded_weight(const float* __restrict__ fold, const float* __restrict__ W, int K, int j,
                                               int64_t stride_k, int64_t off) {
  float w = 0.f;
  int k = 0;
  for (; k + 8 <= K; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = W[(int64_t)(k + u) * stride_k + off];
#pragma unroll
    for (int u = 0; u < 8; ++u) w = fmaf(fold[(k + u) * K + j], v[u], w);
  }
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = k + u < K ? W[(int64_t)(k + u) * stride_k + off] : 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (k + u < K) w = fmaf(fold[(k + u) * K + j], v[u], w);
  return w;
}

constexpr int kSmallMaxN = 1024;  // one thread per vertex
constexpr int kSmallCMax = 128;   // longest input row; up to 32 floats of it live in registers at a time
inline int small_cpad(int C) { return C <= 4 ? 4 : (C <= 16 ? 16 : (C + 31) / 32 * 32); }   // rows of the LDS weight tile

struct SmallParams {
  const int32_t* rowptr;
  const tgcn_edge* ev;
  const float* Ld;     // dense n x n copy of the operand (small_dense_kernel), nullable
  const float* x;
  const float* W;      // (K, C, N)
  const float* fold;   // (K, K) or null: W'_j = sum_k fold[k][j] W_k applied while staging (mode 0)
  const float* bias;
  float* out;
  int32_t n, nnz, q, K, C, N, mode, bias_kind, dense, spw, npad;   // spw samples per workgroup, npad threads per sample
  int32_t relu, pool;      // fused epilogue: out = max over `pool` consecutive vertices of relu(layer output)
  uint8_t* pool_idx;       // (q, n/pool, N) arg-max offset for the backward (nullable)
};

template <int NTC, int CP>   // CP: floats of the input row held in registers (C <= CP, or CP == 32 and C in pieces)
__global__ __launch_bounds__(kSmallMaxN) void small_forward_kernel(const SmallParams p) {
  extern __shared__ __align__(16) float smem[];
  const int n = p.n, nnz = p.nnz, C = p.C;
  const int nthr = blockDim.x;
  // LDS carve-up (all offsets multiples of 4 floats)
  // graph region: CSR (entries + rowptr), or -- for dense small operands such as the 148-vertex DTI graph of
  // load/res -- the operand as a dense n x ldn matrix (ldn odd: a column read by all threads is conflict-free)
  const int ldn = n | 1;
  tgcn_edge* ev = reinterpret_cast<tgcn_edge*>(smem);                      // nnz (padded to even)
  int32_t* rowptr = reinterpret_cast<int32_t*>(smem + 2 * ((nnz + 1) / 2 * 2));
  float* Ld = smem;
  float* Wt = p.dense ? smem + (n * ldn + 3) / 4 * 4 : reinterpret_cast<float*>(rowptr) + (n + 1 + 3) / 4 * 4;   // CP x NTC
  const int nbuf = p.mode == 0 ? 2 : 3;
  const int tid = threadIdx.x;
  // the workgroup runs spw samples side by side (occupancy for small n); thread = (sample slot, vertex)
  const int slot = tid / p.npad, li = tid % p.npad;
  const int q = blockIdx.x * p.spw + slot, n0 = blockIdx.y * NTC;
  const bool live = q < p.q;
  const int cpad = (C + CP - 1) / CP * CP;                          // rows of the weight tile
  const bool pieces = C > CP;                                       // input row longer than the register copy
  float* Ybase = Wt + cpad * NTC + slot * (nbuf * n * NTC);         // this sample's NB buffers of n x NTC

  // ---- stage CSR and this sample's input (through the Y buffers, which are free now) into LDS / registers
  if (p.dense) {
    for (int e = tid; e < n * ldn; e += nthr) Ld[e] = 0.f;
    __syncthreads();
    if (tid < n)    // first n threads: one row each (own row only, no atomics)
      for (int e = p.rowptr[tid]; e < p.rowptr[tid + 1]; ++e) Ld[tid * ldn + p.ev[e].col] += p.ev[e].val;
  } else {
    for (int e = tid; e < nnz; e += nthr) ev[e] = p.ev[e];
    for (int i = tid; i <= n; i += nthr) rowptr[i] = p.rowptr[i];
  }
  float xr[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) xr[c] = 0.f;
  const float* xq = p.x + (int64_t)(live ? q : 0) * n * C;
  if (!pieces) {
    const int total = n * C, cap = nbuf * n * NTC;
    for (int base = 0; base < total; base += cap) {     // one piece unless C > nbuf*NTC
      const int cnt = min(cap, total - base);
      __syncthreads();
      if (live)
        for (int e = li; e < cnt; e += p.npad) Ybase[e] = xq[base + e];
      __syncthreads();
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        const int e = li * C + c - base;
        if (live && li < n && c < C && e >= 0 && e < cnt) xr[c] = Ybase[e];
      }
    }
  }
  __syncthreads();

  int cur = 0;   // buffer that receives this step's result
  for (int j = p.K - 1; j >= 0; --j) {
    // ---- weight tile of this step -> LDS (folding the reference_power basis on the fly when asked to)
    for (int e = tid; e < cpad * NTC; e += nthr) {     // rows c >= C and columns >= N are zero
      const int c = e / NTC, g = e % NTC;
      float w = 0.f;
      if (c < C && n0 + g < p.N) {
        if (p.fold) {
          w = folded_weight(p.fold, p.W, p.K, j, (int64_t)C * p.N, (int64_t)c * p.N + n0 + g);
        } else {
          w = p.W[((int64_t)j * C + c) * p.N + n0 + g];
        }
      }
      Wt[e] = w;
    }
    __syncthreads();
    const bool first = (j == p.K - 1);
    const float alpha = (p.mode == 1 && j > 0) ? 2.f : 1.f;
    const bool sub = (p.mode == 1) && (j <= p.K - 3);              // b_{k+2} exists
    const float* B1 = Ybase + ((cur + nbuf - 1) % nbuf) * n * NTC;  // previous result
    const float* B2 = Ybase + ((cur + nbuf - 2) % nbuf) * n * NTC;  // the one before (mode 1)
    float* Yn = Ybase + cur * n * NTC;
    const int i = li;
    if (live && i < n) {
      float acc[NTC];
#pragma unroll
      for (int g = 0; g < NTC; ++g) acc[g] = 0.f;
      if (!first && p.dense) {                        // alpha * (L B1)[i], dense operand: B1 rows are LDS broadcasts
        for (int col = 0; col < n; ++col) {
          const float lv = Ld[i * ldn + col];
          const float4* src = reinterpret_cast<const float4*>(B1 + col * NTC);
          const int sw = (col >> 2) & (NTC / 4 - 1);
#pragma unroll
          for (int g4 = 0; g4 < NTC / 4; ++g4) {
            const float4 y = src[g4 ^ sw];
            acc[g4 * 4 + 0] = fmaf(lv, y.x, acc[g4 * 4 + 0]);
            acc[g4 * 4 + 1] = fmaf(lv, y.y, acc[g4 * 4 + 1]);
            acc[g4 * 4 + 2] = fmaf(lv, y.z, acc[g4 * 4 + 2]);
            acc[g4 * 4 + 3] = fmaf(lv, y.w, acc[g4 * 4 + 3]);
          }
        }
      } else if (!first) {                            // alpha * (L B1)[i], CSR walk
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
          const tgcn_edge ed = ev[e];
          const float4* src = reinterpret_cast<const float4*>(B1 + ed.col * NTC);
          const int sw = (ed.col >> 2) & (NTC / 4 - 1);   // rows are stored with their 16-byte quads XOR-swizzled
#pragma unroll
          for (int g4 = 0; g4 < NTC / 4; ++g4) {
            const float4 y = src[g4 ^ sw];
            acc[g4 * 4 + 0] = fmaf(ed.val, y.x, acc[g4 * 4 + 0]);
            acc[g4 * 4 + 1] = fmaf(ed.val, y.y, acc[g4 * 4 + 1]);
            acc[g4 * 4 + 2] = fmaf(ed.val, y.z, acc[g4 * 4 + 2]);
            acc[g4 * 4 + 3] = fmaf(ed.val, y.w, acc[g4 * 4 + 3]);
          }
        }
      }
      if (!first) {
#pragma unroll
        for (int g = 0; g < NTC; ++g) acc[g] *= alpha;
        if (sub) {
          const int swi = (i >> 2) & (NTC / 4 - 1);
#pragma unroll
          for (int g4 = 0; g4 < NTC / 4; ++g4) {
            const float4 z = reinterpret_cast<const float4*>(B2 + i * NTC)[g4 ^ swi];
            acc[g4 * 4 + 0] -= z.x; acc[g4 * 4 + 1] -= z.y; acc[g4 * 4 + 2] -= z.z; acc[g4 * 4 + 3] -= z.w;
          }
        }
      }
      for (int cb = 0; cb < cpad; cb += CP) {          // + X W_j  (padded rows of Wt are zero: no per-c condition)
        if (pieces) {                                   // this piece of the own input row, straight from global / L2
          const float* xrow = xq + (int64_t)i * C + cb;
          if ((C & 3) == 0) {
#pragma unroll
            for (int c4 = 0; c4 < CP / 4; ++c4) {
              float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
              if (cb + c4 * 4 < C) v = reinterpret_cast<const float4*>(xrow)[c4];
              xr[c4 * 4] = v.x; xr[c4 * 4 + 1] = v.y; xr[c4 * 4 + 2] = v.z; xr[c4 * 4 + 3] = v.w;
            }
          } else {
#pragma unroll
            for (int c = 0; c < CP; ++c) xr[c] = cb + c < C ? xrow[c] : 0.f;
          }
        }
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const float xv = xr[c];
          const float4* wrow = reinterpret_cast<const float4*>(Wt + (cb + c) * NTC);
#pragma unroll
          for (int g4 = 0; g4 < NTC / 4; ++g4) {
            const float4 w = wrow[g4];
            acc[g4 * 4 + 0] = fmaf(xv, w.x, acc[g4 * 4 + 0]);
            acc[g4 * 4 + 1] = fmaf(xv, w.y, acc[g4 * 4 + 1]);
            acc[g4 * 4 + 2] = fmaf(xv, w.z, acc[g4 * 4 + 2]);
            acc[g4 * 4 + 3] = fmaf(xv, w.w, acc[g4 * 4 + 3]);
          }
          if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the unrolled LDS reads from piling up in registers
        }
      }
      if (j > 0) {
        const int swi = (i >> 2) & (NTC / 4 - 1);
#pragma unroll
        for (int g4 = 0; g4 < NTC / 4; ++g4)
          reinterpret_cast<float4*>(Yn + i * NTC)[g4 ^ swi] = make_float4(acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]);
      } else if (p.pool > 0) {                        // last step, pooled epilogue: biased row stays in LDS (plain layout)
        const float* bp = p.bias_kind == 1 ? p.bias + n0 : (p.bias_kind == 2 ? p.bias + (int64_t)i * p.N + n0 : nullptr);
#pragma unroll
        for (int g = 0; g < NTC; ++g) Yn[i * NTC + g] = acc[g] + ((bp && n0 + g < p.N) ? bp[g] : 0.f);
      } else {                                        // last step: bias and straight to HBM
        float* o = p.out + ((int64_t)q * n + i) * p.N + n0;
        const float* bp = p.bias_kind == 1 ? p.bias + n0 : (p.bias_kind == 2 ? p.bias + (int64_t)i * p.N + n0 : nullptr);
        if (n0 + NTC <= p.N && (p.N & 3) == 0) {      // whole tile, 16-byte stores
#pragma unroll
          for (int g4 = 0; g4 < NTC / 4; ++g4) {
            float4 v4 = make_float4(acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]);
            if (bp) { v4.x += bp[g4 * 4]; v4.y += bp[g4 * 4 + 1]; v4.z += bp[g4 * 4 + 2]; v4.w += bp[g4 * 4 + 3]; }
            reinterpret_cast<float4*>(o)[g4] = v4;
          }
        } else {
          for (int g = 0; g < NTC; ++g)               // ragged last tile: through LDS to keep register indices static
            Yn[i * NTC + g] = 0.f;
#pragma unroll
          for (int g = 0; g < NTC; ++g) Yn[i * NTC + g] = acc[g];
          for (int g = 0; g < NTC && n0 + g < p.N; ++g) o[g] = Yn[i * NTC + g] + (bp ? bp[g] : 0.f);
        }
      }
    }
    __syncthreads();
    cur = (cur + 1) % nbuf;
  }
  if (p.pool > 0 && live) {   // relu + max over `pool` consecutive vertices (gcn.py:246-255 after F.relu), from LDS
    const float* Yf = Ybase + ((cur + nbuf - 1) % nbuf) * n * NTC;
    const int np = n / p.pool;
    for (int e = li; e < np * NTC; e += p.npad) {
      const int ip = e / NTC, g = e % NTC;
      if (n0 + g >= p.N) continue;
      float best = Yf[(ip * p.pool) * NTC + g];
      int bi = 0;
      for (int jj = 1; jj < p.pool; ++jj) {
        const float v = Yf[(ip * p.pool + jj) * NTC + g];
        if (v > best || (v != v && best == best)) { best = v; bi = jj; }
      }
      if (p.relu) best = best > 0.f ? best : (best != best ? best : 0.f);
      const int64_t o = ((int64_t)q * np + ip) * p.N + n0 + g;
      p.out[o] = best;
      if (p.pool_idx) p.pool_idx[o] = (uint8_t)bi;
    }
  }
}

inline size_t small_lds_bytes(int n, int nnz, int C, int ntc, int mode, int dense, int spw = 1) {
  const size_t graph = dense ? (size_t)((n * (n | 1) + 3) / 4 * 4)
                             : 2 * (size_t)((nnz + 1) / 2 * 2) + (size_t)((n + 1 + 3) / 4 * 4);
  const size_t fl = graph + (size_t)small_cpad(C) * ntc + (size_t)spw * (mode == 0 ? 2 : 3) * n * ntc;
  return fl * sizeof(float);
}

// -> channel tile (16 / 8), *dense set to the cheaper LDS form of the operand; 0 when nothing fits
inline int small_config(int64_t n, int64_t nnz, int32_t C, int32_t mode, int* dense) {
  if (n < 1 || n > (int64_t)kSmallMaxN || nnz < 0 || nnz > (1 << 20) || C < 1 || C > kSmallCMax) return 0;
  if (mode != 0 && mode != 1) return 0;
  for (int ntc = 16; ntc >= 8; ntc /= 2) {
    const size_t sparse_b = small_lds_bytes((int)n, (int)nnz, C, ntc, mode, 0);
    const size_t dense_b = n <= 512 ? small_lds_bytes((int)n, (int)nnz, C, ntc, mode, 1) : (size_t)-1;
    const size_t best = sparse_b < dense_b ? sparse_b : dense_b;
    if (best <= 160 * 1024) {
      *dense = dense_b < sparse_b;
      return ntc;
    }
  }
  return 0;
}

// ---- first layers (C <= 4 input channels, typically 1): the recursion is cheaper on the INPUT side -- the hop tensors
// are 4 floats per vertex and stay in LDS, every step adds its term P_k W_k into NT output accumulators held in
// registers (thread = vertex), so one workgroup covers NT = 64 / 32 / 16 output channels with ONE recursion instead of
// one per 16-channel tile:  mode 0: P_k = L P_{k-1} (monomials, folded weight);  mode 1: T_k = 2 L T_{k-1} - T_{k-2}.
// Fused relu + pool epilogue through wave shuffles (the `pool` vertices of a group are neighbouring lanes).
template <int NT>
__global__ __launch_bounds__(kSmallMaxN) void small_narrow_kernel(const SmallParams p) {
  extern __shared__ __align__(16) float smem[];
  constexpr int CP = 4;
  const int n = p.n, nnz = p.nnz, C = p.C;
  const int nthr = blockDim.x, tid = threadIdx.x;
  const int ldn = n | 1;
  tgcn_edge* ev = reinterpret_cast<tgcn_edge*>(smem);
  int32_t* rowptr = reinterpret_cast<int32_t*>(smem + 2 * ((nnz + 1) / 2 * 2));
  float* Ld = smem;
  float* Wt = p.dense ? smem + (n * ldn + 3) / 4 * 4 : reinterpret_cast<float*>(rowptr) + (n + 1 + 3) / 4 * 4;   // CP x NT
  const int nbuf = p.mode == 0 ? 2 : 3;
  const int slot = tid / p.npad, i = tid % p.npad;
  const int q = blockIdx.x * p.spw + slot, n0 = blockIdx.y * NT;
  const bool live = q < p.q && i < n;
  float4* Pb = reinterpret_cast<float4*>(Wt + CP * NT) + slot * (nbuf * n);     // this sample's nbuf buffers of n float4
  if (p.dense) {
    for (int e = tid; e < n * ldn; e += nthr) Ld[e] = 0.f;
    __syncthreads();
    if (tid < n)
      for (int e = p.rowptr[tid]; e < p.rowptr[tid + 1]; ++e) Ld[tid * ldn + p.ev[e].col] += p.ev[e].val;
  } else {
    for (int e = tid; e < nnz; e += nthr) ev[e] = p.ev[e];
    for (int r = tid; r <= n; r += nthr) rowptr[r] = p.rowptr[r];
  }
  float4 pk = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float* xr = p.x + ((int64_t)q * n + i) * C;
    pk.x = xr[0];
    if (C > 1) pk.y = xr[1];
    if (C > 2) pk.z = xr[2];
    if (C > 3) pk.w = xr[3];
    Pb[i] = pk;
  }
  float acc[NT];
#pragma unroll
  for (int g = 0; g < NT; ++g) acc[g] = 0.f;
  int cur = 1;
  for (int k = 0; k < p.K; ++k) {
    __syncthreads();                                   // previous step's P is complete; the weight tile is free
    for (int e = tid; e < CP * NT; e += nthr) {        // W'_k tile (rows c >= C and columns >= N are zero)
      const int c = e / NT, g = e % NT;
      float w = 0.f;
      if (c < C && n0 + g < p.N) {
        if (p.fold) {
          w = folded_weight(p.fold, p.W, p.K, k, (int64_t)C * p.N, (int64_t)c * p.N + n0 + g);
        } else {
          w = p.W[((int64_t)k * C + c) * p.N + n0 + g];
        }
      }
      Wt[e] = w;
    }
    if (k > 0 && live) {
      const float4* B1 = Pb + ((cur + nbuf - 1) % nbuf) * n;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.dense) {
        for (int col = 0; col < n; ++col) {
          const float lv = Ld[i * ldn + col];
          const float4 y = B1[col];
          s.x = fmaf(lv, y.x, s.x); s.y = fmaf(lv, y.y, s.y); s.z = fmaf(lv, y.z, s.z); s.w = fmaf(lv, y.w, s.w);
        }
      } else {
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
          const tgcn_edge ed = ev[e];
          const float4 y = B1[ed.col];
          s.x = fmaf(ed.val, y.x, s.x); s.y = fmaf(ed.val, y.y, s.y); s.z = fmaf(ed.val, y.z, s.z); s.w = fmaf(ed.val, y.w, s.w);
        }
      }
      if (p.mode == 1 && k >= 2) {                     // one rounding, like 2*X - Xt[k-2] of the reference
        const float4 z = Pb[((cur + nbuf - 2) % nbuf) * n + i];
        s.x = fmaf(2.f, s.x, -z.x); s.y = fmaf(2.f, s.y, -z.y); s.z = fmaf(2.f, s.z, -z.z); s.w = fmaf(2.f, s.w, -z.w);
      }
      pk = s;
      if (k + 1 < p.K) Pb[cur * n + i] = pk;
    }
    __syncthreads();                                   // weight tile staged (and nobody still reads the buffer written next)
    if (live) {
      const float pc[CP] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        if (c >= C) break;
        const float4* wrow = reinterpret_cast<const float4*>(Wt + c * NT);
#pragma unroll
        for (int g4 = 0; g4 < NT / 4; ++g4) {
          const float4 w = wrow[g4];
          acc[g4 * 4 + 0] = fmaf(pc[c], w.x, acc[g4 * 4 + 0]);
          acc[g4 * 4 + 1] = fmaf(pc[c], w.y, acc[g4 * 4 + 1]);
          acc[g4 * 4 + 2] = fmaf(pc[c], w.z, acc[g4 * 4 + 2]);
          acc[g4 * 4 + 3] = fmaf(pc[c], w.w, acc[g4 * 4 + 3]);
        }
      }
    }
    if (k > 0) cur = (cur + 1) % nbuf;
  }
  // ---- epilogue: bias, optional relu + max over `pool` consecutive vertices (neighbouring lanes), store
  const float* bp = p.bias_kind == 1 ? p.bias + n0 : (p.bias_kind == 2 ? p.bias + (int64_t)(live ? i : 0) * p.N + n0 : nullptr);
  const bool vec = (p.N & 3) == 0;
  if (p.pool > 0) {
    const int np = n / p.pool;
    const bool writer = live && (i % p.pool) == 0;
    const int64_t obase = ((int64_t)q * np + i / p.pool) * p.N + n0;
#pragma unroll
    for (int g = 0; g < NT; ++g) {
      float v = acc[g] + ((bp && n0 + g < p.N) ? bp[g] : 0.f);
      float best = v;
      int bi = 0;
      for (int jj = 1; jj < p.pool; ++jj) {             // lanes i+1 .. i+pool-1 of the same wave (npad and 64 are multiples of pool's group)
        const float o = __shfl_down(v, jj, 64);
        if (o > best || (o != o && best == best)) { best = o; bi = jj; }
      }
      if (p.relu) best = best > 0.f ? best : (best != best ? best : 0.f);
      if (writer && n0 + g < p.N) {
        p.out[obase + g] = best;
        if (p.pool_idx) p.pool_idx[obase + g] = (uint8_t)bi;
      }
    }
    return;
  }
  if (!live) return;
  float* o = p.out + ((int64_t)q * n + i) * p.N + n0;
  if (vec) {
#pragma unroll
    for (int g4 = 0; g4 < NT / 4; ++g4) {
      if (n0 + g4 * 4 >= p.N) break;
      float4 v4 = make_float4(acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]);
      if (bp) { v4.x += bp[g4 * 4]; v4.y += bp[g4 * 4 + 1]; v4.z += bp[g4 * 4 + 2]; v4.w += bp[g4 * 4 + 3]; }
      reinterpret_cast<float4*>(o)[g4] = v4;
    }
  } else {
#pragma unroll
    for (int g = 0; g < NT; ++g)
      if (n0 + g < p.N) o[g] = acc[g] + (bp ? bp[g] : 0.f);
  }
}

inline size_t narrow_lds_bytes(int n, int nnz, int nt, int mode, int dense, int spw = 1) {
  const size_t graph = dense ? (size_t)((n * (n | 1) + 3) / 4 * 4)
                             : 2 * (size_t)((nnz + 1) / 2 * 2) + (size_t)((n + 1 + 3) / 4 * 4);
  return (graph + (size_t)4 * nt + (size_t)spw * (mode == 0 ? 2 : 3) * n * 4) * sizeof(float);
}

// ---- the basis of the layer for small graphs, for the weight gradient: terms k = 1 .. K-1 of
//   mode 0:  P_k = L P_{k-1}                      (monomials, the basis of the folded weight)
//   mode 1:  T_k = 2 L T_{k-1} - T_{k-2}          (T_1 = L x)
// written to stack (K, q, n, C) (term 0 is x itself and is not copied).  Same LDS-resident operand and thread = vertex
// layout as small_forward_kernel; workgroup = (spw samples, tile of CT input channels).
template <int CT>
__global__ __launch_bounds__(kSmallMaxN) void small_basis_kernel(const SmallParams p) {
  extern __shared__ __align__(16) float smem[];
  const int n = p.n, nnz = p.nnz, C = p.C;
  const int nthr = blockDim.x, tid = threadIdx.x;
  const int ldn = n | 1;
  tgcn_edge* ev = reinterpret_cast<tgcn_edge*>(smem);
  int32_t* rowptr = reinterpret_cast<int32_t*>(smem + 2 * ((nnz + 1) / 2 * 2));
  float* Ld = smem;
  float* Y0 = p.dense ? smem + (n * ldn + 3) / 4 * 4 : reinterpret_cast<float*>(rowptr) + (n + 1 + 3) / 4 * 4;
  const int nbuf = p.mode == 0 ? 2 : 3;
  const int slot = tid / p.npad, i = tid % p.npad;
  const int q = blockIdx.x * p.spw + slot, c0 = blockIdx.y * CT;
  const bool live = q < p.q && i < n;
  float* Ybase = Y0 + slot * (nbuf * n * CT);
  if (p.dense) {
    for (int e = tid; e < n * ldn; e += nthr) Ld[e] = 0.f;
    __syncthreads();
    if (tid < n)
      for (int e = p.rowptr[tid]; e < p.rowptr[tid + 1]; ++e) Ld[tid * ldn + p.ev[e].col] += p.ev[e].val;
  } else {
    for (int e = tid; e < nnz; e += nthr) ev[e] = p.ev[e];
    for (int r = tid; r <= n; r += nthr) rowptr[r] = p.rowptr[r];
  }
  const int swi = (i >> 2) & (CT / 4 - 1);       // rows are stored with their 16-byte quads XOR-swizzled
  if (live) {
    const float* xr = p.x + ((int64_t)q * n + i) * C + c0;
#pragma unroll
    for (int g4 = 0; g4 < CT / 4; ++g4) {
      float4 v;
      v.x = c0 + g4 * 4 + 0 < C ? xr[g4 * 4 + 0] : 0.f;
      v.y = c0 + g4 * 4 + 1 < C ? xr[g4 * 4 + 1] : 0.f;
      v.z = c0 + g4 * 4 + 2 < C ? xr[g4 * 4 + 2] : 0.f;
      v.w = c0 + g4 * 4 + 3 < C ? xr[g4 * 4 + 3] : 0.f;
      reinterpret_cast<float4*>(Ybase + i * CT)[g4 ^ swi] = v;
    }
  }
  __syncthreads();
  int cur = 1;   // buffer that receives this step's result; buffer 0 holds x
  for (int k = 1; k < p.K; ++k) {
    const float* B1 = Ybase + ((cur + nbuf - 1) % nbuf) * n * CT;
    const float* B2 = Ybase + ((cur + nbuf - 2) % nbuf) * n * CT;
    float* Yn = Ybase + cur * n * CT;
    if (live) {
      float acc[CT];
#pragma unroll
      for (int g = 0; g < CT; ++g) acc[g] = 0.f;
      if (p.dense) {
        for (int col = 0; col < n; ++col) {
          const float lv = Ld[i * ldn + col];
          const float4* src = reinterpret_cast<const float4*>(B1 + col * CT);
          const int sw = (col >> 2) & (CT / 4 - 1);
#pragma unroll
          for (int g4 = 0; g4 < CT / 4; ++g4) {
            const float4 y = src[g4 ^ sw];
            acc[g4 * 4 + 0] = fmaf(lv, y.x, acc[g4 * 4 + 0]);
            acc[g4 * 4 + 1] = fmaf(lv, y.y, acc[g4 * 4 + 1]);
            acc[g4 * 4 + 2] = fmaf(lv, y.z, acc[g4 * 4 + 2]);
            acc[g4 * 4 + 3] = fmaf(lv, y.w, acc[g4 * 4 + 3]);
          }
        }
      } else {
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
          const tgcn_edge ed = ev[e];
          const float4* src = reinterpret_cast<const float4*>(B1 + ed.col * CT);
          const int sw = (ed.col >> 2) & (CT / 4 - 1);
#pragma unroll
          for (int g4 = 0; g4 < CT / 4; ++g4) {
            const float4 y = src[g4 ^ sw];
            acc[g4 * 4 + 0] = fmaf(ed.val, y.x, acc[g4 * 4 + 0]);
            acc[g4 * 4 + 1] = fmaf(ed.val, y.y, acc[g4 * 4 + 1]);
            acc[g4 * 4 + 2] = fmaf(ed.val, y.z, acc[g4 * 4 + 2]);
            acc[g4 * 4 + 3] = fmaf(ed.val, y.w, acc[g4 * 4 + 3]);
          }
        }
      }
      if (p.mode == 1 && k >= 2) {                // one rounding, like 2*X - Xt[k-2] of the reference
#pragma unroll
        for (int g4 = 0; g4 < CT / 4; ++g4) {
          const float4 z = reinterpret_cast<const float4*>(B2 + i * CT)[g4 ^ swi];
          acc[g4 * 4 + 0] = fmaf(2.f, acc[g4 * 4 + 0], -z.x);
          acc[g4 * 4 + 1] = fmaf(2.f, acc[g4 * 4 + 1], -z.y);
          acc[g4 * 4 + 2] = fmaf(2.f, acc[g4 * 4 + 2], -z.z);
          acc[g4 * 4 + 3] = fmaf(2.f, acc[g4 * 4 + 3], -z.w);
        }
      }
      float* o = p.out + (((int64_t)k * p.q + q) * n + i) * C + c0;
      if (c0 + CT <= C && (C & 3) == 0) {
#pragma unroll
        for (int g4 = 0; g4 < CT / 4; ++g4)
          reinterpret_cast<float4*>(o)[g4] = make_float4(acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]);
      } else {
#pragma unroll
        for (int g = 0; g < CT; ++g)
          if (c0 + g < C) o[g] = acc[g];
      }
      if (k + 1 < p.K) {
#pragma unroll
        for (int g4 = 0; g4 < CT / 4; ++g4)
          reinterpret_cast<float4*>(Yn + i * CT)[g4 ^ swi] = make_float4(acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]);
      }
    }
    __syncthreads();
    cur = (cur + 1) % nbuf;
  }
}

inline size_t basis_lds_bytes(int n, int nnz, int ct, int mode, int dense, int spw = 1) {
  const size_t graph = dense ? (size_t)((n * (n | 1) + 3) / 4 * 4)
                             : 2 * (size_t)((nnz + 1) / 2 * 2) + (size_t)((n + 1 + 3) / 4 * 4);
  return (graph + (size_t)spw * (mode == 0 ? 2 : 3) * n * ct) * sizeof(float);
}

// -> channel tile (16 / 8 / 4) of small_basis_kernel, 0 when the operand does not fit
inline int basis_config(int64_t n, int64_t nnz, int32_t C, int32_t mode, int* dense) {
  if (n < 1 || n > (int64_t)kSmallMaxN || nnz < 0 || nnz > (1 << 20) || C < 1 || (mode != 0 && mode != 1)) return 0;
  for (int ct = (C <= 4 ? 4 : (C <= 8 ? 8 : 16)); ct >= 4; ct /= 2) {
    const size_t sparse_b = basis_lds_bytes((int)n, (int)nnz, ct, mode, 0);
    const size_t dense_b = n <= 512 ? basis_lds_bytes((int)n, (int)nnz, ct, mode, 1) : (size_t)-1;
    const size_t best = sparse_b < dense_b ? sparse_b : dense_b;
    if (best <= 160 * 1024) {
      *dense = dense_b < sparse_b;
      return ct;
    }
  }
  return 0;
}

// ---- small DENSE operands (the 148-parcel DTI graph of load/res: 34 % of the entries stored) on the fp32 matrix pipe.
// Same recursions as small_forward_kernel / small_basis_kernel, but L . Y is a dense (npad x npad) x (npad x S*16)
// product per step:  one wave per 16-row tile of L, whose A-fragments (npad/4 registers, read from the dense copy
// tgcn_csr.dense) stay in registers for the whole kernel; Y (S samples x 16 channels per workgroup) lives in LDS with a row stride of S*16+16 floats (the four
// k rows of a B-fragment read fall into different banks).  v_mfma_f32_16x16x4_f32: k-ordered fp32 fmaf chain.
//   A lane (r = lane&15, kq = lane>>4) = A[row r][k kq];  B = B[k kq][col r];  D[i] = D[row 4*kq+i][col r].
constexpr int kDenseMaxN = 256;      // vertices (16 row tiles -> 16 waves)
constexpr int kDenseMaxC = 32;       // input row length (X fragments in registers) ...
constexpr int kDenseMaxCSmall = 64;  // ... 64 for operands of up to 128 vertices (8 waves: 256 registers per lane)
constexpr int kDenseWFloats = 4096;  // LDS for weight tiles: all K of them when they fit (staged once), else one per step
// NW: most waves (16-row tiles) of a workgroup -> register budget and size of Lf; XKMAX: k-steps of the input row (C <= 4 XKMAX)
template <int S, bool BASIS, int NW, int XKMAX>
__global__ __launch_bounds__(NW * 64) void small_dense_kernel(const SmallParams p) {
  extern __shared__ __align__(16) float smem[];
  constexpr int LDY = S * 16 + 16;
  constexpr int NKMAX = NW * 4;
  constexpr int kDenseKB = 8 / S;                        // k-steps of B fragments per batch (8 LDS reads in flight)
  const int n = p.n, C = p.C;
  const int npad = (n + 15) / 16 * 16, nk = npad / 4;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int i0 = wave * 16;                              // this wave's row tile
  const int nbuf = p.mode == 0 ? 2 : 3;
  const int cpad = (C + 3) / 4 * 4, xk = cpad / 4;
  float* Wt = smem;                                      // (cpad, 16) weight tile of the step (not for BASIS)
  float* Ybase = smem + (BASIS ? 0 : kDenseWFloats);     // nbuf buffers of npad x LDY
  const bool w_all = !BASIS && p.K * cpad * 16 <= kDenseWFloats;
  const int q0 = blockIdx.x * S, n0 = blockIdx.y * 16;   // first sample; first output channel (BASIS: input channel)

  // ---- L fragments straight from the dense copy of the operand (L2-resident: every workgroup reads the same 4 n^2 bytes)
  float Lf[NKMAX];
#pragma unroll
  for (int kt = 0; kt < NKMAX; ++kt) {
    const int row = i0 + r, col = kt * 4 + kq;
    Lf[kt] = (kt < nk && row < n && col < n) ? p.Ld[(int64_t)row * n + col] : 0.f;
  }
  // ---- X fragments (forward: the wave's 16 input rows of every sample) / initial Y = x tile (basis)
  float Xf[BASIS ? 1 : S][BASIS ? 1 : XKMAX];
  if constexpr (!BASIS) {
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int kt = 0; kt < XKMAX; ++kt) {
        const int c = kt * 4 + kq, i = i0 + r;
        Xf[s][kt] = (q0 + s < p.q && i < n && c < C) ? p.x[((int64_t)(q0 + s) * n + i) * C + c] : 0.f;
      }
  } else {
    for (int e = tid; e < npad * S * 16; e += nthr) {
      const int i = e / (S * 16), sc = e % (S * 16), s = sc >> 4, c = n0 + (sc & 15);
      Ybase[i * LDY + sc] = (q0 + s < p.q && i < n && c < C) ? p.x[((int64_t)(q0 + s) * n + i) * C + c] : 0.f;
    }
  }
  __syncthreads();

  auto stage_w = [&](float* dst, int j) {                 // (cpad, 16) tile of W'_j: rows c >= C and columns >= N are zero
    for (int e = tid; e < cpad * 16; e += nthr) {
      const int c = e >> 4, g = e & 15;
      float w = 0.f;
      if (c < C && n0 + g < p.N) {
        if (p.fold) {
          w = folded_weight(p.fold, p.W, p.K, j, (int64_t)C * p.N, (int64_t)c * p.N + n0 + g);
        } else {
          w = p.W[((int64_t)j * C + c) * p.N + n0 + g];
        }
      }
      dst[e] = w;
    }
  };
  if constexpr (!BASIS) {
    if (w_all) {
      for (int j = 0; j < p.K; ++j) stage_w(Wt + j * cpad * 16, j);
      __syncthreads();
    }
  }
  int cur = BASIS ? 1 : 0;
  const int nsteps = BASIS ? p.K - 1 : p.K;
  for (int st = 0; st < nsteps; ++st) {
    const int j = BASIS ? st + 1 : p.K - 1 - st;         // basis: term being produced; forward: Horner / Clenshaw index
    if constexpr (!BASIS) {
      if (!w_all) {
        stage_w(Wt, j);                                   // weight tile of this step
        __syncthreads();
      }
    }
    const float* Wj = w_all ? Wt + j * cpad * 16 : Wt;
    const bool first = !BASIS && st == 0;
    const float alpha = BASIS ? ((p.mode == 1 && j >= 2) ? 2.f : 1.f) : ((p.mode == 1 && j > 0) ? 2.f : 1.f);
    const bool sub = p.mode == 1 && (BASIS ? j >= 2 : j <= p.K - 3);
    const float* B1 = Ybase + ((cur + nbuf - 1) % nbuf) * npad * LDY;
    const float* B2 = Ybase + ((cur + nbuf - 2) % nbuf) * npad * LDY;
    float* Yn = Ybase + cur * npad * LDY;
    f32x4 acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!first) {
      // fully unrolled (Lf stays in registers); B fragments are read kDenseKB k-steps ahead of the MFMAs that use them
      float bb[2][kDenseKB][S];
#pragma unroll
      for (int u = 0; u < kDenseKB; ++u)
#pragma unroll
        for (int s = 0; s < S; ++s) bb[0][u][s] = u < nk ? B1[(u * 4 + kq) * LDY + r + s * 16] : 0.f;
#pragma unroll
      for (int kt0 = 0; kt0 < NKMAX; kt0 += kDenseKB) {
        if (kt0 < nk) {
          const int b = (kt0 / kDenseKB) & 1;
          if (kt0 + kDenseKB < nk) {
#pragma unroll
            for (int u = 0; u < kDenseKB; ++u)
#pragma unroll
              for (int s = 0; s < S; ++s)
                bb[b ^ 1][u][s] = kt0 + kDenseKB + u < nk ? B1[((kt0 + kDenseKB + u) * 4 + kq) * LDY + r + s * 16] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < kDenseKB; ++u) {
            if (kt0 + u < nk) {
#pragma unroll
              for (int s = 0; s < S; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(Lf[kt0 + u], bb[b][u][s], acc[s], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = alpha * acc[s][i];
          if (sub) v = fmaf(alpha, acc[s][i], -B2[(i0 + kq * 4 + i) * LDY + s * 16 + r]);
          acc[s][i] = v;
        }
    }
    if constexpr (!BASIS) {
#pragma unroll
      for (int kt = 0; kt < XKMAX; ++kt) {
        if (kt < xk) {
          const float wv = Wj[(kt * 4 + kq) * 16 + r];
#pragma unroll
          for (int s = 0; s < S; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(Xf[s][kt], wv, acc[s], 0, 0, 0);
        }
      }
    }
    const bool last = st == nsteps - 1;
    if (!last) {
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) Yn[(i0 + kq * 4 + i) * LDY + s * 16 + r] = acc[s][i];
    }
    if (BASIS || last) {                                  // basis: every term goes out; forward: the last step + bias
      const int ch = n0 + r;
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i0 + kq * 4 + i;
          if (q0 + s >= p.q || row >= n) continue;
          if constexpr (BASIS) {
            if (ch < C) p.out[(((int64_t)j * p.q + q0 + s) * n + row) * C + ch] = acc[s][i];
          } else {
            if (ch < p.N) {
              float v = acc[s][i];
              if (p.bias_kind == 1) v += p.bias[ch];
              else if (p.bias_kind == 2) v += p.bias[(int64_t)row * p.N + ch];
              p.out[((int64_t)(q0 + s) * n + row) * p.N + ch] = v;
            }
          }
        }
    }
    __syncthreads();
    cur = (cur + 1) % nbuf;
  }
}

// ---- the same dense recursion with L . Y on the bf16 matrix pipe, fp32-accurate through the three-way split of project.h
// (a = a1 + a2 + a3 exactly, six v_mfma_f32_16x16x32_bf16 per 32 k instead of eight fp32 MFMAs of twice the cycles).
// L is split once: a wave keeps the three planes of its 16 x npad strip in registers (12 per 32 k).  Y lives in LDS
// TRANSPOSED and already split -- YT[plane][column (S*16)][k = vertex] in bf16 -- so that a B fragment (8 consecutive
// k of one column) is one ds_read_b128 and a wave's result (4 consecutive vertices of one column per lane) is one
// ds_write_b64 per plane; rows are padded by 8 elements so that the 16 columns of a read fall on different banks.
// The fp32 value of a previous step (Clenshaw's b_{k+2}) is the exact sum of its three planes.  X W_j stays on the fp32
// MFMA (a small part of the work).
template <int S, bool BASIS, int NW, int XKMAX>
__global__ __launch_bounds__(NW * 64) void small_dense_x3_kernel(const SmallParams p) {
  extern __shared__ __align__(16) float smem[];
  constexpr int NKT = NW / 2 + (NW & 1);                 // 32-k tiles of L (npad <= 16 NW)
  const int n = p.n, C = p.C;
  const int npad = (n + 15) / 16 * 16;
  const int kpad = (npad + 31) / 32 * 32, nkt = kpad / 32;
  const int ldk = kpad + 8;                              // bf16 elements per YT row
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int i0 = wave * 16;
  const int nbuf = p.mode == 0 ? 2 : 3;
  const int cpad = (C + 3) / 4 * 4, xk = cpad / 4;
  float* Wt = smem;
  unsigned short* Yb = reinterpret_cast<unsigned short*>(smem + (BASIS ? 0 : kDenseWFloats));   // nbuf x [3][S*16][ldk]
  const int plane = S * 16 * ldk, buf = 3 * plane;
  const bool w_all = !BASIS && p.K * cpad * 16 <= kDenseWFloats;
  const int q0 = blockIdx.x * S, n0 = blockIdx.y * 16;

  // ---- L planes: lane (r, kq) holds L[i0 + r][32 t + 8 kq + j], j = 0..7, split three ways
  bf16x8 La[NKT][3];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = i0 + r, col = t * 32 + kq * 8 + j;
      v[j] = (t < nkt && row < n && col < n) ? p.Ld[(int64_t)row * n + col] : 0.f;
    }
    unsigned pl[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3(v[2 * j], v[2 * j + 1], pl[0][j], pl[1][j], pl[2][j]);
#pragma unroll
    for (int q3 = 0; q3 < 3; ++q3) {
      using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
      La[t][q3] = __builtin_bit_cast(bf16x8, u32x4{pl[q3][0], pl[q3][1], pl[q3][2], pl[q3][3]});
    }
  }
  // zero every Y buffer once: k padding (vertices >= npad) must read as zero
  for (int e = tid; e < nbuf * buf / 2; e += nthr) reinterpret_cast<unsigned*>(Yb)[e] = 0u;
  float Xf[BASIS ? 1 : S][BASIS ? 1 : XKMAX];
  if constexpr (!BASIS) {
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int kt = 0; kt < XKMAX; ++kt) {
        const int c = kt * 4 + kq, i = i0 + r;
        Xf[s][kt] = (q0 + s < p.q && i < n && c < C) ? p.x[((int64_t)(q0 + s) * n + i) * C + c] : 0.f;
      }
  }
  __syncthreads();
  // write 4 consecutive vertices (i4 .. i4+3) of column sc into buffer b, split into the three planes
  auto put4 = [&](int b, int sc, int i4, float v0, float v1, float v2, float v3) {
    unsigned a1, a2, a3, b1, b2, b3;
    split3(v0, v1, a1, a2, a3);
    split3(v2, v3, b1, b2, b3);
    unsigned short* d = Yb + b * buf + sc * ldk + i4;
    *reinterpret_cast<uint2*>(d) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(d + plane) = make_uint2(a2, b2);
    *reinterpret_cast<uint2*>(d + 2 * plane) = make_uint2(a3, b3);
  };
  auto bf2f = [](unsigned short h) { return __uint_as_float((unsigned)h << 16); };
  if constexpr (BASIS) {                                  // buffer 0 = x tile
    for (int e = tid; e < (npad / 4) * S * 16; e += nthr) {
      const int sc = e % (S * 16), i4 = (e / (S * 16)) * 4, s = sc >> 4, c = n0 + (sc & 15);
      float v[4];
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
        v[ii] = (q0 + s < p.q && i4 + ii < n && c < C) ? p.x[((int64_t)(q0 + s) * n + i4 + ii) * C + c] : 0.f;
      put4(0, sc, i4, v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
  }
  auto stage_w = [&](float* dst, int j) {
    for (int e = tid; e < cpad * 16; e += nthr) {
      const int c = e >> 4, g = e & 15;
      float w = 0.f;
      if (c < C && n0 + g < p.N) {
        if (p.fold) w = folded_weight(p.fold, p.W, p.K, j, (int64_t)C * p.N, (int64_t)c * p.N + n0 + g);
        else w = p.W[((int64_t)j * C + c) * p.N + n0 + g];
      }
      dst[e] = w;
    }
  };
  if constexpr (!BASIS) {
    if (w_all) {
      for (int j = 0; j < p.K; ++j) stage_w(Wt + j * cpad * 16, j);
      __syncthreads();
    }
  }
  int cur = BASIS ? 1 : 0;
  const int nsteps = BASIS ? p.K - 1 : p.K;
  for (int st = 0; st < nsteps; ++st) {
    const int j = BASIS ? st + 1 : p.K - 1 - st;
    if constexpr (!BASIS) {
      if (!w_all) {
        stage_w(Wt, j);
        __syncthreads();
      }
    }
    const float* Wj = w_all ? Wt + j * cpad * 16 : Wt;
    const bool first = !BASIS && st == 0;
    const float alpha = BASIS ? ((p.mode == 1 && j >= 2) ? 2.f : 1.f) : ((p.mode == 1 && j > 0) ? 2.f : 1.f);
    const bool sub = p.mode == 1 && (BASIS ? j >= 2 : j <= p.K - 3);
    const int b1 = (cur + nbuf - 1) % nbuf, b2 = (cur + nbuf - 2) % nbuf;
    f32x4 acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!first) {
      const unsigned short* Y1 = Yb + b1 * buf + r * ldk + kq * 8;     // + s*16*ldk + plane*q3 + 32 t
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        if (t < nkt) {
#pragma unroll
          for (int s = 0; s < S; ++s) {
            bf16x8 w[3];
#pragma unroll
            for (int q3 = 0; q3 < 3; ++q3) w[q3] = *reinterpret_cast<const bf16x8*>(Y1 + s * 16 * ldk + q3 * plane + t * 32);
            f32x4 c = acc[s];     // smallest terms first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(La[t][2], w[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(La[t][1], w[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(La[t][0], w[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(La[t][1], w[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(La[t][0], w[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(La[t][0], w[0], c, 0, 0, 0);
            acc[s] = c;
          }
        }
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (sub) {                                          // b_{k+2}: exact sum of its three planes
          const unsigned short* d = Yb + b2 * buf + (s * 16 + r) * ldk + i0 + kq * 4;
#pragma unroll
          for (int q3 = 0; q3 < 3; ++q3) {
            const uint2 h = *reinterpret_cast<const uint2*>(d + q3 * plane);
            z[0] += bf2f((unsigned short)(h.x & 0xFFFF)); z[1] += bf2f((unsigned short)(h.x >> 16));
            z[2] += bf2f((unsigned short)(h.y & 0xFFFF)); z[3] += bf2f((unsigned short)(h.y >> 16));
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[s][i] = sub ? fmaf(alpha, acc[s][i], -z[i]) : alpha * acc[s][i];
      }
    }
    if constexpr (!BASIS) {
#pragma unroll
      for (int kt = 0; kt < XKMAX; ++kt) {
        if (kt < xk) {
          const float wv = Wj[(kt * 4 + kq) * 16 + r];
#pragma unroll
          for (int s = 0; s < S; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(Xf[s][kt], wv, acc[s], 0, 0, 0);
        }
      }
    }
    const bool last = st == nsteps - 1;
    if (!last) {
#pragma unroll
      for (int s = 0; s < S; ++s) put4(cur, s * 16 + r, i0 + kq * 4, acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
    }
    if (BASIS || last) {
      const int ch = n0 + r;
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i0 + kq * 4 + i;
          if (q0 + s >= p.q || row >= n) continue;
          if constexpr (BASIS) {
            if (ch < C) p.out[(((int64_t)j * p.q + q0 + s) * n + row) * C + ch] = acc[s][i];
          } else {
            if (ch < p.N) {
              float v = acc[s][i];
              if (p.bias_kind == 1) v += p.bias[ch];
              else if (p.bias_kind == 2) v += p.bias[(int64_t)row * p.N + ch];
              p.out[((int64_t)(q0 + s) * n + row) * p.N + ch] = v;
            }
          }
        }
    }
    __syncthreads();
    cur = (cur + 1) % nbuf;
  }
}

// samples per workgroup (4 / 2 / 1) of small_dense_kernel, 0 when the shape is not for it
inline int dense_mfma_config(int64_t n, int64_t nnz, int32_t C, int32_t mode, int64_t q, int64_t col_tiles, bool basis) {
  if (n < 16 || n > kDenseMaxN || (mode != 0 && mode != 1)) return 0;
  if (!basis && C > (n <= 128 ? kDenseMaxCSmall : kDenseMaxC)) return 0;
  if (nnz * 4 < n * n) return 0;                       // at least a quarter of the entries stored: dense arithmetic pays
  const int npad = (int)(n + 15) / 16 * 16;
  const int nbuf = mode == 0 ? 2 : 3;
  const int nw = npad / 16;            // register budget per lane shrinks with the wave count: fewer samples (accumulators)
  const int smax = basis ? (nw > 12 ? 2 : 4) : (nw > 12 ? 1 : (nw > 10 ? 2 : 4));
  for (int S = smax; S >= 1; S /= 2) {
    const size_t fl = (size_t)(basis ? 0 : kDenseWFloats) + (size_t)nbuf * npad * (S * 16 + 16);
    if (fl * sizeof(float) > 160 * 1024) continue;
    if (S > 1 && (q + S - 1) / S * col_tiles < 192) continue;                         // keep most CUs busy
    return S;
  }
  return 0;
}

template <bool BASIS>
inline void launch_small_dense(hipStream_t st, const SmallParams& p, int S, int64_t col_tiles) {
  const int npad = (p.n + 15) / 16 * 16;
  const int nbuf = p.mode == 0 ? 2 : 3;
  const size_t lds = ((size_t)(BASIS ? 0 : kDenseWFloats) + (size_t)nbuf * npad * (S * 16 + 16)) * sizeof(float);
  const dim3 grid((unsigned)((p.q + S - 1) / S), (unsigned)col_tiles);
  const dim3 block((unsigned)(npad / 16 * 64));
#define TGCN_DENSE(SV, NWV, XKV)                                                                   \
  {                                                                                                \
    allow_large_lds((const void*)small_dense_kernel<SV, BASIS, NWV, XKV>, 160 * 1024);             \
    hipLaunchKernelGGL((small_dense_kernel<SV, BASIS, NWV, XKV>), grid, block, lds, st, p);        \
  }
#define TGCN_DENSE_X(SV, NWV)                                                      \
  if (BASIS || p.C <= 16) TGCN_DENSE(SV, NWV, 4)                                   \
  else if (p.C <= kDenseMaxC) TGCN_DENSE(SV, NWV, kDenseMaxC / 4)                  \
  else if constexpr (NWV == 8) TGCN_DENSE(SV, NWV, kDenseMaxCSmall / 4)
#define TGCN_DENSE_S(NWV) \
  if (S == 4) { TGCN_DENSE_X(4, NWV) } else if (S == 2) { TGCN_DENSE_X(2, NWV) } else { TGCN_DENSE_X(1, NWV) }
  const int nw = npad / 16;
  // 9-10 waves: the forward kernel gets its own register budget (4 samples per workgroup fit); the basis kernel measured
  // faster under the 12-wave bound
  if (nw <= 8) { TGCN_DENSE_S(8) } else if (nw <= 10 && !BASIS) { TGCN_DENSE_S(10) } else if (nw <= 12) { TGCN_DENSE_S(12) } else { TGCN_DENSE_S(16) }
#undef TGCN_DENSE_S
#undef TGCN_DENSE_X
#undef TGCN_DENSE
}

inline size_t dense_x3_lds_bytes(int n, int S, int mode, bool basis) {
  const int npad = (n + 15) / 16 * 16, kpad = (npad + 31) / 32 * 32;
  const size_t ybytes = (size_t)(mode == 0 ? 2 : 3) * 3 * S * 16 * (kpad + 8) * sizeof(unsigned short);
  return (basis ? 0 : (size_t)kDenseWFloats * sizeof(float)) + ybytes;
}

// samples per workgroup (4 / 2) of small_dense_x3_kernel, 0: use the fp32 form
inline int dense_x3_config(int64_t n, int64_t nnz, int32_t C, int32_t mode, int64_t q, int64_t col_tiles, bool basis) {
  if (!dense_mfma_config(n, nnz, C, mode, q, col_tiles, basis)) return 0;
  const int nw = ((int)n + 15) / 16;
  if (nw > 12) return 0;                                     // 13-16 waves: 128 registers per lane do not hold the L planes
  for (int S = ((nw > 10 || (basis && nw > 8)) ? 2 : 4); S >= 2; S /= 2) {
    if (dense_x3_lds_bytes((int)n, S, mode, basis) > 160 * 1024) continue;
    if ((q + S - 1) / S * col_tiles < 192) continue;       // keep most CUs busy
    return S;
  }
  return 0;
}

template <bool BASIS>
inline void launch_small_dense_x3(hipStream_t st, const SmallParams& p, int S, int64_t col_tiles) {
  const int npad = (p.n + 15) / 16 * 16;
  const size_t lds = dense_x3_lds_bytes(p.n, S, p.mode, BASIS);
  const dim3 grid((unsigned)((p.q + S - 1) / S), (unsigned)col_tiles);
  const dim3 block((unsigned)(npad / 16 * 64));
#define TGCN_DX3(SV, NWV, XKV)                                                                       \
  {                                                                                                  \
    allow_large_lds((const void*)small_dense_x3_kernel<SV, BASIS, NWV, XKV>, 160 * 1024);            \
    hipLaunchKernelGGL((small_dense_x3_kernel<SV, BASIS, NWV, XKV>), grid, block, lds, st, p);       \
  }
#define TGCN_DX3_X(SV, NWV)                                                          \
  if (BASIS || p.C <= 16) TGCN_DX3(SV, NWV, 4)                                       \
  else if (p.C <= kDenseMaxC) TGCN_DX3(SV, NWV, kDenseMaxC / 4)                      \
  else if constexpr (NWV == 8) TGCN_DX3(SV, NWV, kDenseMaxCSmall / 4)
#define TGCN_DX3_S(NWV) \
  if (S == 4) { TGCN_DX3_X(4, NWV) } else { TGCN_DX3_X(2, NWV) }
  const int nw = npad / 16;
  if (nw <= 8) { TGCN_DX3_S(8) } else if (nw <= 10 && !BASIS) { TGCN_DX3_S(10) } else if (nw <= 12) { TGCN_DX3_S(12) } else { TGCN_DX3_S(16) }
#undef TGCN_DX3_S
#undef TGCN_DX3_X
#undef TGCN_DX3
}
