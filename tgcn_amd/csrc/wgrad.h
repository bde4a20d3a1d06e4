// wgrad.h -- dW_t = A_t^T G (backward of the projection), two deterministic stages
// Part of the single translation unit tgcn_hip.hip (included once, inside its anonymous namespace).
#pragma once

// --------------------------------------------------------------------------------------------------
// weight gradient: dW[t][c][n] = sum_m A_t[m][c] * G[m][n]   (backward of the projection; fp32 MFMA)
// --------------------------------------------------------------------------------------------------
// Stage 1: block b sums rows [b*rows_per_block, ...) into partial[b]; wave w owns the 16-wide c tiles w, w+4, ...
// and, per tile, TG terms x all n tiles (<= 4) as MFMA accumulators (A^T and G fragments are read straight from
// global: lane (r, kq) reads row m0+kq, column c0+r).  Stage 2 folds the partials in block order: deterministic.
struct WgradParams {
  const float* a[kMaxTerms];
  int64_t lda[kMaxTerms];
  const float* G;
  float* partial;   // [nblocks][nterms*Kc][N]
  float* dW;        // [nterms*Kc][N]
  int64_t M, ldg, rows_per_block;
  int32_t Kc, N, nterms, nblocks;
};



constexpr int kWgTerms = 5;   // terms accumulated at once per wave (register budget: 5 * 4 tiles * 4 regs)

// One wave per (row block, 64-column tile of G, 16-row tile of the weight, group of kWgTerms terms): dW_t tile = A_t^T G over the block's rows
// on the fp32 MFMA (k = 4 rows per instruction), fragments straight from global memory, kWgUnroll steps of loads in
// flight.  Row blocks are small (>= 64 rows) so that a few thousand waves cover even the q*n ~ 50 k rows of the
// small-graph configs; the per-block partials are folded in block order by wgrad_reduce_kernel (deterministic).
constexpr int kWgUnroll = 4;
__global__ __launch_bounds__(64) void wgrad_partial_kernel(const WgradParams p) {
  const int lane = threadIdx.x;
  const int r = lane & 15, kq = lane >> 4;
  const int64_t m_lo = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t m_hi = min(p.M, m_lo + p.rows_per_block);
  const int n0 = blockIdx.y * 64;
  const int tgroups = (p.nterms + kWgTerms - 1) / kWgTerms;
  const int ct = blockIdx.z / tgroups;                     // 16-row tile of the weight
  const int tg = blockIdx.z % tgroups;                     // group of kWgTerms terms: its own wave, not a serial pass
  float* part = p.partial + (size_t)blockIdx.x * p.nterms * p.Kc * p.N;
  const int c = ct * 16 + r;
  {
    const int t0 = tg * kWgTerms;
    f32x4 acc[kWgTerms][4];
#pragma unroll
    for (int t = 0; t < kWgTerms; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t m0 = m_lo; m0 < m_hi; m0 += 4 * kWgUnroll) {
      float gv[kWgUnroll][4], av[kWgUnroll][kWgTerms];
#pragma unroll
      for (int u = 0; u < kWgUnroll; ++u) {
        const int64_t m = m0 + u * 4 + kq;
        const bool mok = m < m_hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + j * 16 + r;
          gv[u][j] = (mok && n < p.N) ? p.G[m * p.ldg + n] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < kWgTerms; ++t)
          av[u][t] = (mok && c < p.Kc && t0 + t < p.nterms) ? p.a[t0 + t][m * p.lda[t0 + t] + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kWgUnroll; ++u)
#pragma unroll
        for (int t = 0; t < kWgTerms; ++t) {
          if (t0 + t >= p.nterms) break;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][t], gv[u][j], acc[t][j], 0, 0, 0);
        }
    }
    // D layout: col = lane&15 (n within tile), row = (lane>>4)*4 + i (c within tile)
#pragma unroll
    for (int t = 0; t < kWgTerms; ++t) {
      if (t0 + t >= p.nterms) break;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cc = ct * 16 + kq * 4 + i, n = n0 + j * 16 + r;
          if (cc < p.Kc && n < p.N) part[((size_t)(t0 + t) * p.Kc + cc) * p.N + n] = acc[t][j][i];
        }
    }
  }
}

// Folds the per-block partials: workgroup = 64 consecutive elements of dW x 16 waves, wave w sums its contiguous
// share of the blocks (four interleaved chains, 256-byte coalesced reads), the 16 shares are combined through LDS in
// wave order -> the same association for every launch.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const WgradParams p) {
  __shared__ float red[16][64];
  const int64_t total = (int64_t)p.nterms * p.Kc * p.N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  const int per = (p.nblocks + 15) / 16;
  const int b0 = wave * per, b1 = min(p.nblocks, b0 + per);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (e < total) {
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += p.partial[(size_t)(b + u) * total + e];
    }
    for (; b < b1; ++b) s[0] += p.partial[(size_t)b * total + e];
  }
  red[wave][lane] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (wave == 0 && e < total) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][lane];
    p.dW[e] = t;
  }
}
