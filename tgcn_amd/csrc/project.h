// project.h -- out = sum_t A_t W_t + bias: exact-fp32 MFMA kernels, bf16x3 kernels, the vector-ALU kernel for narrow contractions
// Part of the single translation unit tgcn_hip.hip (included once, inside its anonymous namespace).
#pragma once

// --------------------------------------------------------------------------------------------------
// projection (fp32 MFMA)
// --------------------------------------------------------------------------------------------------
constexpr int kMaxTerms = 32;

struct ProjParams {
  const float* a[kMaxTerms];
  int64_t lda[kMaxTerms];
  const float* W;
  const float* bias;
  float* out;
  int64_t M, ldo, n_vertices, interleave;
  int32_t Kc, N, nterms, bias_kind, accumulate, vec_epilogue;
  int32_t bias_ld, bias_cols;   // bias row length and number of leading output columns that receive it
  int32_t win_n, win_t;   // > 0: row m of A_t is the window starting at A_t[(m / win_n) * win_t + (m % win_n)]
  // Row map (compacted operands: the hop tensors hold only the vertices that have stored entries): tile row m is the
  // caller's vertex rowmap[m] -- its output row, its bias row, and its row in every term whose bit is set in `mapped`
  // (term 0 = x in the caller's labels); the other terms are read at row m.  With bit 31 of `mapped` set (kProjMapTermsOnly) the
  // map applies to the flagged terms ONLY: output and bias rows are the tile rows themselves (tiles over all vertices in order, the
  // compact hop tensors read through the vertex -> compact id map, empty vertices pointing at the zero row).
  const int32_t* rowmap;
  uint32_t mapped;
  // Batch of samples sharing the tile rows (project_x3_kernel<NT, true> only; the host loops for the other kernels): sample b
  // reads term t at a[t] + b * a_bs[t] and writes out + b * out_bs.  Consecutive workgroups take the SAME tile for the nbatch
  // samples, so they run at about the same time and the tile's per-vertex bias rows reach HBM once -- the other samples find
  // them in the Infinity Cache (cfg5: 2.56 GB of bias per time step otherwise).  Keeping the bias tile in registers across an
  // in-kernel sample loop instead cost 32 VGPRs and half the occupancy (projection 45 -> 71 ms per forward).
  int32_t nbatch;
  int64_t a_bs[kMaxTerms];
  int64_t out_bs;
  // Fused epilogue of the callers' x = gcn_pool_4(F.relu(layer(x))) (tgcn/nn/gcn.py:246-255, examples/pytorch_based/
  // pytorch_hcp_tgcn.py:134-141): pool > 1 stores max over `pool` consecutive tile rows (vertices of one sample) of relu(result +
  // bias) at out row m / pool, so the (q, n, N) layer output is never written; pool_idx (nullable) gets the arg-max offset
  // for the backward.  Kernels with the vector epilogue only (project_x3_kernel NT <= 4, project_resident_kernel); no row map,
  // no interleave, no accumulate; pool divides 16.
  int32_t pool;
  uint8_t* pool_idx;
  // Fused LAST HOP (compacted forward, project_x3_gather_kernel): term g_term of tile row m is not read from memory but gathered --
  // S[m] = sum_e val_e * g_X[col_e] over the stored entries [g_rowptr[m], g_rowptr[m + 1]) of compact row m, summed in stored order with
  // fmaf exactly like hop_kernel's row blocks -- when the row has at most g_thresh entries; longer rows read a[g_term] as usual (the
  // segment path of the hop kernel wrote them).  The hop tensor of the last hop is then neither written nor read for ~95 % of the rows.
  const int32_t* g_rowptr;
  const tgcn_edge* g_edges;
  const float* g_X;
  int64_t g_xbs;
  int32_t g_term, g_thresh;
};

// relu + max over p.pool consecutive rows of a wave's finished tile rows held in LDS scratch (`rows` rows of `stride` floats, NW
// columns from n0), first tile row = vertex row `mbase` (a multiple of p.pool): what relu_pool_kernel computes from the stored
// layer output (first maximum wins, NaN propagates), without storing it.
__device__ __forceinline__ void pooled_store(const ProjParams& p, const float* my, int stride, int rows, int64_t mbase, int n0, int NW,
                                             int lane) {
  const int segs = NW >> 2, groups = rows / p.pool;
  for (int idx = lane; idx < groups * segs; idx += 64) {
    const int pr = idx / segs, seg = (idx % segs) * 4;
    const int64_t m = mbase + (int64_t)pr * p.pool;
    const int col = n0 + seg;
    if (m >= p.M || col >= p.N) continue;          // M is a multiple of pool: a group is inside or outside as a whole
    float best[4];
    int bi[4] = {0, 0, 0, 0};
    for (int j = 0; j < p.pool; ++j) {
      const float* src = my + (pr * p.pool + j) * stride + seg;
      float v[4] = {src[0], src[1], src[2], src[3]};
      if (p.bias_kind && col < p.bias_cols) {
        const int64_t vert = (m + j) % p.n_vertices;
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + (p.bias_kind == 2 ? vert * p.bias_ld : 0) + col);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (j == 0 || v[c] > best[c] || (v[c] != v[c] && best[c] == best[c])) { best[c] = v[c]; bi[c] = j; }
    }
    float4 o;
    o.x = best[0] > 0.f ? best[0] : (best[0] != best[0] ? best[0] : 0.f);
    o.y = best[1] > 0.f ? best[1] : (best[1] != best[1] ? best[1] : 0.f);
    o.z = best[2] > 0.f ? best[2] : (best[2] != best[2] ? best[2] : 0.f);
    o.w = best[3] > 0.f ? best[3] : (best[3] != best[3] ? best[3] : 0.f);
    const int64_t orow = m / p.pool;
    *reinterpret_cast<float4*>(p.out + orow * p.ldo + col) = o;
    if (p.pool_idx)
      *reinterpret_cast<uint32_t*>(p.pool_idx + orow * p.ldo + col) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
  }
}

constexpr uint32_t kProjMapTermsOnly = 0x80000000u;

// row of term `term` that tile row m reads
__device__ __forceinline__ int64_t proj_arow(const ProjParams& p, int term, int64_t m) {
  return (p.rowmap && ((p.mapped >> term) & 1u)) ? (int64_t)p.rowmap[m] : m;
}
// output row of tile row m: the row map, or the layout-1 interleave (vertex-major tile rows -> sample-major output)
__device__ __forceinline__ int64_t proj_orow(const ProjParams& p, int64_t m) {
  if (p.rowmap && !(p.mapped & kProjMapTermsOnly)) return (int64_t)p.rowmap[m];
  return (p.interleave == 1) ? m : (m % p.interleave) * p.n_vertices + m / p.interleave;
}
// Row map AND interleave together (vertex-major operands of a compacted layer: tile row m = (mapped vertex m / interleave, sample
// m % interleave); the map applies to the vertex).  Only project_narrow_kernel takes this form: the 64-bit divisions cost the MFMA kernels
// registers they do not have (project_x3_kernel<4, true>: 124 -> 138 VGPRs = one workgroup per CU instead of two, measured 44 -> 67 ms).
__device__ __forceinline__ int64_t proj_arow_il(const ProjParams& p, int term, int64_t m) {
  if (!(p.rowmap && ((p.mapped >> term) & 1u))) return m;
  return (p.interleave == 1) ? (int64_t)p.rowmap[m] : (int64_t)p.rowmap[m / p.interleave] * p.interleave + m % p.interleave;
}
__device__ __forceinline__ int64_t proj_orow_il(const ProjParams& p, int64_t m) {
  const bool mapped = p.rowmap && !(p.mapped & kProjMapTermsOnly);
  if (p.interleave == 1) return mapped ? (int64_t)p.rowmap[m] : m;
  const int64_t v = m / p.interleave;
  return (m % p.interleave) * p.n_vertices + (mapped ? (int64_t)p.rowmap[v] : v);
}

// float offset of row m of a term: plain row stride, or overlapping time windows of a (vertex, T) series
__device__ __forceinline__ int64_t proj_row_off(const ProjParams& p, int64_t m, int64_t lda) {
  return p.win_n > 0 ? (m / p.win_n) * p.win_t + (m % p.win_n) : m * lda;
}

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Streaming-W kernel: block = 4 waves, 128 output rows; wave w owns rows [32w,32w+32) x NT*16 columns as 2*NT accumulators of
// v_mfma_f32_16x16x4_f32 (A[l&15][k=l>>4], B[k=l>>4][l&15], D col=l&15,row=(l>>4)*4+reg).
// LDS strides: As 34 (== 2 mod 32) and Ws == 16 mod 32 make both fragment reads conflict-free.
template <int NT, bool VEC4>
__global__ __launch_bounds__(kBlock) void project_kernel(const ProjParams p) {
  constexpr int BM = 128, KT = 32, AS = KT + 2;   // 4 waves x 32 rows; wave = two 16-row MFMA tiles sharing B fragments
  constexpr int NW = NT * 16;
  constexpr int NS = (NW % 32 == 0) ? NW + 16 : NW;
  constexpr int WREG = (KT * NW) / kBlock;
  __shared__ float As[BM * AS];
  __shared__ float Ws[KT * NS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * NW;
  f32x4 acc[2][NT];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ktiles = (p.Kc + KT - 1) / KT;
  const int total = p.nterms * ktiles;

  // software pipeline: tile t+1 travels global -> registers while tile t is multiplied out of LDS
  float ra[16], rw[WREG];
  auto load_tile = [&](int ti) {
    const int term = ti / ktiles, k0 = (ti % ktiles) * KT;
    const float* __restrict__ A = p.a[term];
    const int64_t lda = p.lda[term];
    const float* __restrict__ Wt = p.W + (int64_t)term * p.Kc * p.N;
    if constexpr (VEC4) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = (tid >> 3) + h * 32, kk = (tid & 7) * 4;
        const bool ok = (m0 + row < p.M) && (k0 + kk < p.Kc);
        const int64_t rr = (m0 + row < p.M) ? m0 + row : p.M - 1;
        const int kc = (k0 + kk < p.Kc) ? k0 + kk : 0;
        const float4 v = *reinterpret_cast<const float4*>(A + proj_arow(p, term, rr) * lda + kc);
        ra[h * 4 + 0] = ok ? v.x : 0.f; ra[h * 4 + 1] = ok ? v.y : 0.f; ra[h * 4 + 2] = ok ? v.z : 0.f; ra[h * 4 + 3] = ok ? v.w : 0.f;
      }
    } else {
#pragma unroll
      for (int h = 0; h < 16; ++h) {
        const int row = (tid >> 5) + h * 8, kk = tid & 31;
        const bool ok = (m0 + row < p.M) && (k0 + kk < p.Kc);
        const int64_t rr = (m0 + row < p.M) ? m0 + row : p.M - 1;
        const int kc = (k0 + kk < p.Kc) ? k0 + kk : 0;
        const float v = A[proj_row_off(p, proj_arow(p, term, rr), lda) + kc];
        ra[h] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int h = 0; h < WREG; ++h) {
      const int idx = tid + h * kBlock;
      const int kk = idx / NW, cc = idx % NW;
      const bool ok = (k0 + kk < p.Kc) && (n0 + cc < p.N);
      const float v = Wt[(int64_t)(ok ? k0 + kk : 0) * p.N + (ok ? n0 + cc : 0)];
      rw[h] = ok ? v : 0.f;
    }
  };
  auto store_tile = [&]() {
    if constexpr (VEC4) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = (tid >> 3) + h * 32, kk = (tid & 7) * 4;
        float2* d = reinterpret_cast<float2*>(&As[row * AS + kk]);
        d[0] = make_float2(ra[h * 4 + 0], ra[h * 4 + 1]);
        d[1] = make_float2(ra[h * 4 + 2], ra[h * 4 + 3]);
      }
    } else {
#pragma unroll
      for (int h = 0; h < 16; ++h) As[((tid >> 5) + h * 8) * AS + (tid & 31)] = ra[h];
    }
#pragma unroll
    for (int h = 0; h < WREG; ++h) {
      const int idx = tid + h * kBlock;
      Ws[(idx / NW) * NS + (idx % NW)] = rw[h];
    }
  };

  load_tile(0);
  const float* arow = &As[(wave * 32 + (lane & 15)) * AS + (lane >> 4)];
  const float* brow = &Ws[(lane >> 4) * NS + (lane & 15)];
  for (int ti = 0; ti < total; ++ti) {
    __syncthreads();   // everyone is done reading the previous tile
    store_tile();
    __syncthreads();
    if (ti + 1 < total) load_tile(ti + 1);
#pragma unroll
    for (int ks = 0; ks < KT / 4; ++ks) {   // K tail: the staged tile is zero-padded
      const float a0 = arow[ks * 4];
      const float a1 = arow[16 * AS + ks * 4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = brow[ks * 4 * NS + nt * 16];
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][nt], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: bias, row map, store
  const int col_l = lane & 15;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + wave * 32 + r * 16 + (lane >> 4) * 4 + i;
      if (m >= p.M) continue;
      const int64_t orow = proj_orow(p, m);
      const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + nt * 16 + col_l;
        if (col >= p.N) continue;
        float v = acc[r][nt][i];
        if (p.bias_kind == 1 && col < p.bias_cols) v += p.bias[col];
        else if (p.bias_kind == 2 && col < p.bias_cols) v += p.bias[vert * p.bias_ld + col];
        float* o = p.out + orow * p.ldo + col;
        if (p.accumulate) v += *o;
        *o = v;
      }
    }
}

// ---- bf16x3 variant of the streaming-W kernel: fp32-accurate products on the bf16 matrix pipe (16x the fp32 MFMA
// rate).  Every fp32 operand is split into three bf16 terms a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1),
// a3 = bf16(a - a1 - a2): 24 mantissa bits in all, the subtractions are exact) and the product is summed as
// a3w1 + a2w2 + a1w3 + a2w1 + a1w2 + a1w1 with v_mfma_f32_16x16x32_bf16 in fp32 accumulators; the three dropped
// cross terms are below 2^-24 of |a w|.  6 bf16 MFMAs replace 8 fp32 ones per 32 k at 1/2 the cycles each.
// Operand maps (gfx950): A[row = l&15][k = 8*(l>>4) + j], B[k = 8*(l>>4) + j][col = l&15], j = 0..7; C/D as fp32.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ void split3(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  const bf16x2 h = {(__bf16)a, (__bf16)b};                    // v_cvt_pk_bf16_f32, round to nearest even
  const unsigned hu = __builtin_bit_cast(unsigned, h);
  const float ra = a - __uint_as_float(hu << 16), rb = b - __uint_as_float(hu & 0xFFFF0000u);
  const bf16x2 m = {(__bf16)ra, (__bf16)rb};
  const unsigned mu = __builtin_bit_cast(unsigned, m);
  const float sa = ra - __uint_as_float(mu << 16), sb = rb - __uint_as_float(mu & 0xFFFF0000u);
  const bf16x2 l = {(__bf16)sa, (__bf16)sb};
  p1 = hu; p2 = mu; p3 = __builtin_bit_cast(unsigned, l);
}

// Swizzle of the four 16-byte chunks (8 k each) of a 64-byte LDS row: chunk c of row r lives at c ^ G[(r >> 2) & 3],
// G = {0,2,3,1}.  With ds_read_b128's lane groups ({0-3,12-15,20-27}, ...) the 16 fragment reads of a group then fall
// on 16 different 16-byte slots of the 256-byte bank row, and ds_write_b64 of whole rows is conflict-free too.
__device__ __forceinline__ int x3_chunk(int r, int c) { return c ^ ((0x1320 >> (((r >> 2) & 3) * 4)) & 3); }

template <int NT, bool VEC4>
__global__ __launch_bounds__(512) void project_x3_kernel(const ProjParams p) {
  constexpr int XT = 512;                          // 8 waves x 32 rows: one W tile (and its split) serves 256 rows
  constexpr int BM = 256, KT = 32, RS = KT;       // LDS rows of 32 bf16 (64 B), 16-byte chunks XOR-swizzled (x3_chunk)
  constexpr int NW = NT * 16;
  constexpr int WPAIRS = (KT / 2 * NW + XT - 1) / XT;  // (k, k+1) pairs of one column per thread
  __shared__ __align__(16) unsigned short Ap[3][BM * RS];
  __shared__ __align__(16) unsigned short Wp[3][NW * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bb = (int)(blockIdx.x % (unsigned)p.nbatch);          // sample of the batch: consecutive workgroups share a tile
  const int64_t m0 = (int64_t)(blockIdx.x / (unsigned)p.nbatch) * BM;
  const int n0 = blockIdx.y * NW;
  float* const outb = p.out + (int64_t)bb * p.out_bs;
  f32x4 acc[2][NT];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ktiles = (p.Kc + KT - 1) / KT;
  const int total = p.nterms * ktiles;

  float ra[16], rw[2 * WPAIRS];
  auto load_tile = [&](int ti) {
    const int term = ti / ktiles, k0 = (ti % ktiles) * KT;
    const float* __restrict__ A = p.a[term] + (int64_t)bb * p.a_bs[term];
    const int64_t lda = p.lda[term];
    const float* __restrict__ Wt = p.W + (int64_t)term * p.Kc * p.N;
    if constexpr (VEC4) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = (tid >> 3) + h * 64, kk = (tid & 7) * 4;
        const bool ok = (m0 + row < p.M) && (k0 + kk < p.Kc);
        const int64_t rr = (m0 + row < p.M) ? m0 + row : p.M - 1;
        const int kc = (k0 + kk < p.Kc) ? k0 + kk : 0;
        const float4 v = *reinterpret_cast<const float4*>(A + proj_arow(p, term, rr) * lda + kc);
        ra[h * 4 + 0] = ok ? v.x : 0.f; ra[h * 4 + 1] = ok ? v.y : 0.f; ra[h * 4 + 2] = ok ? v.z : 0.f; ra[h * 4 + 3] = ok ? v.w : 0.f;
      }
    } else {      // thread = (row, k pair)
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int row = (tid >> 4) + h * 32, kk = (tid & 15) * 2;
        const int64_t rr = (m0 + row < p.M) ? m0 + row : p.M - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool ok = (m0 + row < p.M) && (k0 + kk + j < p.Kc);
          const float v = A[proj_row_off(p, proj_arow(p, term, rr), lda) + (ok ? k0 + kk + j : 0)];
          ra[h * 2 + j] = ok ? v : 0.f;
        }
      }
    }
    {
#pragma unroll
      for (int h = 0; h < WPAIRS; ++h) {
        const int idx = min(tid + h * XT, KT / 2 * NW - 1);
        const int cc = idx % NW, kk = (idx / NW) * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool ok = (k0 + kk + j < p.Kc) && (n0 + cc < p.N);
          const float v = Wt[(int64_t)(ok ? k0 + kk + j : 0) * p.N + (ok ? n0 + cc : 0)];
          rw[h * 2 + j] = ok ? v : 0.f;
        }
      }
    }
  };
  auto store_tile = [&]() {      // split into the three bf16 planes on the way into LDS
    if constexpr (VEC4) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = (tid >> 3) + h * 64, kk = (tid & 7) * 4;
        unsigned a1, a2, a3, b1, b2, b3;
        split3(ra[h * 4 + 0], ra[h * 4 + 1], a1, a2, a3);
        split3(ra[h * 4 + 2], ra[h * 4 + 3], b1, b2, b3);
        const int o = row * RS + x3_chunk(row, kk >> 3) * 8 + (kk & 7);
        *reinterpret_cast<uint2*>(&Ap[0][o]) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(&Ap[1][o]) = make_uint2(a2, b2);
        *reinterpret_cast<uint2*>(&Ap[2][o]) = make_uint2(a3, b3);
      }
    } else {
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int row = (tid >> 4) + h * 32, kk = (tid & 15) * 2;
        unsigned a1, a2, a3;
        split3(ra[h * 2 + 0], ra[h * 2 + 1], a1, a2, a3);
        const int o = row * RS + x3_chunk(row, kk >> 3) * 8 + (kk & 7);
        *reinterpret_cast<unsigned*>(&Ap[0][o]) = a1;
        *reinterpret_cast<unsigned*>(&Ap[1][o]) = a2;
        *reinterpret_cast<unsigned*>(&Ap[2][o]) = a3;
      }
    }
    {
#pragma unroll
      for (int h = 0; h < WPAIRS; ++h) {       // W tile transposed: [column][k], so a fragment's 8 k are contiguous
        const int idx = tid + h * XT;
        if (idx >= KT / 2 * NW) continue;
        const int cc = idx % NW, kk = (idx / NW) * 2;
        unsigned w1, w2, w3;
        split3(rw[h * 2 + 0], rw[h * 2 + 1], w1, w2, w3);
        const int o = cc * RS + x3_chunk(cc, kk >> 3) * 8 + (kk & 7);
        *reinterpret_cast<unsigned*>(&Wp[0][o]) = w1;
        *reinterpret_cast<unsigned*>(&Wp[1][o]) = w2;
        *reinterpret_cast<unsigned*>(&Wp[2][o]) = w3;
      }
    }
  };

  load_tile(0);
  const int frag = (lane & 15) * RS + x3_chunk(lane & 15, lane >> 4) * 8;   // this lane's 8 consecutive k of row / column (lane & 15)
  for (int ti = 0; ti < total; ++ti) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (ti + 1 < total) load_tile(ti + 1);
    bf16x8 a[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        a[r][pl] = *reinterpret_cast<const bf16x8*>(&Ap[pl][(wave * 32 + r * 16) * RS + frag]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bf16x8 w[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const bf16x8*>(&Wp[pl][(nt * 16) * RS + frag]);
#pragma unroll
      for (int r = 0; r < 2; ++r) {      // smallest terms first
        f32x4 c = acc[r][nt];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][2], w[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][1], w[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][0], w[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][1], w[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][0], w[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][0], w[0], c, 0, 0, 0);
        acc[r][nt] = c;
      }
    }
  }
  // ---- epilogue
  if constexpr (NT <= 4) {
    if (p.vec_epilogue) {
      // accumulators -> wave-private scratch (the A planes are free now) -> float4 rows: coalesced bias loads, 16-byte stores
      constexpr int ES = NW + 4;                       // scratch row stride in floats
      constexpr int SEGS = NW / 4, ITER = (16 * SEGS) / 64;
      __syncthreads();                                 // every wave is done reading the last tile's planes
      float* my = reinterpret_cast<float*>(&Ap[0][0]) + wave * (16 * ES);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int i = 0; i < 4; ++i) my[((lane >> 4) * 4 + i) * ES + nt * 16 + (lane & 15)] = acc[r][nt][i];
        if (p.pool > 1) {            // relu + max over consecutive vertices instead of the plain store (wave-uniform branch)
          pooled_store(p, my, ES, 16, m0 + wave * 32 + r * 16, n0, NW, lane);
          continue;
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int idx = lane + 64 * it, row = idx / SEGS, seg = (idx % SEGS) * 4;
          const int64_t m = m0 + wave * 32 + r * 16 + row;
          const int col = n0 + seg;
          if (m >= p.M || col >= p.N) continue;
          float4 v = *reinterpret_cast<const float4*>(&my[row * ES + seg]);
          const int64_t orow = proj_orow(p, m);
          if (p.bias_kind && col < p.bias_cols) {
            const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + (p.bias_kind == 2 ? vert * p.bias_ld : 0) + col);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          }
          float4* o = reinterpret_cast<float4*>(outb + orow * p.ldo + col);
          if (p.accumulate) { const float4 ov = *o; v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w; }
          *o = v;
        }
      }
      return;
    }
  }
  const int col_l = lane & 15;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + wave * 32 + r * 16 + (lane >> 4) * 4 + i;
      if (m >= p.M) continue;
      const int64_t orow = proj_orow(p, m);
      const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + nt * 16 + col_l;
        if (col >= p.N) continue;
        float v = acc[r][nt][i];
        if (p.bias_kind == 1 && col < p.bias_cols) v += p.bias[col];
        else if (p.bias_kind == 2 && col < p.bias_cols) v += p.bias[vert * p.bias_ld + col];
        float* o = outb + orow * p.ldo + col;
        if (p.accumulate) v += *o;
        *o = v;
      }
    }
}

// ---- project_x3_kernel<NT, true> with the LAST HOP fused in (ProjParams.g_*): the A tiles of term g_term are produced by gathers for
// the rows of at most g_thresh stored entries.  A thread owns 4 floats of 4 tile rows (as in the plain kernel's tile load); per row it
// walks the stored entries four at a time -- the 8 threads of a row read the same entries (one request) and each gathers its own 16 bytes
// of the neighbour row's 128-byte half -- and adds val * x in stored order (fmaf), so the values equal hop_kernel's bit for bit.
// Everything else (split into bf16 planes, MFMA order, epilogue) is the plain kernel's, hence the same results as hop + projection.
template <int NT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void project_x3_gather_kernel(const ProjParams p) {   // two workgroups per CU like the plain kernel: at most 128 VGPRs
  constexpr bool VEC4 = true;
  constexpr int XT = 512;                          // 8 waves x 32 rows: one W tile (and its split) serves 256 rows
  constexpr int BM = 256, KT = 32, RS = KT;       // LDS rows of 32 bf16 (64 B), 16-byte chunks XOR-swizzled (x3_chunk)
  constexpr int NW = NT * 16;
  constexpr int WPAIRS = (KT / 2 * NW + XT - 1) / XT;  // (k, k+1) pairs of one column per thread
  __shared__ __align__(16) unsigned short Ap[3][BM * RS];
  __shared__ __align__(16) unsigned short Wp[3][NW * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bb = (int)(blockIdx.x % (unsigned)p.nbatch);          // sample of the batch: consecutive workgroups share a tile
  const int64_t m0 = (int64_t)(blockIdx.x / (unsigned)p.nbatch) * BM;
  const int n0 = blockIdx.y * NW;
  float* const outb = p.out + (int64_t)bb * p.out_bs;
  f32x4 acc[2][NT];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ktiles = (p.Kc + KT - 1) / KT;
  const int total = p.nterms * ktiles;

  float ra[16], rw[2 * WPAIRS];
  auto load_tile = [&](int ti) {
    const int term = ti / ktiles, k0 = (ti % ktiles) * KT;
    const float* __restrict__ A = p.a[term] + (int64_t)bb * p.a_bs[term];
    const int64_t lda = p.lda[term];
    const float* __restrict__ Wt = p.W + (int64_t)term * p.Kc * p.N;
    if constexpr (VEC4) {
      const int kk = (tid & 7) * 4;
      const int kc = (k0 + kk < p.Kc) ? k0 + kk : 0;
      int64_t rr[4];
      bool ok[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = (tid >> 3) + h * 64;
        ok[h] = (m0 + row < p.M) && (k0 + kk < p.Kc);
        rr[h] = (m0 + row < p.M) ? m0 + row : p.M - 1;
      }
      if (term == p.g_term) {
        // this thread's 4 tile rows side by side, two stored entries each per step: 8 independent entry -> gather chains in flight.
        // Rows above the threshold are read from memory like any other term (the hop launch wrote them).
        int e[4], e1[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) { e[h] = p.g_rowptr[rr[h]]; e1[h] = p.g_rowptr[rr[h] + 1]; }      // tile row = compact row: hop tensors are never row-mapped
        int len = 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          if (e1[h] - e[h] > p.g_thresh) {
            const float4 v = *reinterpret_cast<const float4*>(A + rr[h] * lda + kc);
            ra[h * 4 + 0] = v.x; ra[h * 4 + 1] = v.y; ra[h * 4 + 2] = v.z; ra[h * 4 + 3] = v.w;
            e1[h] = e[h];
          } else {
            ra[h * 4 + 0] = ra[h * 4 + 1] = ra[h * 4 + 2] = ra[h * 4 + 3] = 0.f;
          }
          len = max(len, e1[h] - e[h]);
        }
        const float* __restrict__ Xg = p.g_X + (int64_t)bb * p.g_xbs + kc;
        for (int j = 0; j < len; j += 2) {
          int c[4][2];
          float w[4][2];
          float4 xv[4][2];
#pragma unroll
          for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              c[h][u] = 0; w[h][u] = 0.f;
              if (e[h] + j + u < e1[h]) { const tgcn_edge t = p.g_edges[e[h] + j + u]; c[h][u] = t.col; w[h][u] = t.val; }
            }
#pragma unroll
          for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              xv[h][u] = make_float4(0.f, 0.f, 0.f, 0.f);
              if (e[h] + j + u < e1[h]) xv[h][u] = *reinterpret_cast<const float4*>(Xg + (int64_t)c[h][u] * lda);
            }
#pragma unroll
          for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (e[h] + j + u < e1[h]) {      // predicated, not multiplied by zero: Inf / NaN rows of X propagate as in hop_kernel; stored order
                ra[h * 4 + 0] = fmaf(w[h][u], xv[h][u].x, ra[h * 4 + 0]); ra[h * 4 + 1] = fmaf(w[h][u], xv[h][u].y, ra[h * 4 + 1]);
                ra[h * 4 + 2] = fmaf(w[h][u], xv[h][u].z, ra[h * 4 + 2]); ra[h * 4 + 3] = fmaf(w[h][u], xv[h][u].w, ra[h * 4 + 3]);
              }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h)
          if (!ok[h]) ra[h * 4 + 0] = ra[h * 4 + 1] = ra[h * 4 + 2] = ra[h * 4 + 3] = 0.f;
      } else {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const float4 v = *reinterpret_cast<const float4*>(A + proj_arow(p, term, rr[h]) * lda + kc);
          ra[h * 4 + 0] = ok[h] ? v.x : 0.f; ra[h * 4 + 1] = ok[h] ? v.y : 0.f; ra[h * 4 + 2] = ok[h] ? v.z : 0.f; ra[h * 4 + 3] = ok[h] ? v.w : 0.f;
        }
      }
    } else {      // thread = (row, k pair)
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int row = (tid >> 4) + h * 32, kk = (tid & 15) * 2;
        const int64_t rr = (m0 + row < p.M) ? m0 + row : p.M - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool ok = (m0 + row < p.M) && (k0 + kk + j < p.Kc);
          const float v = A[proj_row_off(p, proj_arow(p, term, rr), lda) + (ok ? k0 + kk + j : 0)];
          ra[h * 2 + j] = ok ? v : 0.f;
        }
      }
    }
    {
#pragma unroll
      for (int h = 0; h < WPAIRS; ++h) {
        const int idx = min(tid + h * XT, KT / 2 * NW - 1);
        const int cc = idx % NW, kk = (idx / NW) * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool ok = (k0 + kk + j < p.Kc) && (n0 + cc < p.N);
          const float v = Wt[(int64_t)(ok ? k0 + kk + j : 0) * p.N + (ok ? n0 + cc : 0)];
          rw[h * 2 + j] = ok ? v : 0.f;
        }
      }
    }
  };
  auto store_tile = [&]() {      // split into the three bf16 planes on the way into LDS
    if constexpr (VEC4) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = (tid >> 3) + h * 64, kk = (tid & 7) * 4;
        unsigned a1, a2, a3, b1, b2, b3;
        split3(ra[h * 4 + 0], ra[h * 4 + 1], a1, a2, a3);
        split3(ra[h * 4 + 2], ra[h * 4 + 3], b1, b2, b3);
        const int o = row * RS + x3_chunk(row, kk >> 3) * 8 + (kk & 7);
        *reinterpret_cast<uint2*>(&Ap[0][o]) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(&Ap[1][o]) = make_uint2(a2, b2);
        *reinterpret_cast<uint2*>(&Ap[2][o]) = make_uint2(a3, b3);
      }
    } else {
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int row = (tid >> 4) + h * 32, kk = (tid & 15) * 2;
        unsigned a1, a2, a3;
        split3(ra[h * 2 + 0], ra[h * 2 + 1], a1, a2, a3);
        const int o = row * RS + x3_chunk(row, kk >> 3) * 8 + (kk & 7);
        *reinterpret_cast<unsigned*>(&Ap[0][o]) = a1;
        *reinterpret_cast<unsigned*>(&Ap[1][o]) = a2;
        *reinterpret_cast<unsigned*>(&Ap[2][o]) = a3;
      }
    }
    {
#pragma unroll
      for (int h = 0; h < WPAIRS; ++h) {       // W tile transposed: [column][k], so a fragment's 8 k are contiguous
        const int idx = tid + h * XT;
        if (idx >= KT / 2 * NW) continue;
        const int cc = idx % NW, kk = (idx / NW) * 2;
        unsigned w1, w2, w3;
        split3(rw[h * 2 + 0], rw[h * 2 + 1], w1, w2, w3);
        const int o = cc * RS + x3_chunk(cc, kk >> 3) * 8 + (kk & 7);
        *reinterpret_cast<unsigned*>(&Wp[0][o]) = w1;
        *reinterpret_cast<unsigned*>(&Wp[1][o]) = w2;
        *reinterpret_cast<unsigned*>(&Wp[2][o]) = w3;
      }
    }
  };

  load_tile(0);
  const int frag = (lane & 15) * RS + x3_chunk(lane & 15, lane >> 4) * 8;   // this lane's 8 consecutive k of row / column (lane & 15)
  for (int ti = 0; ti < total; ++ti) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (ti + 1 < total) load_tile(ti + 1);
    bf16x8 a[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        a[r][pl] = *reinterpret_cast<const bf16x8*>(&Ap[pl][(wave * 32 + r * 16) * RS + frag]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bf16x8 w[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const bf16x8*>(&Wp[pl][(nt * 16) * RS + frag]);
#pragma unroll
      for (int r = 0; r < 2; ++r) {      // smallest terms first
        f32x4 c = acc[r][nt];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][2], w[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][1], w[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][0], w[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][1], w[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][0], w[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][0], w[0], c, 0, 0, 0);
        acc[r][nt] = c;
      }
    }
  }
  // ---- epilogue
  if constexpr (NT <= 4) {
    if (p.vec_epilogue) {
      // accumulators -> wave-private scratch (the A planes are free now) -> float4 rows: coalesced bias loads, 16-byte stores
      constexpr int ES = NW + 4;                       // scratch row stride in floats
      constexpr int SEGS = NW / 4, ITER = (16 * SEGS) / 64;
      __syncthreads();                                 // every wave is done reading the last tile's planes
      float* my = reinterpret_cast<float*>(&Ap[0][0]) + wave * (16 * ES);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int i = 0; i < 4; ++i) my[((lane >> 4) * 4 + i) * ES + nt * 16 + (lane & 15)] = acc[r][nt][i];
        if (p.pool > 1) {            // relu + max over consecutive vertices instead of the plain store (wave-uniform branch)
          pooled_store(p, my, ES, 16, m0 + wave * 32 + r * 16, n0, NW, lane);
          continue;
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int idx = lane + 64 * it, row = idx / SEGS, seg = (idx % SEGS) * 4;
          const int64_t m = m0 + wave * 32 + r * 16 + row;
          const int col = n0 + seg;
          if (m >= p.M || col >= p.N) continue;
          float4 v = *reinterpret_cast<const float4*>(&my[row * ES + seg]);
          const int64_t orow = proj_orow(p, m);
          if (p.bias_kind && col < p.bias_cols) {
            const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + (p.bias_kind == 2 ? vert * p.bias_ld : 0) + col);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          }
          float4* o = reinterpret_cast<float4*>(outb + orow * p.ldo + col);
          if (p.accumulate) { const float4 ov = *o; v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w; }
          *o = v;
        }
      }
      return;
    }
  }
  const int col_l = lane & 15;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + wave * 32 + r * 16 + (lane >> 4) * 4 + i;
      if (m >= p.M) continue;
      const int64_t orow = proj_orow(p, m);
      const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + nt * 16 + col_l;
        if (col >= p.N) continue;
        float v = acc[r][nt][i];
        if (p.bias_kind == 1 && col < p.bias_cols) v += p.bias[col];
        else if (p.bias_kind == 2 && col < p.bias_cols) v += p.bias[vert * p.bias_ld + col];
        float* o = outb + orow * p.ldo + col;
        if (p.accumulate) v += *o;
        *o = v;
      }
    }
}

// ---- bf16x3, second form (16-byte aligned operands): a wave's A rows are used by that wave only, so its A fragments
// go global -> registers -> split -> MFMA operand with no LDS round trip and no barrier; only the W tile (shared by the
// 8 waves) is split into LDS, double-buffered, ONE barrier per 32-k tile.  A lane loads the 8 consecutive k of its row
// as two float4 (the four k groups of a row are adjacent: whole 128-byte lines per row).
template <int NT, int RT>
__device__ __forceinline__ void x3v2_body(const ProjParams& p, unsigned char* lds_raw, const int64_t m0, const int n0) {
  constexpr int XT = 512, KT = 32, RS = KT;          // 8 waves x RT 16-row MFMA tiles: 128 * RT rows per workgroup
  constexpr int NW = NT * 16;
  constexpr int WPAIRS = (KT / 2 * NW + XT - 1) / XT;
  constexpr int WBUF = 3 * NW * RS;                                   // bf16 elements of one W buffer (3 planes)
  constexpr int ES = NW + 4;                                          // epilogue scratch row stride (floats)
  unsigned short* Wp = reinterpret_cast<unsigned short*>(lds_raw);    // [2][3][NW * RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  f32x4 acc[RT][NT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ktiles = (p.Kc + KT - 1) / KT;
  const int total = p.nterms * ktiles;
  // per-lane, tile-invariant parts of every address (the loop below adds only wave-uniform tile offsets: the vector ALU
  // is the co-bottleneck of this kernel -- an MFMA holds vector issue for 8 of its 16 cycles)
  int64_t rowc[RT], rowm[RT];
  const uint32_t mapped_bits = p.rowmap ? (p.mapped & ~kProjMapTermsOnly) : 0u;
  const bool any_mapped = mapped_bits != 0u;
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int64_t m = m0 + wave * (16 * RT) + r * 16 + r16;
    rowc[r] = m < p.M ? m : p.M - 1;                 // rows past the end re-read the last row; their results are never stored
    // the row a MAPPED term reads, fetched once per tile row (round 6: proj_arow inside load_a re-read the map and re-tested the pointer for
    // every load of every k tile -- with a row map that maps no term at all, as the vertex shards' Z projection passes, cfg4's
    // 90 k x 1200 x 160 contraction went from 0.22 to 0.30 ms)
    // (kept as the DIFFERENCE to the tile row: `tm ? rowm[r] : rowc[r]` becomes a select between two arrays, which puts both into scratch)
    rowm[r] = any_mapped ? (int64_t)p.rowmap[rowc[r]] - rowc[r] : 0;
  }
  int wsrc[WPAIRS], wdst[WPAIRS], wkk[WPAIRS];
  bool wcol[WPAIRS];
#pragma unroll
  for (int h = 0; h < WPAIRS; ++h) {
    const int idx = min(tid + h * XT, KT / 2 * NW - 1);
    const int cc = idx % NW, kk = (idx / NW) * 2;
    wkk[h] = kk;
    wcol[h] = (tid + h * XT < KT / 2 * NW) && (n0 + cc < p.N);
    wsrc[h] = kk * p.N + (n0 + cc < p.N ? n0 + cc : 0);
    wdst[h] = cc * RS + x3_chunk(cc, kk >> 3) * 8 + (kk & 7);
  }
  float ra[RT][8], rw[2 * WPAIRS];
  auto load_a = [&](int ti, float (&dst)[RT][8]) {
    const int term = ti / ktiles, k0 = (ti % ktiles) * KT;           // wave-uniform
    const float* __restrict__ A = p.a[term] + k0 + kg * 8;
    const int64_t lda = p.lda[term];
    const int64_t tm = -(int64_t)((mapped_bits >> term) & 1u);       // wave-uniform mask: all ones for a mapped term
    if (k0 + KT <= p.Kc) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 v = *reinterpret_cast<const float4*>(A + (rowc[r] + (rowm[r] & tm)) * lda + h * 4);
          dst[r][h * 4 + 0] = v.x; dst[r][h * 4 + 1] = v.y; dst[r][h * 4 + 2] = v.z; dst[r][h * 4 + 3] = v.w;
        }
    } else {                                                         // last k tile of a term: k past Kc reads as zero
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = k0 + kg * 8 + h * 4 < p.Kc;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok) v = *reinterpret_cast<const float4*>(A + (rowc[r] + (rowm[r] & tm)) * lda + h * 4);
          dst[r][h * 4 + 0] = v.x; dst[r][h * 4 + 1] = v.y; dst[r][h * 4 + 2] = v.z; dst[r][h * 4 + 3] = v.w;
        }
    }
  };
  auto load_w = [&](int ti) {
    const int term = ti / ktiles, k0 = (ti % ktiles) * KT;
    const float* __restrict__ Wt = p.W + ((int64_t)term * p.Kc + k0) * p.N;
    if (k0 + KT <= p.Kc) {
#pragma unroll
      for (int h = 0; h < WPAIRS; ++h) {
        rw[h * 2 + 0] = wcol[h] ? Wt[wsrc[h]] : 0.f;
        rw[h * 2 + 1] = wcol[h] ? Wt[wsrc[h] + p.N] : 0.f;
      }
    } else {
#pragma unroll
      for (int h = 0; h < WPAIRS; ++h) {
        rw[h * 2 + 0] = (wcol[h] && k0 + wkk[h] < p.Kc) ? Wt[wsrc[h]] : 0.f;
        rw[h * 2 + 1] = (wcol[h] && k0 + wkk[h] + 1 < p.Kc) ? Wt[wsrc[h] + p.N] : 0.f;
      }
    }
  };
  auto store_w = [&](int buf) {            // W tile transposed [column][k], split into the three planes
    unsigned short* W0 = Wp + buf * WBUF;
#pragma unroll
    for (int h = 0; h < WPAIRS; ++h) {
      if (tid + h * XT >= KT / 2 * NW) continue;
      unsigned w1, w2, w3;
      split3(rw[h * 2 + 0], rw[h * 2 + 1], w1, w2, w3);
      *reinterpret_cast<unsigned*>(&W0[wdst[h]]) = w1;
      *reinterpret_cast<unsigned*>(&W0[NW * RS + wdst[h]]) = w2;
      *reinterpret_cast<unsigned*>(&W0[2 * NW * RS + wdst[h]]) = w3;
    }
  };
  const int frag = r16 * RS + x3_chunk(r16, kg) * 8;
  // split the 8 floats per row tile of a wave into the three bf16 planes (the MFMA A operands of one k tile)
  auto split_a = [&](const float (&src)[RT][8], bf16x8 (&dst)[RT][3]) {
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      unsigned pl[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split3(src[r][2 * j], src[r][2 * j + 1], pl[0][j], pl[1][j], pl[2][j]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        dst[r][q] = __builtin_bit_cast(bf16x8, u32x4{pl[q][0], pl[q][1], pl[q][2], pl[q][3]});
      }
    }
  };
  // the MFMAs of column tiles [nt0, nt1) of one k tile: smallest terms first; the row tiles of a wave alternate so that consecutive MFMAs are independent
  auto mfma_tiles = [&](const bf16x8 (&a)[RT][3], const unsigned short* W0, const int nt0, const int nt1) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt < nt0 || nt >= nt1) continue;
      bf16x8 w[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) w[q] = *reinterpret_cast<const bf16x8*>(&W0[q * NW * RS + (nt * 16) * RS + frag]);
      f32x4 c[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) c[r] = acc[r][nt];
#pragma unroll
      for (int r = 0; r < RT; ++r) c[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][2], w[0], c[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) c[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][1], w[1], c[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) c[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][0], w[2], c[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) c[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][1], w[0], c[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) c[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][0], w[1], c[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) c[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][0], w[0], c[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r][nt] = c[r];
    }
  };
  int ti = 0;
  load_a(0, ra);
  load_w(0);
  store_w(0);
  // ---- software-pipelined main loop (round 5, wide outputs).  The plain loop below runs every k tile as three PHASES -- split the A rows
  // (~90 vector instructions), 12 * NT MFMAs, split + stage the next W tile (~60 vector instructions) -- and with one barrier per tile both waves of
  // a SIMD are in the same phase: SQ_VALU_MFMA_COEXEC_CYCLES was 4 % of SQ_VALU_MFMA_BUSY_CYCLES on cfg4 (matrix pipe 36 % busy, profiles/
  // r05_cfg4_projection_sq.json).  Here an iteration multiplies tile t while the SAME basic block splits the A rows of tile t+1 and splits + stages
  // W of tile t+1 (so that the scheduler can place that vector work next to the MFMAs -- it puts the W split between the MFMAs of the second half
  // and the A split behind them), and the loads of tile t+2 are issued in the MIDDLE of the iteration, as soon as the registers they fill are free:
  // one and a half iterations of slack instead of one.  cfg4 (90,000 x 1200 x 160): 0.277 -> 0.212 ms.  Forcing an instruction-level
  // MFMA / vector interleave in the first half too (slices of the split pinned between fences by empty asm statements) or issuing each W pair's
  // load right after its split measured 0.219 / 0.225 ms: not it (docs/EXPERIMENTS.md A.5).  Only full 32-k tiles run here; the last two (and a
  // partial one) drain through the plain loop.  Bitwise the plain loop's result (same products, same order).
  const int nfull = (p.Kc % KT == 0) ? total : (p.nterms == 1 ? ktiles - 1 : 0);
  // (a row map that maps no TERM -- output rows only, as the vertex shards' Z projection passes -- takes this loop too since round 6: the
  //  `!p.rowmap` it was guarded by sent that call through the plain loop, 0.26 -> 0.31 ms on cfg4's contraction)
  if (NT >= 6 && nfull >= 4 && !any_mapped) {
    float rn[RT][8];
    bf16x8 a_cur[RT][3], a_next[RT][3];
    auto load_a_full = [&](int t, float (&dst)[RT][8]) {            // full tiles only: no bounds on k
      const int term = t / ktiles, k0 = (t % ktiles) * KT;
      const float* __restrict__ A = p.a[term] + k0 + kg * 8;
      const int64_t lda = p.lda[term];
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 v = *reinterpret_cast<const float4*>(A + rowc[r] * lda + h * 4);           // (no row map on this path)
          dst[r][h * 4 + 0] = v.x; dst[r][h * 4 + 1] = v.y; dst[r][h * 4 + 2] = v.z; dst[r][h * 4 + 3] = v.w;
        }
    };
    auto load_w_full = [&](int t) {                                  // unconditional loads (wsrc is clamped into the tile) + select: no branch in the loop body
      const int term = t / ktiles, k0 = (t % ktiles) * KT;
      const float* __restrict__ Wt = p.W + ((int64_t)term * p.Kc + k0) * p.N;
#pragma unroll
      for (int h = 0; h < WPAIRS; ++h) {
        const float v0 = Wt[wsrc[h]], v1 = Wt[wsrc[h] + p.N];
        rw[h * 2 + 0] = wcol[h] ? v0 : 0.f;
        rw[h * 2 + 1] = wcol[h] ? v1 : 0.f;
      }
    };
    split_a(ra, a_cur);
    load_a_full(1, rn);
    load_w_full(1);
    __syncthreads();
    for (; ti + 2 < nfull; ++ti) {                   // invariant: a_cur = planes of tile ti, rn / rw = raw A / W of tile ti+1, LDS buffer ti & 1 = W planes of tile ti
      const unsigned short* W0 = Wp + (ti & 1) * WBUF;
      mfma_tiles(a_cur, W0, 0, NT / 2);
      split_a(rn, a_next);                           // tile ti+1 (its loads were issued in the middle of the previous iteration)
      // requested order of the first half: per column tile its 3 fragment reads, then its 6 * RT MFMAs two at a time with vector instructions between
#pragma unroll
      for (int i = 0; i < NT / 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int j = 0; j < 3 * RT; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_a_full(ti + 2, rn);
      mfma_tiles(a_cur, W0, NT / 2, NT);
      store_w((ti + 1) & 1);                         // split + stage W of tile ti+1 into the other buffer (free since the barrier of iteration ti-1)
      // second half: the W split (about 11 vector instructions per pair of weights and 3 LDS writes) between the MFMAs
#pragma unroll
      for (int i = 0; i < NT - NT / 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int j = 0; j < 3 * RT; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
      }
      load_w_full(ti + 2);
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) a_cur[r][q] = a_next[r][q];
      __syncthreads();
    }
    // hand over to the plain loop at tile ti: it wants the RAW rows of tile ti in `ra` -- they are gone (only the planes are kept), so the plain
    // loop's first iteration is done here by hand: multiply tile ti from a_cur, stage W of ti+1, raw rows of ti+1 become `ra`
    {
      const unsigned short* W0 = Wp + (ti & 1) * WBUF;
      mfma_tiles(a_cur, W0, 0, NT);
      store_w((ti + 1) & 1);
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) ra[r][j] = rn[r][j];
      __syncthreads();
      ++ti;
    }
  } else {
    __syncthreads();
  }
  for (; ti < total; ++ti) {
    float rn[RT][8];
    const bool more = ti + 1 < total;
    if (more) { load_a(ti + 1, rn); load_w(ti + 1); }
    bf16x8 a[RT][3];
    split_a(ra, a);
    mfma_tiles(a, Wp + (ti & 1) * WBUF, 0, NT);
    if (more) {
      store_w((ti + 1) & 1);
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) ra[r][j] = rn[r][j];
    }
    __syncthreads();
  }
  // ---- epilogue
  // through the wave's LDS scratch for EVERY width (round 6; the kernel's LDS was already sized for it): 16-byte bias loads and 16-byte stores of
  // whole 640-byte rows.  Until round 5 the widths this kernel is actually chosen for (>= 96 columns) took the scalar form below -- 80 dependent
  // bias-load / wait / add / 4-byte-store sequences per lane at one workgroup per CU.
  if (p.vec_epilogue && p.pool <= 1) {
    constexpr int SEGS = NW / 4, ITER = (16 * SEGS) / 64;
    float* my = reinterpret_cast<float*>(lds_raw) + wave * (16 * ES);     // the loop ended with a barrier: W buffers are free
#pragma unroll
    for (int r = 0; r < RT; ++r) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) my[(kg * 4 + i) * ES + nt * 16 + r16] = acc[r][nt][i];
      // three passes over the lane's ITER 16-byte pieces: the row-map entries, then the bias pieces (their address depends on the mapped row),
      // then the LDS reads and the stores.  Left interleaved, every piece is a chain  map load -> wait -> bias load -> wait -> store  (the stores
      // may alias anything as far as the compiler knows): 20 round trips to memory per lane at one workgroup per CU, measured +0.05 ms on a
      // 45 k-row tile round (tools/proj_rowmap_cost.py)
      int64_t orow_[ITER];
      float4 bv_[ITER];
      bool ok_[ITER];
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int idx = lane + 64 * it, row = idx / SEGS, seg = (idx % SEGS) * 4;
        const int64_t m = m0 + wave * (16 * RT) + r * 16 + row;
        ok_[it] = m < p.M && n0 + seg < p.N;
        orow_[it] = proj_orow(p, ok_[it] ? m : p.M - 1);
      }
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int col = n0 + ((lane + 64 * it) % SEGS) * 4;
        bv_[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok_[it] && p.bias_kind && col < p.bias_cols) {
          const int64_t vert = orow_[it] < p.n_vertices ? orow_[it] : orow_[it] % p.n_vertices;
          bv_[it] = *reinterpret_cast<const float4*>(p.bias + (p.bias_kind == 2 ? vert * p.bias_ld : 0) + col);
        }
      }
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        if (!ok_[it]) continue;
        const int idx = lane + 64 * it, row = idx / SEGS, seg = (idx % SEGS) * 4;
        float4 v = *reinterpret_cast<const float4*>(&my[row * ES + seg]);
        v.x += bv_[it].x; v.y += bv_[it].y; v.z += bv_[it].z; v.w += bv_[it].w;
        float4* o = reinterpret_cast<float4*>(p.out + orow_[it] * p.ldo + n0 + seg);
        if (p.accumulate) { const float4 ov = *o; v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w; }
        *o = v;
      }
    }
    return;
  }
  if (p.pool > 1) {
    // fused relu + max over p.pool in {2, 4} consecutive tile rows (round 6: the wide kernel too): the four accumulators a lane holds for one
    // column ARE four consecutive rows (D[row = 4 (l >> 4) + i][col = l & 15]), so the groups are folded in registers -- bias first (it differs
    // per vertex), first maximum wins, NaN propagates: relu_pool_kernel's rules -- and the lane stores one float per group (16 lanes = 64
    // contiguous bytes of a pooled row).  M % pool == 0 and the tile's row base is a multiple of 4: a group lies inside M as a whole or not at all.
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int64_t mb = m0 + wave * (16 * RT) + r * 16 + kg * 4;
      if (mb >= p.M) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + nt * 16 + r16;
        if (col >= p.N) continue;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[r][nt][i];
          if (p.bias_kind == 1 && col < p.bias_cols) v[i] += p.bias[col];
          else if (p.bias_kind == 2 && col < p.bias_cols && mb + i < p.M) v[i] += p.bias[((mb + i) % p.n_vertices) * p.bias_ld + col];     // (a lane's four rows may straddle two samples when n % 4 == 2; a group never does)
        }
        // (static indices only: a runtime-indexed v[] would live in scratch memory)
        auto later = [](float c, float best) { return c > best || (c != c && best == best); };
        auto emit = [&](int64_t m, float best, int bi) {
          const int64_t orow = m / p.pool;
          p.out[orow * p.ldo + col] = best > 0.f ? best : (best != best ? best : 0.f);
          if (p.pool_idx) p.pool_idx[orow * p.ldo + col] = (uint8_t)bi;
        };
        float b01 = v[0], b23 = v[2];
        int i01 = 0, i23 = 0;
        if (later(v[1], b01)) { b01 = v[1]; i01 = 1; }
        if (later(v[3], b23)) { b23 = v[3]; i23 = 1; }
        if (p.pool == 2) {
          emit(mb, b01, i01);
          if (mb + 2 < p.M) emit(mb + 2, b23, i23);
        } else {              // pool == 4: entries 0, 1, 2, 3 in order, the first maximum wins
          if (later(b23, b01)) { b01 = b23; i01 = 2 + i23; }
          emit(mb, b01, i01);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + wave * (16 * RT) + r * 16 + kg * 4 + i;
      if (m >= p.M) continue;
      const int64_t orow = proj_orow(p, m);
      const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + nt * 16 + r16;
        if (col >= p.N) continue;
        float v = acc[r][nt][i];
        if (p.bias_kind == 1 && col < p.bias_cols) v += p.bias[col];
        else if (p.bias_kind == 2 && col < p.bias_cols) v += p.bias[vert * p.bias_ld + col];
        float* o = p.out + orow * p.ldo + col;
        if (p.accumulate) v += *o;
        *o = v;
      }
    }
}

// The workgroups of the LAST, partly filled round of 256-row tiles (one workgroup per CU: 218 VGPRs at NT = 10) would leave most
// CUs idle for a whole tile time (90 k rows: 352 tiles = 256 + 96).  Their rows are handed out as 128-row tiles instead (one
// 16-row MFMA tile per wave): twice as many workgroups of about half the duration, so the tail takes ~0.55 instead of 1 tile time.
template <int NT>
__global__ __launch_bounds__(512) void project_x3v2_kernel(const ProjParams p, const int main_blocks) {
  constexpr int NW = NT * 16, WBUF = 3 * NW * 32, ES = NW + 4;
  constexpr int LDS_BYTES = (2 * WBUF * 2 > 8 * 16 * ES * 4) ? 2 * WBUF * 2 : 8 * 16 * ES * 4;
  __shared__ __align__(16) unsigned char lds_raw[LDS_BYTES];
  const int bx = (int)blockIdx.x, n0 = blockIdx.y * NW;
  if (bx < main_blocks) x3v2_body<NT, 2>(p, lds_raw, (int64_t)bx * 256, n0);
  else x3v2_body<NT, 1>(p, lds_raw, (int64_t)main_blocks * 256 + (int64_t)(bx - main_blocks) * 128, n0);
}

// ---- bf16x3, STREAMING form (round 5): rows of 32 / 64 floats, <= 64 output columns, the whole weight resident -- the shape of the
// row-mapped projections of the compacted forward (cfg5: 4.73 M compact rows x 5 terms, 5.27 M empty rows x 1 term, per time step).
// project_x3_kernel runs those at 4.0 / 3.3 TB/s: per 32-k tile a workgroup loads, waits, barriers, splits A and W into LDS, barriers and
// multiplies, so loads are in flight for a small part of each tile's time (46 line requests per CU on average, profiles/r04_projection_limiter.json),
// and the matrix, vector and LDS pipes (each ~30 % busy) take turns instead of overlapping.  Here nothing in the loop is shared between waves:
//   * the three bf16 planes of the WHOLE weight are split once per workgroup into LDS, already in MFMA fragment order (one conflict-free
//     ds_read_b128 per fragment), and stay there: no W traffic, no split and no barrier in the loop;
//   * the product is evaluated transposed, out^T = W^T x^T: the weight is the A operand (rows = output columns), a wave's 16 rows of x are the
//     B operand -- B[k = 8 (l >> 4) + j][col = l & 15] is 8 floats of row (l & 15), loaded global -> registers -> split -> MFMA with no LDS
//     round trip -- and the result D[row = 4 (l >> 4) + i][col = l & 15] leaves lane l with FOUR CONSECUTIVE output columns of its row:
//     bias loads and stores are 16-byte pieces straight from the accumulators, no epilogue scratch;
//   * the k slots of an MFMA are a permutation of the row's floats (slot (g, j) of k tile kt = float 32 kt + 16 (j >> 2) + 4 g + (j & 3),
//     the weight fragments are staged with the same map), so that the four lanes of a row read 64 contiguous bytes per load instruction;
//   * a wave walks its own 16-row tiles (persistent grid, one 1024-thread workgroup per CU), the samples of the pass and the terms in an
//     inner loop with the NEXT unit's loads issued before the current unit's 48 MFMAs, and a per-vertex bias row is read once per tile for
//     all samples of the pass (registers).
// Same arithmetic as project_x3_kernel (three-way split, six products, smallest first, fp32 accumulate over terms and k tiles in order);
// the order of the 32 products inside one MFMA differs (slot permutation), so results agree to fp32 rounding, not bit for bit.
template <int NT, int KTILES>
__global__ __launch_bounds__(1024) void project_x3_stream_kernel(const ProjParams p, const int64_t ntiles) {
  extern __shared__ __align__(16) unsigned char stream_smem[];
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  bf16x8* Wf = reinterpret_cast<bf16x8*>(stream_smem);        // [term][kt][nt][plane][lane]: 16 bytes per lane
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the tile counters below live in scalar registers (the <4, 2> form is at its 128-VGPR cap)
  const int r16 = lane & 15, g = lane >> 4;
  {
    const int nfrag = p.nterms * KTILES * NT * 64;
    for (int f = tid; f < nfrag; f += 1024) {
      const int fl = f & 63, blk = f >> 6;                      // blk = (term * KTILES + kt) * NT + nt
      const int fnt = blk % NT, fkt = (blk / NT) % KTILES, ft = blk / (NT * KTILES);
      const int n = fnt * 16 + (fl & 15), fg = fl >> 4;
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = fkt * 32 + 16 * (j >> 2) + 4 * fg + (j & 3);
        w[j] = (n < p.N) ? p.W[((int64_t)ft * p.Kc + k) * p.N + n] : 0.f;
      }
      unsigned pl[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split3(w[2 * j], w[2 * j + 1], pl[0][j], pl[1][j], pl[2][j]);
#pragma unroll
      for (int q = 0; q < 3; ++q) Wf[(blk * 3 + q) * 64 + fl] = __builtin_bit_cast(bf16x8, u32x4{pl[q][0], pl[q][1], pl[q][2], pl[q][3]});
    }
  }
  __syncthreads();
  const int64_t nwaves = (int64_t)gridDim.x * 16;
  const bool map_out = p.rowmap && !(p.mapped & kProjMapTermsOnly);
  const int units = p.nbatch * p.nterms;                        // (sample, term) units of one tile, sample-major

  // ---- per-tile row state: this lane's row in the tile order (clamped) and through the row map; the map entry of the NEXT tile is
  // loaded a whole tile ahead so that no unit's loads wait for it
  // (row numbers as 32-bit values: the host refuses M or n_vertices beyond int32 for this kernel; four 64-bit row registers cost the <4, 2>
  //  instantiation its last spill-free registers)
  // (round 6: the tile number is wave-uniform and lives in scalar registers, and a lane's clamped row of a tile is RECOMPUTED from it where
  //  it is needed -- two vector instructions -- instead of being carried across the tile loop for this and the next tile: the <4, 2>
  //  instantiation kept four loop-invariant registers in scratch before its loop)
  int64_t tile = (int64_t)blockIdx.x * 16 + wave;
  int32_t rrow = 0, rrow_next = 0;
  auto row_of = [&](int64_t t) -> int32_t { const int64_t m = t * 16 + r16; return (int32_t)(m < p.M ? m : p.M - 1); };
  auto ok_of = [&](int64_t t) -> bool { return t * 16 + r16 < p.M; };
  auto fetch_map = [&](int64_t t) -> int32_t {
    if (t >= ntiles) return 0;
    const int32_t m_c = row_of(t);
    return p.rowmap ? p.rowmap[m_c] : m_c;
  };
  float xa[KTILES * 8], xb[KTILES * 8];
  // unit u of the tile whose rows are (m_c, r_c): term u % nterms of sample u / nterms
  auto load_unit = [&](int u, int32_t m_c, int32_t r_c, float (&dst)[KTILES * 8]) {
    const int b = u / p.nterms, t = u - b * p.nterms;            // wave-uniform
    const int64_t arow = ((p.mapped >> t) & 1u) && p.rowmap ? r_c : m_c;
    const float* __restrict__ src = p.a[t] + (int64_t)b * p.a_bs[t] + arow * p.lda[t] + 4 * g;
#pragma unroll
    for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 v = *reinterpret_cast<const float4*>(src + kt * 32 + h * 16);
        dst[kt * 8 + h * 4 + 0] = v.x; dst[kt * 8 + h * 4 + 1] = v.y; dst[kt * 8 + h * 4 + 2] = v.z; dst[kt * 8 + h * 4 + 3] = v.w;
      }
  };
  f32x4 acc[NT];
  float4 bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bv[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_bias = [&](int64_t orow) {
    if (!p.bias_kind) return;
    const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
    const float* __restrict__ brow = p.bias + (p.bias_kind == 2 ? vert * p.bias_ld : 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = nt * 16 + 4 * g;
      bv[nt] = (col < p.bias_cols && col < p.N) ? *reinterpret_cast<const float4*>(brow + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // the 48 (KTILES = 2) MFMAs of one unit: per k tile split the lane's 8 floats into the three planes, then for pairs of column tiles
  // (two independent accumulator chains) the six products, smallest terms first
  auto compute_unit = [&](int u, const float (&src)[KTILES * 8], int64_t orow, bool ok) {
    const int b = u / p.nterms, t = u - b * p.nterms;
    if (t == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kt = 0; kt < KTILES; ++kt) {
      unsigned pl[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split3(src[kt * 8 + 2 * j], src[kt * 8 + 2 * j + 1], pl[0][j], pl[1][j], pl[2][j]);
      bf16x8 xs[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) xs[q] = __builtin_bit_cast(bf16x8, u32x4{pl[q][0], pl[q][1], pl[q][2], pl[q][3]});
      const bf16x8* wbase = Wf + ((t * KTILES + kt) * NT) * 3 * 64 + lane;
      constexpr int PAIR = NT >= 2 ? 2 : 1;
#pragma unroll
      for (int n0 = 0; n0 < NT; n0 += PAIR) {
        bf16x8 w[PAIR][3];
#pragma unroll
        for (int i = 0; i < PAIR; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q) w[i][q] = wbase[((n0 + i) * 3 + q) * 64];
        f32x4 c[PAIR];
#pragma unroll
        for (int i = 0; i < PAIR; ++i) c[i] = acc[n0 + i];
#pragma unroll
        for (int i = 0; i < PAIR; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][0], xs[2], c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < PAIR; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][1], xs[1], c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < PAIR; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][2], xs[0], c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < PAIR; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][0], xs[1], c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < PAIR; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][1], xs[0], c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < PAIR; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][0], xs[0], c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < PAIR; ++i) acc[n0 + i] = c[i];
      }
    }
    if (t == p.nterms - 1 && ok) {                            // last term of the sample: bias and store, 16 bytes per column tile
      float* __restrict__ o = p.out + (int64_t)b * p.out_bs + orow * p.ldo + 4 * g;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt * 16 + 4 * g >= p.N) continue;
        f32x4 v = acc[nt];
        v[0] += bv[nt].x; v[1] += bv[nt].y; v[2] += bv[nt].z; v[3] += bv[nt].w;
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o + nt * 16));        // written once, read by a later kernel
      }
    }
  };

  if (tile < ntiles) {
    rrow = fetch_map(tile);
    rrow_next = fetch_map(tile + nwaves);
    load_unit(0, row_of(tile), rrow, xa);
  }
  while (tile < ntiles) {
    const int64_t orow = map_out ? rrow : row_of(tile);
    const bool row_ok = ok_of(tile);
    load_bias(orow);
    const int64_t tile_n = tile + nwaves;
    // units of this tile two at a time (statically named register buffers); the unit after the tile's last one is unit 0 of the next tile
    for (int u = 0; u < units; u += 2) {
      if (u + 1 < units) load_unit(u + 1, row_of(tile), rrow, xb);
      else if (tile_n < ntiles) load_unit(0, row_of(tile_n), rrow_next, xb);
      compute_unit(u, xa, orow, row_ok);
      if (u + 1 >= units) {                                      // odd unit count: the prefetched unit belongs to the next tile -> move it to xa
#pragma unroll
        for (int i = 0; i < KTILES * 8; ++i) xa[i] = xb[i];
        break;
      }
      if (u + 2 < units) load_unit(u + 2, row_of(tile), rrow, xa);
      else if (tile_n < ntiles) load_unit(0, row_of(tile_n), rrow_next, xa);
      compute_unit(u + 1, xb, orow, row_ok);
    }
    tile = tile_n; rrow = rrow_next;
    rrow_next = fetch_map(tile + nwaves);
  }
}

// W-resident variant for the common case where the whole folded weight fits in LDS (nterms*Kc*N*4 <= 80 KB).
// 512 threads = 8 waves; the block loads W once, then every wave streams its own 32-row tiles:
//   global (float4, row-contiguous) -> registers -> wave-private LDS scratch [32][66] -> MFMA A fragments,
// with the next piece's global loads issued before the current piece's MFMAs.  No block barrier in the loop.
// W image: [term][k padded to 4][NT*16 columns], odd k rows have their 16-column halves swapped when the row
// is a multiple of 32 floats, so the B-fragment read (k, k+1 in one 32-lane group) is conflict-free.
constexpr int kResMaxThreads = 1024;
constexpr int kResKT = 64;             // floats of K per staged piece
constexpr int kResAS = kResKT + 2;     // scratch row stride (== 2 mod 32)
constexpr int kResScratchFloats = 8 * 32 * kResAS;   // wave-private A scratch in total: (512*2/RT threads / 64) waves x 16*RT rows
constexpr int kResMaxWBytes = 80 * 1024;

template <int NT>
__device__ __forceinline__ int w_col(int k, int n) {
  if constexpr ((NT & 1) == 0) return n ^ ((k & 1) << 4);
  else return n;
}

// RT = 16-row MFMA tiles per wave: 2 -> 8 waves x 32 rows (B fragments shared by two tiles), 1 -> 16 waves x 16 rows
// (4 waves per SIMD to cover LDS / global latency).
template <int NT, bool VEC4, int RT>
__global__ __launch_bounds__(1024 / RT) void project_resident_kernel(const ProjParams p, const int kc4, const int64_t ntiles) {
  constexpr int kResThreads = 1024 / RT, kResWaves = kResThreads / 64, kResRows = 16 * RT;
  extern __shared__ __align__(16) float smem[];
  constexpr int NW = NT * 16;
  float* Ws = smem;                                         // [nterms*kc4][NW]
  const int ktot = p.nterms * kc4;
  float* scratch = smem + (size_t)ktot * NW;                // [kResWaves][kResRows*kResAS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.y * NW;
  // ---- W -> LDS (once)
  for (int idx = tid; idx < ktot * NW; idx += kResThreads) {
    const int kk = idx / NW, cc = idx % NW;
    const int term = kk / kc4, kin = kk % kc4;
    float v = 0.f;
    if (kin < p.Kc && n0 + cc < p.N) v = p.W[((int64_t)term * p.Kc + kin) * p.N + n0 + cc];
    Ws[kk * NW + w_col<NT>(kk, cc)] = v;
  }
  __syncthreads();
  float* my = scratch + wave * (kResRows * kResAS);
  const int npieces = (p.Kc + kResKT - 1) / kResKT;
  const int total_pieces = p.nterms * npieces;

  for (int64_t tile = (int64_t)blockIdx.x * kResWaves + wave; tile < ntiles; tile += (int64_t)gridDim.x * kResWaves) {
    const int64_t m0 = tile * kResRows;
    f32x4 acc[RT][NT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int NRA = VEC4 ? 4 * RT : kResRows, NVA = VEC4 ? 4 : 1;
    float raA[NRA][NVA], raB[NRA][NVA];   // two pieces in flight (global -> registers) ahead of the one being multiplied
    // loads are unconditional (clamped address, value masked afterwards): no branch per load
    auto load_piece = [&](int pc, float (&ra)[NRA][NVA]) {
      const int term = pc / npieces, k0 = (pc % npieces) * kResKT;
      const float* __restrict__ A = p.a[term];
      const int64_t lda = p.lda[term];
      if constexpr (VEC4) {
#pragma unroll
        for (int i = 0; i < 4 * RT; ++i) {
          const int idx = lane + 64 * i, row = idx >> 4, kk = (idx & 15) * 4;
          const bool ok = (m0 + row < p.M) && (k0 + kk < p.Kc);
          const int64_t rr = (m0 + row < p.M) ? m0 + row : p.M - 1;
          const int kc = (k0 + kk < p.Kc) ? k0 + kk : 0;
          const float4 v = *reinterpret_cast<const float4*>(A + proj_arow(p, term, rr) * lda + kc);
          ra[i][0] = ok ? v.x : 0.f; ra[i][1] = ok ? v.y : 0.f; ra[i][2] = ok ? v.z : 0.f; ra[i][3] = ok ? v.w : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < kResRows; ++i) {
          const bool ok = (m0 + i < p.M) && (k0 + lane < p.Kc);
          const int64_t rr = (m0 + i < p.M) ? m0 + i : p.M - 1;
          const int kc = (k0 + lane < p.Kc) ? k0 + lane : 0;
          const float v = A[proj_row_off(p, proj_arow(p, term, rr), lda) + kc];
          ra[i][0] = ok ? v : 0.f;
        }
      }
    };
    auto store_piece = [&](const float (&ra)[NRA][NVA]) {
      if constexpr (VEC4) {
#pragma unroll
        for (int i = 0; i < 4 * RT; ++i) {
          const int idx = lane + 64 * i, row = idx >> 4, kk = (idx & 15) * 4;
          float2* d = reinterpret_cast<float2*>(&my[row * kResAS + kk]);
          d[0] = make_float2(ra[i][0], ra[i][1]);
          d[1] = make_float2(ra[i][2], ra[i][3]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < kResRows; ++i) my[i * kResAS + lane] = ra[i][0];
      }
    };
    const float* a0 = &my[(lane & 15) * kResAS + (lane >> 4)];
    // w_even / w_odd: this lane's row of the W image with the (lane-constant) column swizzle of even / odd
    // column tiles folded in: (nt*16 + c) ^ sw == nt*16 + c + (nt even ? sw : -sw)
    auto kstep = [&](int ks, const float* w_even, const float* w_odd) {
      float av[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) av[r] = a0[r * 16 * kResAS + ks * 4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = ((nt & 1) ? w_odd : w_even)[ks * 4 * NW + nt * 16];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv, acc[r][nt], 0, 0, 0);
      }
    };

    auto compute_piece = [&](int pc) {
      const int term = pc / npieces, k0 = (pc % npieces) * kResKT;
      const int ksteps = (min(kResKT, p.Kc - k0) + 3) >> 2;
      const int kbase = term * kc4 + k0 + (lane >> 4);
      // (kbase + 4*ks) & 1 == kbase & 1: the column swizzle is the same for every k-step of this lane
      const int sw = ((NT & 1) == 0) ? ((kbase & 1) << 4) : 0;
      const float* w_even = &Ws[kbase * NW + (lane & 15) + sw];
      const float* w_odd = &Ws[kbase * NW + (lane & 15) - sw];
      if (ksteps == kResKT / 4) {          // full piece: straight-line code so LDS reads run ahead of the MFMAs
#pragma unroll
        for (int ks = 0; ks < kResKT / 4; ++ks) kstep(ks, w_even, w_odd);
      } else {
        for (int ks = 0; ks < ksteps; ++ks) kstep(ks, w_even, w_odd);
      }
    };
    load_piece(0, raA);
    if (total_pieces > 1) load_piece(1, raB);
    for (int pc = 0; pc < total_pieces; pc += 2) {
      store_piece(raA);                    // previous piece's fragment reads were issued before (in-order LDS)
      if (pc + 2 < total_pieces) load_piece(pc + 2, raA);
      compute_piece(pc);
      if (pc + 1 < total_pieces) {
        store_piece(raB);
        if (pc + 3 < total_pieces) load_piece(pc + 3, raB);
        compute_piece(pc + 1);
      }
    }
    // ---- epilogue
    if (p.vec_epilogue) {
      // accumulators -> wave scratch (row-major) -> float4 rows: coalesced bias loads and 16-byte stores
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int i = 0; i < 4; ++i) my[(r * 16 + (lane >> 4) * 4 + i) * kResAS + nt * 16 + (lane & 15)] = acc[r][nt][i];
      constexpr int SEGS = NW / 4;                        // float4 per row
      constexpr int ITER = (kResRows * SEGS) / 64;
      if (p.pool > 1) {
        pooled_store(p, my, kResAS, kResRows, m0, n0, NW, lane);
        continue;                                         // next tile of this wave
      }
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int idx = lane + 64 * it, row = idx / SEGS, seg = (idx % SEGS) * 4;
        const int64_t m = m0 + row;
        const int col = n0 + seg;
        if (m >= p.M || col >= p.N) continue;
        const float2 lo = *reinterpret_cast<const float2*>(&my[row * kResAS + seg]);
        const float2 hi = *reinterpret_cast<const float2*>(&my[row * kResAS + seg + 2]);
        float4 v = make_float4(lo.x, lo.y, hi.x, hi.y);
        const int64_t orow = proj_orow(p, m);
        if (p.bias_kind && col < p.bias_cols) {
          const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
          const float4 bv = *reinterpret_cast<const float4*>(p.bias + (p.bias_kind == 2 ? vert * p.bias_ld : 0) + col);
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        float4* o = reinterpret_cast<float4*>(p.out + orow * p.ldo + col);
        if (p.accumulate) { const float4 ov = *o; v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w; }
        *o = v;
      }
    } else {
      const int col_l = lane & 15;
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t m = m0 + r * 16 + (lane >> 4) * 4 + i;
          if (m >= p.M) continue;
          const int64_t orow = proj_orow(p, m);
          const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + nt * 16 + col_l;
            if (col >= p.N) continue;
            float v = acc[r][nt][i];
            if (p.bias_kind == 1 && col < p.bias_cols) v += p.bias[col];
            else if (p.bias_kind == 2 && col < p.bias_cols) v += p.bias[vert * p.bias_ld + col];
            float* o = p.out + orow * p.ldo + col;
            if (p.accumulate) v += *o;
            *o = v;
          }
        }
    }
  }
}

// ---- narrow contraction (sum of Kc over the terms <= 16, e.g. one input channel per time step): the projection is a
// pure streaming write of (M, N) with a handful of scalars read per row, so it runs on the vector ALU.  W and the
// block's A values sit in LDS (A staged with coalesced loads, stored so that a thread's 4 rows are one 16-byte read);
// a thread owns 4 output columns of 4 rows per step; stores are whole 16-byte pieces of an output row.
// k-ordered fmaf chain per output, like the exact MFMA kernels.
constexpr int kNarrowMaxK = 16;
__global__ __launch_bounds__(kBlock) void project_narrow_kernel(const ProjParams p, int iters) {
  extern __shared__ float sW[];   // (ktot, N) weights, then (ktot, iters, RP, 4) A values
  const int ktot = p.nterms * p.Kc;
  const int L = p.N >> 2, RP = kBlock / L;
  const int rows_per_block = RP * 4 * iters;
  float* __restrict__ sA = sW + ktot * p.N;
  const int64_t mb0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t mend = (mb0 + rows_per_block < p.M) ? mb0 + rows_per_block : p.M;
  const int nrows = (int)(mend - mb0);
  for (int i = threadIdx.x; i < ktot * p.N; i += kBlock) sW[i] = p.W[i];
  for (int t = 0; t < p.nterms; ++t) {
    const int64_t ld = p.lda[t];
    if (p.rowmap && ((p.mapped >> t) & 1u)) {      // block-uniform: a mapped term (x of a compacted layer) is read through the row map
      const float* __restrict__ at = p.a[t];
      for (int i = threadIdx.x; i < nrows * p.Kc; i += kBlock) {
        const int l = i / p.Kc, kc = i - l * p.Kc;
        const int it = l / (4 * RP), rem = l - it * 4 * RP, j = rem / RP, r = rem - j * RP;
        sA[(((t * p.Kc + kc) * iters + it) * RP + r) * 4 + j] = at[proj_arow_il(p, t, mb0 + l) * ld + kc];
      }
    } else {
      const float* __restrict__ at = p.a[t] + mb0 * ld;
      for (int i = threadIdx.x; i < nrows * p.Kc; i += kBlock) {
        const int l = i / p.Kc, kc = i - l * p.Kc;                 // local row = (it * 4 + j) * RP + r
        const int it = l / (4 * RP), rem = l - it * 4 * RP, j = rem / RP, r = rem - j * RP;
        sA[(((t * p.Kc + kc) * iters + it) * RP + r) * 4 + j] = at[(int64_t)l * ld + kc];
      }
    }
  }
  __syncthreads();
  const int r_in = threadIdx.x / L, c4 = (threadIdx.x % L) * 4;
  if (r_in >= RP) return;
  for (int it = 0; it * 4 * RP < nrows; ++it) {
    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
    for (int kk = 0; kk < ktot; ++kk) {
      const float4 w = *reinterpret_cast<const float4*>(sW + kk * p.N + c4);
      const float4 a4 = *reinterpret_cast<const float4*>(sA + ((kk * iters + it) * RP + r_in) * 4);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j][0] = fmaf(av[j], w.x, acc[j][0]);
        acc[j][1] = fmaf(av[j], w.y, acc[j][1]);
        acc[j][2] = fmaf(av[j], w.z, acc[j][2]);
        acc[j][3] = fmaf(av[j], w.w, acc[j][3]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t m = mb0 + (it * 4 + j) * RP + r_in;
      if (m >= mend) continue;
      const int64_t orow = p.rowmap ? proj_orow_il(p, m) : proj_orow(p, m);
      float4 v = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
      if (p.bias_kind && c4 < p.bias_cols) {
        const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + (p.bias_kind == 2 ? vert * p.bias_ld : 0) + c4);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      }
      float4* o = reinterpret_cast<float4*>(p.out + orow * p.ldo + c4);
      if (p.accumulate) { const float4 ov = *o; v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w; }
      using f4 = __attribute__((ext_vector_type(4))) float;
      __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(o));   // written once, read by a later kernel
    }
  }
}
