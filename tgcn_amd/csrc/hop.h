// hop.h -- S = L X, Y = alpha S + beta Z (+ gamma Z2): hop_kernel, hop_fixup_kernel, their geometry and launch helpers
// Part of the single translation unit tgcn_hip.hip (included once, inside its anonymous namespace).
#pragma once

// --------------------------------------------------------------------------------------------------
// hop
// --------------------------------------------------------------------------------------------------
struct HopParams {
  const int32_t* rowptr;
  const tgcn_edge* ev;
  const int32_t* blk_row;
  const int32_t* seg_row;
  const int32_t* seg_e0;
  const int32_t* seg_e1;
  const int32_t* seg_slot;
  const int32_t* long_row;
  const int32_t* long_slot;
  const float* X;
  const float* Z;
  const float* Z2;
  float* Y;
  float* P;
  float* partial;
  int64_t x_bs, x_ld, z_bs, z_ld, z2_bs, z2_ld, y_bs, y_ld, p_bs, p_ld;
  float alpha, beta, gamma;
  int32_t nblk, nseg, nlong, nhuge, row_thresh;
  int32_t C, nb, nchunks, cpad, remap;
  int32_t nwseg;                 // leading segments that are whole rows of up to 32 * (64 / LPR) entries, one WAVE each (no partial rows)
  int32_t mix_period;            // kernel: > 1 one row block every mix_period block ids, < 0 segment blocks first, 0 / 1 row blocks first (host: the request, see launch_hop)
  int32_t stream_out;            // the output tensor is larger than the Infinity Cache: entries, results and partial rows with non-temporal hints
  int32_t seg_mode, seg_remap;   // seg_mode 1: one WAVE per segment (tgcn_csr_sched.seg_mode); seg_remap: XCD-contiguous segment ranges
  int32_t long_rows_only;        // host only: the launch covers the rows above row_thresh only (fused last hop) -- names the profile record
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}

template <int VEC>
__device__ __forceinline__ void load_vec_nt(const float* __restrict__ p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    using f4 = __attribute__((ext_vector_type(4))) float;
    const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = __builtin_nontemporal_load(p);
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    *p = v[0];
  }
}

// XCD-aware block id: blocks b and b+8 share an XCD (observed round-robin dispatch), so hand each XCD a
// contiguous range of row blocks -- neighbouring rows share neighbour columns in its private L2.
// Bijective for every nblk (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Tuning bits of the hop kernel (NTM): which accesses carry the non-temporal hint, and ev prefetch.
constexpr int kNtEdges = 1, kNtStores = 2, kNtPartials = 4;
// kNtColdGather (round-5 experiment, tools/hop_bench.py --cold-last RANK --cold-nt): an entry whose column has bit 31 set points at a COLD row
// of X (outside the RANK most referenced); its gather carries the non-temporal hint so that the lines that miss anyway do not evict the hub
// rows from the 4-MB L2s.  The flag is part of the experiment's operand, not of the ABI: no shipped operand sets it.
constexpr int kNtColdGather = 8;

template <int VEC>
__device__ __forceinline__ void store_vec_nt(float* __restrict__ p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    using f4 = __attribute__((ext_vector_type(4))) float;
    __builtin_nontemporal_store(f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4*>(p));
  } else {
    __builtin_nontemporal_store(v[0], p);
  }
}

template <int NTM>
__device__ __forceinline__ void load_edge(const tgcn_edge* __restrict__ ev, int e, int& c, float& v) {
  if constexpr (NTM & kNtEdges) {
    using i2 = __attribute__((ext_vector_type(2))) int;
    const i2 t = __builtin_nontemporal_load(reinterpret_cast<const i2*>(ev + e));
    c = t.x;
    v = __int_as_float(t.y);
  } else {
    const tgcn_edge t = ev[e];
    c = t.col;
    v = t.val;
  }
}

template <int VEC, int NTM>
__device__ __forceinline__ void finish_row(const HopParams& p, int b, int r, int c0, const float (&s)[VEC]) {
  if (p.P) {
    if constexpr (NTM & kNtStores) store_vec_nt<VEC>(p.P + (int64_t)b * p.p_bs + (int64_t)r * p.p_ld + c0, s);
    else store_vec<VEC>(p.P + (int64_t)b * p.p_bs + (int64_t)r * p.p_ld + c0, s);
  }
  float y[VEC];
  if (p.Z) {
    float z[VEC];
    load_vec_nt<VEC>(p.Z + (int64_t)b * p.z_bs + (int64_t)r * p.z_ld + c0, z);
#pragma unroll
    for (int i = 0; i < VEC; ++i) y[i] = fmaf(p.alpha, s[i], p.beta * z[i]);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) y[i] = p.alpha * s[i];
  }
  if (p.Z2) {   // second addend (Clenshaw step of the project-first path): y += gamma * z2
    float z2[VEC];
    load_vec_nt<VEC>(p.Z2 + (int64_t)b * p.z2_bs + (int64_t)r * p.z2_ld + c0, z2);
#pragma unroll
    for (int i = 0; i < VEC; ++i) y[i] = fmaf(p.gamma, z2[i], y[i]);
  }
  if (p.Y) {
    if constexpr (NTM & kNtStores) store_vec_nt<VEC>(p.Y + (int64_t)b * p.y_bs + (int64_t)r * p.y_ld + c0, y);
    else store_vec<VEC>(p.Y + (int64_t)b * p.y_bs + (int64_t)r * p.y_ld + c0, y);
  }
}

// Value of lane `lane` of every LPR-lane group (lane is a constant once the caller's loops are unrolled): a DPP row
// broadcast for 16-lane groups, a quad permute for 4-lane groups, v_readlane for whole-wave groups -- one vector-ALU
// instruction instead of a trip through the LDS crossbar (ds_bpermute) in the entry -> gather chain.
template <int LPR>
__device__ __forceinline__ int group_bcast(int v, int lane) {
  if constexpr (LPR == 64) {
    return __builtin_amdgcn_readlane(v, lane);
  } else if constexpr (LPR == 16) {
#define TGCN_RB(N) case N: return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xF, 0xF, false);   /* row_newbcast:N */
    switch (lane) {
      TGCN_RB(0) TGCN_RB(1) TGCN_RB(2) TGCN_RB(3) TGCN_RB(4) TGCN_RB(5) TGCN_RB(6) TGCN_RB(7)
      TGCN_RB(8) TGCN_RB(9) TGCN_RB(10) TGCN_RB(11) TGCN_RB(12) TGCN_RB(13) TGCN_RB(14) default: return __builtin_amdgcn_update_dpp(0, v, 0x15F, 0xF, 0xF, false);
    }
#undef TGCN_RB
  } else if constexpr (LPR == 4) {
    switch (lane) {
      case 0: return __builtin_amdgcn_update_dpp(0, v, 0x00, 0xF, 0xF, false);    // quad_perm:[0,0,0,0]
      case 1: return __builtin_amdgcn_update_dpp(0, v, 0x55, 0xF, 0xF, false);
      case 2: return __builtin_amdgcn_update_dpp(0, v, 0xAA, 0xF, 0xF, false);
      default: return __builtin_amdgcn_update_dpp(0, v, 0xFF, 0xF, 0xF, false);
    }
  } else {
    return __shfl(v, lane, LPR);
  }
}

// Sum of val_e * X[col_e, c0..c0+VEC) over stored entries [e0[rr], e1[rr]) of R rows (or segments) at once, by one
// group of LPR lanes.  Per row the group reads LPR entries with one coalesced 8-byte load per lane and hands them
// round with in-register broadcasts (group_bcast); gathers are issued U at a time per row, so R*U 16-byte loads are in
// flight per lane.  R > 1 keeps R independent rowptr -> entry -> gather chains going, which is what low-degree rows on wide
// operands need (measured on the mesh config); entries are summed in stored order: deterministic.
template <int LPR, int VEC, int UU, int R, int NTM>
__device__ __forceinline__ void accum_multi(const tgcn_edge* __restrict__ ev, const int (&e0)[R], const int (&e1)[R], int t,
                                            const float* __restrict__ Xc, int64_t ldx, float (&acc)[R][VEC]) {
  constexpr int U = LPR < UU ? LPR : UU;
  int len_max = 0;
#pragma unroll
  for (int rr = 0; rr < R; ++rr) len_max = max(len_max, e1[rr] - e0[rr]);
  for (int off = 0; off < len_max; off += LPR) {
    int my_c[R], cnt[R];
    float my_v[R];
    int cmax = 0;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      my_c[rr] = 0;
      my_v[rr] = 0.f;
      const int e = e0[rr] + off + t;
      if (e < e1[rr]) load_edge<NTM>(ev, e, my_c[rr], my_v[rr]);
      cnt[rr] = min(LPR, max(0, e1[rr] - e0[rr] - off));
      cmax = max(cmax, cnt[rr]);
    }
#pragma unroll
    for (int j0 = 0; j0 < LPR; j0 += U) {        // fully unrolled: the broadcast lane is an immediate
      if (j0 < cmax) {
        float xv[R][U][VEC];
        float vv[R][U];
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int c = group_bcast<LPR>(my_c[rr], j0 + u);
            vv[rr][u] = __int_as_float(group_bcast<LPR>(__float_as_int(my_v[rr]), j0 + u));   // 0 past the end of the row
#pragma unroll
            for (int i = 0; i < VEC; ++i) xv[rr][u][i] = 0.f;
            if constexpr (NTM & kNtColdGather) {
              if (j0 + u < cnt[rr]) {
                if (c < 0) load_vec_nt<VEC>(Xc + (int64_t)(c & 0x7fffffff) * ldx, xv[rr][u]);
                else load_vec<VEC>(Xc + (int64_t)c * ldx, xv[rr][u]);
              }
            } else {
              if (j0 + u < cnt[rr]) load_vec<VEC>(Xc + (int64_t)c * ldx, xv[rr][u]);
            }
          }
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[rr][i] = fmaf(vv[rr][u], xv[rr][u][i], acc[rr][i]);
      }
    }
  }
}

template <int LPR, int VEC, int UU, int R, int NTM>
__global__ __launch_bounds__(kBlock) void hop_kernel(const HopParams p) {
  constexpr int GPB = kBlock / LPR;
  const int tid = threadIdx.x;
  const int t = tid % LPR;
  const int gib = tid / LPR;
  const int chunk = blockIdx.y % p.nchunks;
  const int b = blockIdx.y / p.nchunks;
  const int c0 = (chunk * LPR + t) * VEC;
  const bool cact = c0 < p.C;
  const float* Xc = p.X + (int64_t)b * p.x_bs + (cact ? c0 : 0);
  int bid = blockIdx.x;
  if (p.mix_period > 1) {
    // row blocks dealt evenly among the segment blocks (one every mix_period ids) instead of all in front: the short rows wait on
    // HBM, the segments are bound by the gather path into the CUs -- side by side on a CU they fill each other's gaps
    const int q = bid / p.mix_period;
    if (bid % p.mix_period == 0 && q < p.nblk) bid = q;                                              // row block q
    else bid = p.nblk + (bid - min(p.nblk, (bid + p.mix_period - 1) / p.mix_period));                // segment block, in order
  }
  else if (p.mix_period < 0) {               // segment blocks first, row blocks behind them
    const int nsb = (int)gridDim.x - p.nblk;
    bid = bid < nsb ? p.nblk + bid : bid - nsb;
  }
  if (bid < p.nblk) {
    if (p.remap) bid = xcd_remap(bid, p.nblk);
    const int r0 = p.blk_row[bid], r1 = p.blk_row[bid + 1];
    for (int rb = r0 + gib; rb < r1; rb += GPB * R) {
      int e0[R], e1[R];
      bool live[R];
      float acc[R][VEC];
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const int r = rb + rr * GPB;
        e0[rr] = e1[rr] = 0;
        if (r < r1) { e0[rr] = p.rowptr[r]; e1[rr] = p.rowptr[r + 1]; }
        live[rr] = (r < r1) && (e1[rr] - e0[rr] <= p.row_thresh);
        if (!live[rr]) e1[rr] = e0[rr];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[rr][i] = 0.f;
      }
      accum_multi<LPR, VEC, UU, R, NTM>(p.ev, e0, e1, t, Xc, p.x_ld, acc);
#pragma unroll
      for (int rr = 0; rr < R; ++rr)
        if (live[rr] && cact) finish_row<VEC, NTM>(p, b, rb + rr * GPB, c0, acc[rr]);
    }
  } else {
    int sbid = bid - p.nblk;
    if (p.seg_remap) sbid = xcd_remap(sbid, (int)gridDim.x - p.nblk);
    if constexpr (LPR < 64 && R == 1) {
      const int nwblk = (p.nwseg + kBlock / 64 - 1) / (kBlock / 64);     // workgroups of the whole-row wave segments (hybrid schedule)
      if (p.seg_mode == 1 || sbid < nwblk) {
        // One wave per segment of up to 32 * (64 / LPR) entries: its lane groups take consecutive pieces of the segment and the
        // pieces are folded inside the wave (fixed order: neighbours first), so a row of up to that many entries is written
        // directly and longer rows leave one partial row per wave instead of one per lane group.
        constexpr int GPW = 64 / LPR;
        const int s = sbid * (kBlock / 64) + (tid >> 6);
        const int gw = (tid & 63) / LPR;
        int e0[1] = {0}, e1[1] = {0};
        const int wlimit = p.seg_mode == 1 ? p.nseg : p.nwseg;
        if (s < wlimit) {
          const int a = p.seg_e0[s], z = p.seg_e1[s];
          const int per = (z - a + GPW - 1) / GPW;
          e0[0] = min(z, a + gw * per);
          e1[0] = min(z, e0[0] + per);
        }
        float acc[1][VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[0][i] = 0.f;
        accum_multi<LPR, VEC, UU, 1, NTM>(p.ev, e0, e1, t, Xc, p.x_ld, acc);
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[0][i] += __shfl_xor(acc[0][i], off, 64);
        if (s >= wlimit || gw != 0) return;
        const int slot = p.seg_slot[s];
        if (slot < 0) {
          if (cact) finish_row<VEC, NTM>(p, b, p.seg_row[s], c0, acc[0]);
        } else {
          store_vec<VEC>(p.partial + ((int64_t)slot * p.nb + b) * p.cpad + (chunk * LPR + t) * VEC, acc[0]);
        }
        return;
      }
      sbid -= nwblk;
    }
    const int sb = p.nwseg + sbid * GPB * R + gib;
    int e0[R], e1[R];
    float acc[R][VEC];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int s = sb + rr * GPB;
      e0[rr] = e1[rr] = 0;
      if (s < p.nseg) { e0[rr] = p.seg_e0[s]; e1[rr] = p.seg_e1[s]; }
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[rr][i] = 0.f;
    }
    accum_multi<LPR, VEC, UU, R, NTM>(p.ev, e0, e1, t, Xc, p.x_ld, acc);
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int s = sb + rr * GPB;
      if (s >= p.nseg) continue;
      const int slot = p.seg_slot[s];
      if (slot < 0) {
        if (cact) finish_row<VEC, NTM>(p, b, p.seg_row[s], c0, acc[rr]);
      } else {
        float* dst = p.partial + ((int64_t)slot * p.nb + b) * p.cpad + (chunk * LPR + t) * VEC;
        if constexpr (NTM & kNtPartials) store_vec_nt<VEC>(dst, acc[rr]);
        else store_vec<VEC>(dst, acc[rr]);
      }
    }
  }
}

// Folds the partial sums of rows that were cut into several segments, in slot order (deterministic).
// Blocks [0, nhuge): one row each, the block's groups sum interleaved slots and combine through LDS in group
// order; the remaining blocks: one row per lane group.
template <int LPR, int VEC>
__global__ __launch_bounds__(kBlock) void hop_fixup_kernel(const HopParams p) {
  constexpr int GPB = kBlock / LPR;
  constexpr int UF = 4;
  __shared__ float red[GPB * LPR * VEC];
  const int tid = threadIdx.x;
  const int t = tid % LPR;
  const int gib = tid / LPR;
  const int chunk = blockIdx.y % p.nchunks;
  const int b = blockIdx.y / p.nchunks;
  const int c0 = (chunk * LPR + t) * VEC;
  const bool huge = (int)blockIdx.x < p.nhuge;
  const int i = huge ? (int)blockIdx.x : p.nhuge + ((int)blockIdx.x - p.nhuge) * GPB + gib;
  const bool valid = i < p.nlong;
  const int row = valid ? p.long_row[i] : 0;
  const int s0 = valid ? p.long_slot[i] : 0, s1 = valid ? p.long_slot[i + 1] : 0;
  const int first = huge ? s0 + gib : s0, step = huge ? GPB : 1;
  const float* base = p.partial + (int64_t)b * p.cpad + c0;
  const int64_t sstride = (int64_t)p.nb * p.cpad;
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  int s = first;
  for (; s + (UF - 1) * step < s1; s += UF * step) {
    float v[UF][VEC];
#pragma unroll
    for (int u = 0; u < UF; ++u) load_vec_nt<VEC>(base + (int64_t)(s + u * step) * sstride, v[u]);
#pragma unroll
    for (int u = 0; u < UF; ++u)
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += v[u][k];
  }
  for (; s < s1; s += step) {
    float v[VEC];
    load_vec_nt<VEC>(base + (int64_t)s * sstride, v);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += v[k];
  }
  if (huge) {  // block-uniform branch
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[(gib * LPR + t) * VEC + k] = acc[k];
    __syncthreads();
    if (gib != 0) return;
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int g = 0; g < GPB; ++g)
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += red[(g * LPR + t) * VEC + k];
  }
  if (valid && c0 < p.C) finish_row<VEC, 0>(p, b, row, c0, acc);
}

struct HopGeom {
  int vec, lpr, nchunks, cpad;
};

inline HopGeom hop_geom(int32_t C, int aligned16) {
  HopGeom g;
  g.vec = (aligned16 && (C % 4 == 0)) ? 4 : 1;
  const int lanes = (C + g.vec - 1) / g.vec;
  int lpr = 1;
  while (lpr < lanes && lpr < 64) lpr <<= 1;
  g.lpr = lpr;
  g.nchunks = (lanes + lpr - 1) / lpr;
  g.cpad = g.nchunks * lpr * g.vec;
  return g;
}

template <int LPR, int VEC, int U, int R, int NTM = 0>
inline void launch_hop(hipStream_t st, const HopParams& p, dim3 grid) {
  constexpr int GPB = kBlock / LPR;
  const int seg_per_block = (p.seg_mode == 1 && LPR < 64 && R == 1) ? kBlock / 64 : GPB * R;
  const int nw = (LPR < 64 && R == 1 && p.seg_mode != 1) ? p.nwseg : 0;
  const int nsb = (nw + kBlock / 64 - 1) / (kBlock / 64) + (p.nseg - nw + seg_per_block - 1) / seg_per_block;
  grid.x = (unsigned)(p.nblk + nsb);
  HopParams pk = p;
  // p.mix_period arrives as a request (tgcn_csr_hop_f32); the period follows from THIS kernel's segment-block count
  if (p.mix_period > 0) pk.mix_period = (p.nblk > 0 && nsb >= p.nblk && LPR < 64) ? (p.nblk + nsb) / p.nblk : 0;
  else if (p.mix_period < 0) pk.mix_period = nsb > 0 ? -1 : 0;
  // "hop_lds_pad": unused dynamic LDS per workgroup = an occupancy limiter (160 KB / pad workgroups per CU) for A/B runs
  const int pad = g_hop_lds_pad.load();
  if (pad > 65536) allow_large_lds((const void*)hop_kernel<LPR, VEC, U, R, NTM>, pad);
  hipLaunchKernelGGL((hop_kernel<LPR, VEC, U, R, NTM>), grid, dim3(kBlock), (size_t)(pad > 0 ? pad : 0), st, pk);
}

// developer variants of the two float4 shapes that matter for the benchmarks (tools/hop_bench.py)
inline bool launch_hop_variant(int lpr, hipStream_t st, const HopParams& p, dim3 grid) {
  const int v = g_hop_variant.load();
  if (lpr == 16) {
    switch (v) {
      case 1: launch_hop<16, 4, 4, 1>(st, p, grid); return true;
      case 2: launch_hop<16, 4, 8, 1, kNtEdges>(st, p, grid); return true;
      case 3: launch_hop<16, 4, 8, 1, kNtEdges | kNtStores>(st, p, grid); return true;
      case 4: launch_hop<16, 4, 8, 1, 0>(st, p, grid); return true;
      case 5: launch_hop<16, 4, 8, 1, kNtEdges | kNtStores | kNtPartials | kNtColdGather>(st, p, grid); return true;   // experiment: nt gathers of flagged (cold) columns
      case 6: launch_hop<16, 4, 8, 1, kNtEdges | kNtStores | kNtPartials>(st, p, grid); return true;                   // the shipped streaming form, forced
      default: return false;
    }
  }
  if (lpr == 4) {
    switch (v) {
      case 1: launch_hop<4, 4, 8, 1>(st, p, grid); return true;
      case 2: launch_hop<4, 4, 4, 2>(st, p, grid); return true;
      case 3: launch_hop<4, 4, 4, 4>(st, p, grid); return true;
      case 4: launch_hop<4, 4, 2, 4>(st, p, grid); return true;
      default: return false;
    }
  }
  if (lpr == 64) {
    switch (v) {
      case 1: launch_hop<64, 4, 4, 1>(st, p, grid); return true;
      case 2: launch_hop<64, 4, 4, 2>(st, p, grid); return true;
      case 3: launch_hop<64, 4, 8, 2>(st, p, grid); return true;
      case 4: launch_hop<64, 4, 8, 1>(st, p, grid); return true;
      default: return false;
    }
  }
  return false;
}

// rows interleaved per lane group: wide operands (a whole wave per row chunk) run 4 rows at once
template <int L> struct HopRows { static constexpr int value = (L == 64) ? 4 : 1; };
// gathers in flight per lane and row: 8 for 16-lane groups (cfg5 on the compacted operand: 3.840 -> 3.804 ms, four A/B runs), 4 otherwise
template <int L> struct HopUnroll { static constexpr int value = (L == 16) ? 8 : 4; };

template <int VEC>
int launch_hop_vec(hipStream_t st, const HopParams& p, int lpr, dim3 grid, dim3 fix_grid) {
#define TGCN_HOP_CASE(L)                                                                    \
  case L: {                                                                                 \
    { ProfScope ps(p.long_rows_only ? TGCN_PROF_HOP_LONG : TGCN_PROF_HOP, st);              \
      if (!(VEC == 4 && g_hop_variant.load() != 0 && launch_hop_variant(L, st, p, grid))) { \
        /* interleave rows only when the grid still fills the chip afterwards */            \
        if (HopRows<L>::value > 1 && (int64_t)p.nblk * grid.y >= 4096)                       \
          launch_hop<L, VEC, 4, HopRows<L>::value>(st, p, grid);                            \
        else if (L == 16 && VEC == 4 && p.stream_out)                                       \
          /* once-read entries and once-written rows keep out of the L2 that holds the hub rows of X (cfg5: 3.79 -> 3.70 ms) */ \
          launch_hop<L, VEC, HopUnroll<L>::value, 1, kNtEdges | kNtStores | kNtPartials>(st, p, grid); \
        else launch_hop<L, VEC, HopUnroll<L>::value, 1>(st, p, grid);                       \
      } }                                                                                   \
    if (p.nlong > 0) { ProfScope ps(TGCN_PROF_HOP_FIXUP, st);                               \
      hipLaunchKernelGGL((hop_fixup_kernel<L, VEC>), fix_grid, dim3(kBlock), 0, st, p); }   \
  } break;
  switch (lpr) {
    TGCN_HOP_CASE(1)
    TGCN_HOP_CASE(2)
    TGCN_HOP_CASE(4)
    TGCN_HOP_CASE(8)
    TGCN_HOP_CASE(16)
    TGCN_HOP_CASE(32)
    TGCN_HOP_CASE(64)
    default:
      TGCN_FAIL(TGCN_ERR_INVALID, "hop: bad lanes_per_row %d", lpr);
  }
#undef TGCN_HOP_CASE
  TGCN_CHECK_LAUNCH("tgcn_csr_hop_f32");
  return TGCN_OK;
}

inline bool aligned4(const tgcn_dense* d) {
  return d == nullptr || d->ptr == nullptr ||
         (((uintptr_t)d->ptr & 15) == 0 && (d->batch_stride & 3) == 0 && (d->row_stride & 3) == 0);
}
