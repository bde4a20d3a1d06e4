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
  int32_t C, nb, nchunks, cpad;
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
constexpr int kNtEdges = 1, kNtStores = 2;

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
            if (j0 + u < cnt[rr]) load_vec<VEC>(Xc + (int64_t)c * ldx, xv[rr][u]);
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
  if (bid < p.nblk) {
    bid = xcd_remap(bid, p.nblk);
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
    const int sb = (bid - p.nblk) * GPB * R + gib;
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
        store_vec<VEC>(p.partial + ((int64_t)slot * p.nb + b) * p.cpad + (chunk * LPR + t) * VEC, acc[rr]);
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

// --------------------------------------------------------------------------------------------------
// sweep: long rows of large operands, accumulators in LDS, entries walked in order of column popularity
// --------------------------------------------------------------------------------------------------
// (include/tgcn_hip.h, tgcn_csr_sched ABI v3.)  One 1024-thread workgroup per CU holds the accumulators of up to
// 8 * (1024 / lanes_per_row) rows in LDS.  The entries of those rows are sorted by (column popularity panel, row, column
// popularity) and dealt to the lane groups in chunks of lanes_per_row entries, round robin: every lane group of every
// resident workgroup is then in the same popularity panel at about the same time, so a row of X fetched by one of them is
// served to the others by the XCD's L2.  (A first form that gave every lane group its own rows did not hold that lockstep:
// 32 % L2 hits and 24 GB fetched per launch on the 160 M-entry R-MAT, against 47 % / 19.5 GB for the column-ordered
// segments; tools/sim models both.)  Consecutive entries of one row are summed in registers and added to the row's slot
// with no-return LDS float adds when the row changes: several lane groups add to one slot, so the order of those adds -- and
// with it the last bits of the result -- is not fixed run to run (the reference's scatter_add on a GPU is no different,
// gcn.py:308,343); the tests bound it by the 1e-5 parity tolerance.
struct SweepParams {
  const tgcn_edge* ent;
  const int16_t* slot;
  const int32_t* gptr;
  const int32_t* pptr;       // [streams * nbar] end of the first nbar popularity panels inside every stream (round 0 only uses them)
  const int32_t* slot_row;
  int32_t* sync;             // [32] zeroed per launch: sync[0] arrival counter, sync[16] give-up flag
  int32_t rounds, nwg, nbar;
};

constexpr int kSweepBlock = 1024, kSweepSlotsPerGroup = 8;

// byte-free word offset of the 4 words [w, w+4) (w a multiple of 4) of slot sl: rows of >= 64 words are rotated by 16 words per
// slot so that the four lane groups of a wave, adding to four different slots, use different LDS banks
template <int ROWF>
__device__ __forceinline__ int sweep_word(int sl, int w) {
  if constexpr (ROWF >= 64) return sl * ROWF + ((w + 16 * (sl & 3)) & (ROWF - 1));
  else return sl * ROWF + w;
}

// Timing-only rendezvous of all workgroups of the launch (speed, never correctness: no data is handed over, so relaxed
// atomics and no fences): the k-th call returns when gridDim.x * k workgroups have arrived, or after a bounded spin, after
// which every later call of every workgroup returns at once.
__device__ __forceinline__ void sweep_rendezvous(int32_t* sync, int k, int nwg_total) {
  __syncthreads();
  if (threadIdx.x == 0 && __hip_atomic_load(sync + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    (void)__hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int target = nwg_total * k;
    int spins = 0;
    while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > 20000 || __hip_atomic_load(sync + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {   // ~ a millisecond: a workgroup is not resident
        __hip_atomic_store(sync + 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}

// One chunk of LPR entries of a lane group's stream.  FULL: all LPR entries exist (no per-entry predicate).  RUNS: consecutive
// entries of one row are summed in registers and added to the row's slot when the row changes (hot and cold panels: a row has
// many entries there); otherwise every product goes straight to its slot (warm panels: a row has one or two entries per panel,
// so run bookkeeping would cost more than it saves).
template <int LPR, int U, bool FULL, bool RUNS, int ROWF>
__device__ __forceinline__ void sweep_chunk(const float* __restrict__ Xc, int64_t ldx, float* __restrict__ lds, int c0, int my_c, float my_v,
                                            int my_s, int cnt, int& cur, float (&acc)[4]) {
#pragma unroll
  for (int j0 = 0; j0 < LPR; j0 += U) {              // fully unrolled: the broadcast lane is an immediate
    if (FULL || j0 < cnt) {
      float xv[U][4];
      float vv[U];
      int ss[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = group_bcast<LPR>(my_c, j0 + u);
        vv[u] = __int_as_float(group_bcast<LPR>(__float_as_int(my_v), j0 + u));
        ss[u] = group_bcast<LPR>(my_s, j0 + u);
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[u][i] = 0.f;
        if (FULL || j0 + u < cnt) load_vec<4>(Xc + (int64_t)c * ldx, xv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (FULL || j0 + u < cnt) {
          if constexpr (RUNS) {
            if (ss[u] != cur) {                          // same for the lanes of a group: the next row of this chunk
              float* sl = lds + sweep_word<ROWF>(cur, c0);
#pragma unroll
              for (int i = 0; i < 4; ++i) (void)__hip_atomic_fetch_add(sl + i, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              cur = ss[u];
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(vv[u], xv[u][i], acc[i]);
          } else {
            float* sl = lds + sweep_word<ROWF>(ss[u], c0);
#pragma unroll
            for (int i = 0; i < 4; ++i) (void)__hip_atomic_fetch_add(sl + i, vv[u] * xv[u][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
  }
}

// entries [e0, e1) of one stream; the next chunk's entries are loaded under this chunk's gathers
template <int LPR, int U, bool RUNS, int ROWF, int NTM>
__device__ __forceinline__ void sweep_range(const SweepParams& s, const float* __restrict__ Xc, int64_t ldx, float* __restrict__ lds, int c0, int t,
                                            int e0, int e1, int& cur, float (&acc)[4]) {
  int nx_c = 0, nx_s = 0;
  float nx_v = 0.f;
  if (e0 + t < e1) { load_edge<NTM>(s.ent, e0 + t, nx_c, nx_v); nx_s = s.slot[e0 + t]; }
  int e = e0;
  for (; e + LPR <= e1; e += LPR) {
    const int my_c = nx_c, my_s = nx_s;
    const float my_v = nx_v;
    nx_c = 0; nx_s = 0; nx_v = 0.f;
    if (e + LPR + t < e1) { load_edge<NTM>(s.ent, e + LPR + t, nx_c, nx_v); nx_s = s.slot[e + LPR + t]; }
    sweep_chunk<LPR, U, true, RUNS, ROWF>(Xc, ldx, lds, c0, my_c, my_v, my_s, LPR, cur, acc);
  }
  if (e < e1) sweep_chunk<LPR, U, false, RUNS, ROWF>(Xc, ldx, lds, c0, nx_c, nx_v, nx_s, e1 - e, cur, acc);
}

template <int LPR, int NTM, int UU>
__global__ __launch_bounds__(kSweepBlock) void hop_sweep_kernel(const HopParams p, const SweepParams s) {
  constexpr int VEC = 4, G = kSweepBlock / LPR, SPG = kSweepSlotsPerGroup, SLOTS = G * SPG, ROWF = LPR * VEC;
  constexpr int U = LPR < UU ? LPR : UU;   // row loads in flight per lane: one workgroup per CU, so 8 (128 KB per CU)
  extern __shared__ float sweep_acc[];                   // SLOTS x ROWF floats = 128 KB
  const int tid = threadIdx.x;
  const int t = tid % LPR;
  const int g = tid / LPR;
  const int b = blockIdx.y;
  const int c0 = t * VEC;
  const bool cact = c0 < p.C;
  const float* Xc = p.X + (int64_t)b * p.x_bs + (cact ? c0 : 0);
  int n_sync = 0;
  for (int round = 0; round < s.rounds; ++round) {
    const int wg = round * s.nwg + (int)blockIdx.x;
    for (int i = tid; i < SLOTS * ROWF / 4; i += kSweepBlock) reinterpret_cast<float4*>(sweep_acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int e_begin = s.gptr[wg * G + g], e_end = s.gptr[wg * G + g + 1];
    float acc[VEC] = {0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    auto flush = [&]() {   // no-return LDS float adds on 4 consecutive words: nothing in the loop waits for them
      float* sl = sweep_acc + sweep_word<ROWF>(cur, c0);
#pragma unroll
      for (int i = 0; i < VEC; ++i) (void)__hip_atomic_fetch_add(sl + i, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    };
    // Every stream carries the ends of its first npan panels.  Panel 0 (the most referenced rows of X) and the tail behind the
    // listed panels are walked with run sums, the warm panels in between product by product.  In round 0 -- most of the entries
    // that can hit in L2 -- each listed panel ends with a rendezvous of the launch's workgroups, so that the workgroups of an
    // XCD share a panel while it is resident; the other rounds hold too few entries per panel to pay for one.
    const int npan = s.nbar;
    const int32_t* pp = s.pptr + (int64_t)(wg * G + g) * npan;
    const bool meet = round == 0 && gridDim.y == 1;
    int e0 = e_begin;
    for (int seg = 0; seg <= npan; ++seg) {
      const int e1 = seg < npan ? pp[seg] : e_end;
      if (seg == 0 || seg == npan) sweep_range<LPR, U, true, ROWF, NTM>(s, Xc, p.x_ld, sweep_acc, c0, t, e0, e1, cur, acc);
      else { flush(); sweep_range<LPR, U, false, ROWF, NTM>(s, Xc, p.x_ld, sweep_acc, c0, t, e0, e1, cur, acc); }
      e0 = e1;
      if (meet && seg < npan) sweep_rendezvous(s.sync, ++n_sync, (int)gridDim.x);
    }
    flush();
    __syncthreads();
    // write the rows: lane group g takes slots g, g + G, ...
#pragma unroll 1
    for (int j = 0; j < SPG; ++j) {
      const int sl = j * G + g;
      const int row = s.slot_row[(int64_t)wg * SLOTS + sl];
      if (row < 0) continue;                               // same for the lanes of a group
      float a[VEC];
      {
        const float4 o = *reinterpret_cast<const float4*>(sweep_acc + sweep_word<ROWF>(sl, c0));
        a[0] = o.x; a[1] = o.y; a[2] = o.z; a[3] = o.w;
      }
      if (cact) finish_row<VEC, NTM>(p, b, row, c0, a);
    }
    __syncthreads();
  }
}

template <int LPR>
inline void launch_sweep(hipStream_t st, const HopParams& p, const SweepParams& s, int nb) {
  constexpr int lds = (kSweepBlock / LPR) * kSweepSlotsPerGroup * LPR * 4 * (int)sizeof(float);
  if (s.nbar > 0) (void)hipMemsetAsync(s.sync, 0, 128, st);      // arrival counter + give-up flag of the rendezvous
  const int v = g_sweep_loads.load();        // developer A/B (tools/hop_bench.py): row loads in flight per lane
  if (v == 4) {
    allow_large_lds((const void*)hop_sweep_kernel<LPR, 0, 4>, lds);
    hipLaunchKernelGGL((hop_sweep_kernel<LPR, 0, 4>), dim3((unsigned)s.nwg, (unsigned)nb), dim3(kSweepBlock), lds, st, p, s);
  } else if (v == 16) {
    allow_large_lds((const void*)hop_sweep_kernel<LPR, 0, 16>, lds);
    hipLaunchKernelGGL((hop_sweep_kernel<LPR, 0, 16>), dim3((unsigned)s.nwg, (unsigned)nb), dim3(kSweepBlock), lds, st, p, s);
  } else {
    allow_large_lds((const void*)hop_sweep_kernel<LPR, 0, 8>, lds);
    hipLaunchKernelGGL((hop_sweep_kernel<LPR, 0, 8>), dim3((unsigned)s.nwg, (unsigned)nb), dim3(kSweepBlock), lds, st, p, s);
  }
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
  grid.x = (unsigned)(p.nblk + (p.nseg + GPB * R - 1) / (GPB * R));
  hipLaunchKernelGGL((hop_kernel<LPR, VEC, U, R, NTM>), grid, dim3(kBlock), 0, st, p);
}

// developer variants of the two float4 shapes that matter for the benchmarks (tools/hop_bench.py)
inline bool launch_hop_variant(int lpr, hipStream_t st, const HopParams& p, dim3 grid) {
  const int v = g_hop_variant.load();
  if (lpr == 16) {
    switch (v) {
      case 1: launch_hop<16, 4, 8, 1>(st, p, grid); return true;
      case 2: launch_hop<16, 4, 4, 2>(st, p, grid); return true;
      case 3: launch_hop<16, 4, 2, 1>(st, p, grid); return true;
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

template <int VEC>
int launch_hop_vec(hipStream_t st, const HopParams& p, int lpr, dim3 grid, dim3 fix_grid, const SweepParams* sw = nullptr) {
  if (sw && sw->rounds > 0) {
    if (VEC != 4) TGCN_FAIL(TGCN_ERR_INVALID, "hop: the sweep schedule needs 16-byte aligned rows");
    ProfScope ps(TGCN_PROF_HOP_SWEEP, st);
    switch (lpr) {
      case 4: launch_sweep<4>(st, p, *sw, (int)grid.y); break;
      case 8: launch_sweep<8>(st, p, *sw, (int)grid.y); break;
      case 16: launch_sweep<16>(st, p, *sw, (int)grid.y); break;
      case 32: launch_sweep<32>(st, p, *sw, (int)grid.y); break;
      case 64: launch_sweep<64>(st, p, *sw, (int)grid.y); break;
      default: TGCN_FAIL(TGCN_ERR_INVALID, "hop: sweep schedule with %d lanes per row", lpr);
    }
    TGCN_CHECK_LAUNCH("tgcn_csr_hop_f32 (sweep)");
  }
#define TGCN_HOP_CASE(L)                                                                    \
  case L: {                                                                                 \
    { ProfScope ps(TGCN_PROF_HOP, st);                                                      \
      if (!(VEC == 4 && g_hop_variant.load() != 0 && launch_hop_variant(L, st, p, grid))) { \
        /* interleave rows only when the grid still fills the chip afterwards */            \
        if (HopRows<L>::value > 1 && (int64_t)p.nblk * grid.y >= 4096)                       \
          launch_hop<L, VEC, 4, HopRows<L>::value>(st, p, grid);                            \
        else launch_hop<L, VEC, 4, 1>(st, p, grid);                                         \
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
