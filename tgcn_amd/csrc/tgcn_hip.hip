// tgcn_hip.hip -- gfx950 (MI355X) kernels + C ABI for the Chebyshev (time-)graph convolution.
// See include/tgcn_hip.h for the contract and DESIGN.md for layout / roofline notes.
//
// Kernels
//   hop_kernel<LPR,VEC,U,R>  row-block CSR x dense rows, fused  Y = alpha*(L X) + beta*Z  (+ P = L X)
//   hop_fixup_kernel<..>     folds long-row segment partials (fixed order => deterministic)
//   project_kernel<NT,VEC4>  stacked-hop projection on v_mfma_f32_16x16x4_f32 (exact fp32)
//   relayout_kernel          (Q,n,C) -> (n,Q,C)
//   pool_max_kernel / _bwd   gcn_pool / gcn_pool_4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "tgcn_hip.h"

namespace {

thread_local char g_err[512] = "";

#define TGCN_FAIL(code, ...)                    \
  do {                                          \
    snprintf(g_err, sizeof(g_err), __VA_ARGS__); \
    return (code);                              \
  } while (0)

#define TGCN_CHECK_LAUNCH(what)                                                         \
  do {                                                                                  \
    hipError_t e_ = hipGetLastError();                                                  \
    if (e_ != hipSuccess) TGCN_FAIL(TGCN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

constexpr int kBlock = 256;

// ---- optional launch timing (bench / tests): hipEvent pairs recorded around launches on their own stream
struct ProfRec { hipEvent_t a, b; int kind; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
std::atomic<int> g_prof_cap{0};

struct ProfScope {
  hipEvent_t b = nullptr;
  hipStream_t st;
  ProfScope(int kind, hipStream_t s) : st(s) {
    if (g_prof_cap.load(std::memory_order_relaxed) <= 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if ((int)g_prof.size() >= g_prof_cap.load()) return;
    ProfRec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    r.kind = kind;
    hipEventRecord(r.a, st);
    b = r.b;
    g_prof.push_back(r);
  }
  ~ProfScope() { if (b) hipEventRecord(b, st); }
};

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

// Sum of val_e * X[col_e, c0..c0+VEC) over stored entries [e0[rr], e1[rr]) of R rows (or segments) at once, by one
// group of LPR lanes.  Per row the group reads LPR entries with one coalesced 8-byte load per lane and hands them
// round with in-register shuffles; gathers are issued U at a time per row, so R*U 16-byte loads are in flight per
// lane.  R > 1 keeps R independent rowptr -> entry -> gather chains going, which is what low-degree rows on wide
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
    for (int j0 = 0; j0 < LPR; j0 += U) {
      if (j0 >= cmax) break;
      float xv[R][U][VEC];
      float vv[R][U];
#pragma unroll
      for (int rr = 0; rr < R; ++rr)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = __shfl(my_c[rr], j0 + u, LPR);
          vv[rr][u] = __shfl(my_v[rr], j0 + u, LPR);   // 0 past the end of the row
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

// Kernels that take more than 64 KB of dynamic LDS.  The attribute belongs to the calling thread's current device
// (nn.DataParallel drives several devices from one process), so it is set once per (device, kernel).
inline void allow_large_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert(std::make_pair(dev, fn)).second) hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

std::atomic<int> g_hop_variant{0};
std::atomic<int> g_proj_variant{0};   // 1: force the streaming-W kernel
std::atomic<int> g_overlap{0};
std::atomic<int> g_x3_form{2};          // bf16x3 projection, aligned operands, >= 96 output columns: 2 = A fragments from registers (+20 %), 1 = both operands through LDS
std::atomic<int> g_small_narrow{1};    // C <= 4 inputs of the one-launch path: input-side recursion (0: output-side kernel)
std::atomic<int> g_small_dense{1};     // small dense operands on the fp32 matrix pipe (0: vector-ALU kernels only)        // layer driver: projection of pass i on a side stream under the hops of pass i+1

struct SideStream { hipStream_t st = nullptr; hipEvent_t hops_done[2] = {nullptr, nullptr}; hipEvent_t proj_done[2] = {nullptr, nullptr}; };
std::mutex g_side_mu;
SideStream g_side[16];

// One helper stream + 4 events per device, created on first use and kept for the life of the process.
SideStream* side_stream() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStream& s = g_side[dev];
  if (!s.st) {
    if (hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) { s.st = nullptr; return nullptr; }
    for (int i = 0; i < 2; ++i) {
      hipEventCreateWithFlags(&s.hops_done[i], hipEventDisableTiming);
      hipEventCreateWithFlags(&s.proj_done[i], hipEventDisableTiming);
    }
  }
  return &s;
}

template <int LPR, int VEC, int U, int R>
inline void launch_hop(hipStream_t st, const HopParams& p, dim3 grid) {
  constexpr int GPB = kBlock / LPR;
  grid.x = (unsigned)(p.nblk + (p.nseg + GPB * R - 1) / (GPB * R));
  hipLaunchKernelGGL((hop_kernel<LPR, VEC, U, R, 0>), grid, dim3(kBlock), 0, st, p);
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
int launch_hop_vec(hipStream_t st, const HopParams& p, int lpr, dim3 grid, dim3 fix_grid) {
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
};

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
        const float4 v = *reinterpret_cast<const float4*>(A + rr * lda + kc);
        ra[h * 4 + 0] = ok ? v.x : 0.f; ra[h * 4 + 1] = ok ? v.y : 0.f; ra[h * 4 + 2] = ok ? v.z : 0.f; ra[h * 4 + 3] = ok ? v.w : 0.f;
      }
    } else {
#pragma unroll
      for (int h = 0; h < 16; ++h) {
        const int row = (tid >> 5) + h * 8, kk = tid & 31;
        const bool ok = (m0 + row < p.M) && (k0 + kk < p.Kc);
        const int64_t rr = (m0 + row < p.M) ? m0 + row : p.M - 1;
        const int kc = (k0 + kk < p.Kc) ? k0 + kk : 0;
        const float v = A[proj_row_off(p, rr, lda) + kc];
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
      const int64_t orow = (p.interleave == 1) ? m : (m % p.interleave) * p.n_vertices + m / p.interleave;
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
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * NW;
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
    const float* __restrict__ A = p.a[term];
    const int64_t lda = p.lda[term];
    const float* __restrict__ Wt = p.W + (int64_t)term * p.Kc * p.N;
    if constexpr (VEC4) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = (tid >> 3) + h * 64, kk = (tid & 7) * 4;
        const bool ok = (m0 + row < p.M) && (k0 + kk < p.Kc);
        const int64_t rr = (m0 + row < p.M) ? m0 + row : p.M - 1;
        const int kc = (k0 + kk < p.Kc) ? k0 + kk : 0;
        const float4 v = *reinterpret_cast<const float4*>(A + rr * lda + kc);
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
          const float v = A[proj_row_off(p, rr, lda) + (ok ? k0 + kk + j : 0)];
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
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int idx = lane + 64 * it, row = idx / SEGS, seg = (idx % SEGS) * 4;
          const int64_t m = m0 + wave * 32 + r * 16 + row;
          const int col = n0 + seg;
          if (m >= p.M || col >= p.N) continue;
          float4 v = *reinterpret_cast<const float4*>(&my[row * ES + seg]);
          const int64_t orow = (p.interleave == 1) ? m : (m % p.interleave) * p.n_vertices + m / p.interleave;
          if (p.bias_kind && col < p.bias_cols) {
            const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + (p.bias_kind == 2 ? vert * p.bias_ld : 0) + col);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          }
          float4* o = reinterpret_cast<float4*>(p.out + orow * p.ldo + col);
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
      const int64_t orow = (p.interleave == 1) ? m : (m % p.interleave) * p.n_vertices + m / p.interleave;
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

// ---- bf16x3, second form (16-byte aligned operands): a wave's A rows are used by that wave only, so its A fragments
// go global -> registers -> split -> MFMA operand with no LDS round trip and no barrier; only the W tile (shared by the
// 8 waves) is split into LDS, double-buffered, ONE barrier per 32-k tile.  A lane loads the 8 consecutive k of its row
// as two float4 (the four k groups of a row are adjacent: whole 128-byte lines per row).
template <int NT>
__global__ __launch_bounds__(512) void project_x3v2_kernel(const ProjParams p) {
  constexpr int XT = 512, BM = 256, KT = 32, RS = KT;
  constexpr int NW = NT * 16;
  constexpr int WPAIRS = (KT / 2 * NW + XT - 1) / XT;
  constexpr int WBUF = 3 * NW * RS;                                   // bf16 elements of one W buffer (3 planes)
  constexpr int ES = NW + 4;                                          // epilogue scratch row stride (floats)
  constexpr int LDS_BYTES = (2 * WBUF * 2 > 8 * 16 * ES * 4) ? 2 * WBUF * 2 : 8 * 16 * ES * 4;
  __shared__ __align__(16) unsigned char lds_raw[LDS_BYTES];
  unsigned short* Wp = reinterpret_cast<unsigned short*>(lds_raw);    // [2][3][NW * RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * NW;
  f32x4 acc[2][NT];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ktiles = (p.Kc + KT - 1) / KT;
  const int total = p.nterms * ktiles;
  // per-lane, tile-invariant parts of every address (the loop below adds only wave-uniform tile offsets: the vector ALU
  // is the co-bottleneck of this kernel -- an MFMA holds vector issue for 8 of its 16 cycles)
  int64_t arow[2];                                   // element offset of this lane's 8 k inside row r (without lda * row: see aoff)
  int64_t rowc[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int64_t m = m0 + wave * 32 + r * 16 + r16;
    rowc[r] = m < p.M ? m : p.M - 1;                 // rows past the end re-read the last row; their results are never stored
    arow[r] = 0;
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
  float ra[2][8], rw[2 * WPAIRS];
  auto load_a = [&](int ti, float (&dst)[2][8]) {
    const int term = ti / ktiles, k0 = (ti % ktiles) * KT;           // wave-uniform
    const float* __restrict__ A = p.a[term] + k0 + kg * 8;
    const int64_t lda = p.lda[term];
    if (k0 + KT <= p.Kc) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 v = *reinterpret_cast<const float4*>(A + rowc[r] * lda + h * 4);
          dst[r][h * 4 + 0] = v.x; dst[r][h * 4 + 1] = v.y; dst[r][h * 4 + 2] = v.z; dst[r][h * 4 + 3] = v.w;
        }
    } else {                                                         // last k tile of a term: k past Kc reads as zero
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = k0 + kg * 8 + h * 4 < p.Kc;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok) v = *reinterpret_cast<const float4*>(A + rowc[r] * lda + h * 4);
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
  load_a(0, ra);
  load_w(0);
  store_w(0);
  __syncthreads();
  const int frag = r16 * RS + x3_chunk(r16, kg) * 8;
  for (int ti = 0; ti < total; ++ti) {
    float rn[2][8];
    const bool more = ti + 1 < total;
    if (more) { load_a(ti + 1, rn); load_w(ti + 1); }
    bf16x8 a[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      unsigned pl[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split3(ra[r][2 * j], ra[r][2 * j + 1], pl[0][j], pl[1][j], pl[2][j]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        a[r][q] = __builtin_bit_cast(bf16x8, u32x4{pl[q][0], pl[q][1], pl[q][2], pl[q][3]});
      }
    }
    const unsigned short* W0 = Wp + (ti & 1) * WBUF;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bf16x8 w[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) w[q] = *reinterpret_cast<const bf16x8*>(&W0[q * NW * RS + (nt * 16) * RS + frag]);
      // smallest terms first; the two row tiles alternate so that consecutive MFMAs are independent
      f32x4 c0 = acc[0][nt], c1 = acc[1][nt];
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][2], w[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][2], w[0], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][1], w[1], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][1], w[1], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], w[2], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], w[2], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][1], w[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][1], w[0], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], w[1], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], w[1], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], w[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], w[0], c1, 0, 0, 0);
      acc[0][nt] = c0; acc[1][nt] = c1;
    }
    if (more) {
      store_w((ti + 1) & 1);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) ra[r][j] = rn[r][j];
    }
    __syncthreads();
  }
  // ---- epilogue
  if constexpr (NT <= 4) {
    if (p.vec_epilogue) {
      constexpr int SEGS = NW / 4, ITER = (16 * SEGS) / 64;
      float* my = reinterpret_cast<float*>(lds_raw) + wave * (16 * ES);     // the loop ended with a barrier: W buffers are free
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int i = 0; i < 4; ++i) my[(kg * 4 + i) * ES + nt * 16 + r16] = acc[r][nt][i];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int idx = lane + 64 * it, row = idx / SEGS, seg = (idx % SEGS) * 4;
          const int64_t m = m0 + wave * 32 + r * 16 + row;
          const int col = n0 + seg;
          if (m >= p.M || col >= p.N) continue;
          float4 v = *reinterpret_cast<const float4*>(&my[row * ES + seg]);
          const int64_t orow = (p.interleave == 1) ? m : (m % p.interleave) * p.n_vertices + m / p.interleave;
          if (p.bias_kind && col < p.bias_cols) {
            const int64_t vert = orow < p.n_vertices ? orow : orow % p.n_vertices;
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + (p.bias_kind == 2 ? vert * p.bias_ld : 0) + col);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          }
          float4* o = reinterpret_cast<float4*>(p.out + orow * p.ldo + col);
          if (p.accumulate) { const float4 ov = *o; v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w; }
          *o = v;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + wave * 32 + r * 16 + kg * 4 + i;
      if (m >= p.M) continue;
      const int64_t orow = (p.interleave == 1) ? m : (m % p.interleave) * p.n_vertices + m / p.interleave;
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
          const float4 v = *reinterpret_cast<const float4*>(A + rr * lda + kc);
          ra[i][0] = ok ? v.x : 0.f; ra[i][1] = ok ? v.y : 0.f; ra[i][2] = ok ? v.z : 0.f; ra[i][3] = ok ? v.w : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < kResRows; ++i) {
          const bool ok = (m0 + i < p.M) && (k0 + lane < p.Kc);
          const int64_t rr = (m0 + i < p.M) ? m0 + i : p.M - 1;
          const int kc = (k0 + lane < p.Kc) ? k0 + lane : 0;
          const float v = A[proj_row_off(p, rr, lda) + kc];
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
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int idx = lane + 64 * it, row = idx / SEGS, seg = (idx % SEGS) * 4;
        const int64_t m = m0 + row;
        const int col = n0 + seg;
        if (m >= p.M || col >= p.N) continue;
        const float2 lo = *reinterpret_cast<const float2*>(&my[row * kResAS + seg]);
        const float2 hi = *reinterpret_cast<const float2*>(&my[row * kResAS + seg + 2]);
        float4 v = make_float4(lo.x, lo.y, hi.x, hi.y);
        const int64_t orow = (p.interleave == 1) ? m : (m % p.interleave) * p.n_vertices + m / p.interleave;
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
          const int64_t orow = (p.interleave == 1) ? m : (m % p.interleave) * p.n_vertices + m / p.interleave;
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
    const float* __restrict__ at = p.a[t] + mb0 * p.lda[t];
    const int64_t ld = p.lda[t];
    for (int i = threadIdx.x; i < nrows * p.Kc; i += kBlock) {
      const int l = i / p.Kc, kc = i - l * p.Kc;                 // local row = (it * 4 + j) * RP + r
      const int it = l / (4 * RP), rem = l - it * 4 * RP, j = rem / RP, r = rem - j * RP;
      sA[(((t * p.Kc + kc) * iters + it) * RP + r) * 4 + j] = at[(int64_t)l * ld + kc];
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
      const int64_t orow = (p.interleave == 1) ? m : (m % p.interleave) * p.n_vertices + m / p.interleave;
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


constexpr int kWgTerms = 5;   // terms accumulated at once per wave (register budget: 5 * 4 tiles * 4 regs)

// One wave per (row block, 64-column tile of G, 16-row tile of the weight): dW_t tile = A_t^T G over the block's rows
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
  const int ct = blockIdx.z;
  float* part = p.partial + (size_t)blockIdx.x * p.nterms * p.Kc * p.N;
  const int c = ct * 16 + r;
  for (int t0 = 0; t0 < p.nterms; t0 += kWgTerms) {
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
constexpr int kDenseMaxC = 32;       // input row length (X fragments in registers)
constexpr int kDenseWFloats = 4096;  // LDS for weight tiles: all K of them when they fit (staged once), else one per step
template <int S, bool BASIS, int NW>   // NW: most waves (16-row tiles) of a workgroup -> register budget and size of Lf
__global__ __launch_bounds__(NW * 64) void small_dense_kernel(const SmallParams p) {
  extern __shared__ __align__(16) float smem[];
  constexpr int LDY = S * 16 + 16;
  constexpr int NKMAX = NW * 4, XKMAX = kDenseMaxC / 4;
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

// samples per workgroup (4 / 2 / 1) of small_dense_kernel, 0 when the shape is not for it
inline int dense_mfma_config(int64_t n, int64_t nnz, int32_t C, int32_t mode, int64_t q, int64_t col_tiles, bool basis) {
  if (n < 16 || n > kDenseMaxN || (!basis && C > kDenseMaxC) || (mode != 0 && mode != 1)) return 0;
  if (nnz * 4 < n * n) return 0;                       // at least a quarter of the entries stored: dense arithmetic pays
  const int npad = (int)(n + 15) / 16 * 16;
  const int nbuf = mode == 0 ? 2 : 3;
  const int nw = npad / 16;            // register budget per lane shrinks with the wave count: fewer samples (accumulators)
  const int smax = basis ? (nw > 12 ? 2 : 4) : (nw > 12 ? 1 : (nw > 8 ? 2 : 4));
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
#define TGCN_DENSE(SV, NWV)                                                                  \
  {                                                                                          \
    allow_large_lds((const void*)small_dense_kernel<SV, BASIS, NWV>, 160 * 1024);            \
    hipLaunchKernelGGL((small_dense_kernel<SV, BASIS, NWV>), grid, block, lds, st, p);       \
  }
#define TGCN_DENSE_S(NWV) \
  if (S == 4) TGCN_DENSE(4, NWV) else if (S == 2) TGCN_DENSE(2, NWV) else TGCN_DENSE(1, NWV)
  const int nw = npad / 16;
  if (nw <= 8) { TGCN_DENSE_S(8) } else if (nw <= 12) { TGCN_DENSE_S(12) } else { TGCN_DENSE_S(16) }
#undef TGCN_DENSE_S
#undef TGCN_DENSE
}

// --------------------------------------------------------------------------------------------------
// relayout (Q,n,C) -> (n,Q,C), C <= 32
// --------------------------------------------------------------------------------------------------
constexpr int kRelT = 16;
__global__ __launch_bounds__(kBlock) void relayout_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int64_t Q, int64_t n, int C) {
  __shared__ float tile[kRelT * kRelT * 32];
  const int64_t i0 = (int64_t)blockIdx.x * kRelT, q0 = (int64_t)blockIdx.y * kRelT;
  const int seg = kRelT * C;  // floats per (q, 16 vertices) or per (vertex, 16 q)
  for (int e = threadIdx.x; e < kRelT * seg; e += kBlock) {
    const int q = e / seg, rem = e % seg;
    float v = 0.f;
    if (q0 + q < Q && i0 + rem / C < n) v = in[((q0 + q) * n + i0) * C + rem];
    tile[e] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kRelT * seg; e += kBlock) {
    const int i = e / seg, rem = e % seg;
    const int q = rem / C, c = rem % C;
    if (i0 + i < n && q0 + q < Q) out[((i0 + i) * Q + q0) * C + rem] = tile[(q * kRelT + i) * C + c];
  }
}

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

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// ==================================================================================================
// C ABI
// ==================================================================================================
extern "C" {

const char* tgcn_last_error(void) { return g_err; }
int tgcn_abi_version(void) { return TGCN_ABI_VERSION; }

int tgcn_set_tuning(const char* key, int32_t value) {
  if (key && strcmp(key, "hop_variant") == 0) { g_hop_variant.store(value); return TGCN_OK; }
  if (key && strcmp(key, "project_variant") == 0) { g_proj_variant.store(value); return TGCN_OK; }
  if (key && strcmp(key, "small_dense") == 0) { g_small_dense.store(value); return TGCN_OK; }
  if (key && strcmp(key, "small_narrow") == 0) { g_small_narrow.store(value); return TGCN_OK; }
  if (key && strcmp(key, "x3_form") == 0) { g_x3_form.store(value); return TGCN_OK; }
  if (key && strcmp(key, "overlap") == 0) { g_overlap.store(value); return TGCN_OK; }
  TGCN_FAIL(TGCN_ERR_INVALID, "set_tuning: unknown key");
}

int tgcn_profile_start(int32_t capacity) {
  if (capacity <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "profile: capacity %d", capacity);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  g_prof.clear();
  g_prof.reserve(capacity);
  g_prof_cap.store(capacity);
  return TGCN_OK;
}

int tgcn_profile_stop(int32_t* kinds, float* ms, int32_t capacity, int32_t* count) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_cap.store(0);
  int n = 0;
  for (auto& r : g_prof) {
    float t = 0.f;
    hipEventSynchronize(r.b);
    hipEventElapsedTime(&t, r.a, r.b);
    if (n < capacity && kinds && ms) { kinds[n] = r.kind; ms[n] = t; ++n; }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  g_prof.clear();
  if (count) *count = n;
  return TGCN_OK;
}

int tgcn_hop_vec_width(int32_t C, int aligned16) { return C > 0 ? hop_geom(C, aligned16).vec : 0; }
int tgcn_hop_lanes_per_row(int32_t C, int aligned16) { return C > 0 ? hop_geom(C, aligned16).lpr : 0; }
int tgcn_hop_groups_per_block(int32_t C, int aligned16) { return C > 0 ? kBlock / hop_geom(C, aligned16).lpr : 0; }

size_t tgcn_csr_hop_workspace_bytes(const tgcn_csr_sched* sched, int32_t nb, int32_t C, int aligned16) {
  if (!sched || C <= 0 || nb <= 0) return 0;
  return (size_t)sched->npartial * (size_t)nb * (size_t)hop_geom(C, aligned16).cpad * sizeof(float);
}

int tgcn_csr_hop_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t nb, int32_t C,
                     const tgcn_dense* X, const tgcn_dense* Z, float alpha, float beta, const tgcn_dense* Y,
                     const tgcn_dense* P, void* workspace, size_t workspace_bytes) {
  return tgcn_csr_hop2_f32(stream, A, S, nb, C, X, Z, alpha, beta, nullptr, 0.f, Y, P, workspace, workspace_bytes);
}

int tgcn_csr_hop2_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t nb, int32_t C,
                      const tgcn_dense* X, const tgcn_dense* Z, float alpha, float beta, const tgcn_dense* Z2, float gamma,
                      const tgcn_dense* Y, const tgcn_dense* P, void* workspace, size_t workspace_bytes) {
  if (!A || !S || !X || !X->ptr) TGCN_FAIL(TGCN_ERR_INVALID, "hop: null operand");
  if ((!Y || !Y->ptr) && (!P || !P->ptr)) TGCN_FAIL(TGCN_ERR_INVALID, "hop: no output");
  if (A->n <= 0 || A->nnz < 0 || A->nnz >= (int64_t)INT32_MAX || A->n >= (int64_t)INT32_MAX)
    TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "hop: n=%lld nnz=%lld outside int32 index range", (long long)A->n, (long long)A->nnz);
  if (nb <= 0 || C <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "hop: nb=%d C=%d", nb, C);
  if (!A->rowptr || (A->nnz > 0 && !A->edges) || !S->blk_row) TGCN_FAIL(TGCN_ERR_INVALID, "hop: null CSR/schedule array");
  const int al = aligned4(X) && aligned4(Z) && aligned4(Z2) && aligned4(Y) && aligned4(P);
  const HopGeom g = hop_geom(C, al);
  if (S->lanes_per_row != g.lpr)
    TGCN_FAIL(TGCN_ERR_INVALID, "hop: schedule built for %d lanes/row, C=%d (aligned16=%d) needs %d", S->lanes_per_row, C, al, g.lpr);
  if (S->nblk <= 0 || S->row_thresh <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "hop: empty schedule");
  if (S->nseg < 0 || S->nlong < 0 || S->nhuge < 0 || S->nhuge > S->nlong || S->npartial < 0) TGCN_FAIL(TGCN_ERR_INVALID, "hop: bad schedule counts");
  if ((int64_t)nb * g.nchunks > 65535) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "hop: nb*chunks=%lld > 65535", (long long)nb * g.nchunks);
  if (S->nseg > 0 && (!S->seg_row || !S->seg_e0 || !S->seg_e1 || !S->seg_slot)) TGCN_FAIL(TGCN_ERR_INVALID, "hop: null segment arrays");
  if (S->npartial > 0) {
    const size_t need = (size_t)S->npartial * nb * g.cpad * sizeof(float);
    if (!workspace || workspace_bytes < need) TGCN_FAIL(TGCN_ERR_WORKSPACE, "hop: workspace %zu < %zu", workspace_bytes, need);
    if (!S->long_row || !S->long_slot || S->nlong <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "hop: null long-row arrays");
  }
  HopParams p;
  memset(&p, 0, sizeof(p));
  p.rowptr = A->rowptr; p.ev = A->edges; p.blk_row = S->blk_row;
  p.seg_row = S->seg_row; p.seg_e0 = S->seg_e0; p.seg_e1 = S->seg_e1; p.seg_slot = S->seg_slot;
  p.long_row = S->long_row; p.long_slot = S->long_slot;
  p.X = X->ptr; p.x_bs = X->batch_stride; p.x_ld = X->row_stride;
  if (Z && Z->ptr) { p.Z = Z->ptr; p.z_bs = Z->batch_stride; p.z_ld = Z->row_stride; }
  if (Z2 && Z2->ptr) { p.Z2 = Z2->ptr; p.z2_bs = Z2->batch_stride; p.z2_ld = Z2->row_stride; p.gamma = gamma; }
  if (Y && Y->ptr) { p.Y = Y->ptr; p.y_bs = Y->batch_stride; p.y_ld = Y->row_stride; }
  if (P && P->ptr) { p.P = P->ptr; p.p_bs = P->batch_stride; p.p_ld = P->row_stride; }
  p.partial = (float*)workspace;
  p.alpha = alpha; p.beta = beta;
  p.nblk = S->nblk; p.nseg = S->nseg; p.nlong = S->nlong; p.nhuge = S->nhuge; p.row_thresh = S->row_thresh;
  p.C = C; p.nb = nb; p.nchunks = g.nchunks; p.cpad = g.cpad;
  const int gpb = kBlock / g.lpr;
  const int seg_blocks = (S->nseg + gpb - 1) / gpb;
  const dim3 grid((unsigned)(S->nblk + seg_blocks), (unsigned)(nb * g.nchunks));
  const int fix_blocks = S->nhuge + (S->nlong - S->nhuge + gpb - 1) / gpb;
  const dim3 fix_grid((unsigned)(fix_blocks > 0 ? fix_blocks : 1), (unsigned)(nb * g.nchunks));
  hipStream_t st = (hipStream_t)stream;
  return g.vec == 4 ? launch_hop_vec<4>(st, p, g.lpr, grid, fix_grid) : launch_hop_vec<1>(st, p, g.lpr, grid, fix_grid);
}

static int project_impl(void* stream, int64_t M, int32_t Kc, int32_t N, int32_t nterms, const float* const* a,
                        const int64_t* lda, const float* W, const float* bias, int32_t bias_kind,
                        int64_t n_vertices, int64_t interleave, int32_t accumulate, float* out, int64_t ldo,
                        int32_t win_n, int32_t win_t, int32_t bias_cols = -1);

int tgcn_cheb_project_f32(void* stream, int64_t M, int32_t Kc, int32_t N, int32_t nterms, const float* const* a,
                          const int64_t* lda, const float* W, const float* bias, int32_t bias_kind,
                          int64_t n_vertices, int64_t interleave, int32_t accumulate, float* out, int64_t ldo) {
  return project_impl(stream, M, Kc, N, nterms, a, lda, W, bias, bias_kind, n_vertices, interleave, accumulate, out, ldo, 0, 0);
}

int tgcn_cheb_project_windows_f32(void* stream, int64_t n_vertices, int32_t T, int32_t H, int32_t N, int32_t nterms,
                                  const float* const* series, const float* W, const float* bias, int32_t bias_kind,
                                  float* out) {
  if (T < H || H < 1) TGCN_FAIL(TGCN_ERR_INVALID, "project_windows: need 1 <= H <= T");
  const int32_t nwin = T - H + 1;
  int64_t lda[kMaxTerms];
  for (int t = 0; t < kMaxTerms; ++t) lda[t] = T;
  return project_impl(stream, n_vertices * nwin, H, N, nterms, series, lda, W, bias, bias_kind, n_vertices, nwin, 0, out, N, nwin, T);
}

static int project_impl(void* stream, int64_t M, int32_t Kc, int32_t N, int32_t nterms, const float* const* a,
                        const int64_t* lda, const float* W, const float* bias, int32_t bias_kind,
                        int64_t n_vertices, int64_t interleave, int32_t accumulate, float* out, int64_t ldo,
                        int32_t win_n, int32_t win_t, int32_t bias_cols) {
  if (M <= 0 || Kc <= 0 || N <= 0 || nterms <= 0 || !a || !lda || !W || !out) TGCN_FAIL(TGCN_ERR_INVALID, "project: bad argument");
  if (nterms > kMaxTerms) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "project: nterms %d > %d (chunk with accumulate=1)", nterms, kMaxTerms);
  if (bias_kind < 0 || bias_kind > 2 || (bias_kind && !bias)) TGCN_FAIL(TGCN_ERR_INVALID, "project: bias_kind %d", bias_kind);
  if (interleave < 1 || n_vertices < 1) TGCN_FAIL(TGCN_ERR_INVALID, "project: interleave/n_vertices");
  if (interleave > 1 && M != interleave * n_vertices) TGCN_FAIL(TGCN_ERR_INVALID, "project: M != interleave*n_vertices");
  ProjParams p;
  memset(&p, 0, sizeof(p));
  bool vec4 = (Kc % 4 == 0) && win_n == 0;   // windows start at any float: scalar loads
  p.win_n = win_n; p.win_t = win_t;
  p.bias_cols = bias_cols < 0 ? N : bias_cols;   // bias rows have bias_cols floats
  p.bias_ld = p.bias_cols;
  for (int t = 0; t < nterms; ++t) {
    if (!a[t]) TGCN_FAIL(TGCN_ERR_INVALID, "project: null term %d", t);
    p.a[t] = a[t];
    p.lda[t] = lda[t];
    vec4 = vec4 && (((uintptr_t)a[t] & 15) == 0) && (lda[t] % 4 == 0);
  }
  p.W = W; p.bias = bias; p.out = out; p.M = M; p.ldo = ldo; p.n_vertices = n_vertices; p.interleave = interleave;
  p.Kc = Kc; p.N = N; p.nterms = nterms; p.bias_kind = bias_kind; p.accumulate = accumulate;
  p.vec_epilogue = (N % 4 == 0) && (ldo % 4 == 0) && (((uintptr_t)out & 15) == 0) && (!bias || ((uintptr_t)bias & 15) == 0) &&
                   (p.bias_cols % 4 == 0);
  const int nt = N <= 16 ? 1 : (N <= 32 ? 2 : 4);
  hipStream_t st = (hipStream_t)stream;
  const int kc4 = (Kc + 3) / 4 * 4;
  const size_t wbytes = (size_t)nterms * kc4 * nt * 16 * sizeof(float);
  const unsigned gy = (unsigned)((N + nt * 16 - 1) / (nt * 16));
  // project_variant: 0 auto (bf16x3 on large problems, else exact fp32: W-resident when it fits, streaming otherwise),
  // 1 exact-fp32 streaming, 2 exact-fp32 (W-resident with 16-row wave tiles when it fits), 3 bf16x3 always, 4 exact fp32 auto,
  // 5 vector-ALU kernel whenever it applies (auto uses it for sum(Kc) <= 16 and M >= 4096)
  const int pv = g_proj_variant.load();
  if ((pv == 0 || pv == 5) && (int64_t)Kc * nterms <= kNarrowMaxK && win_n == 0 && p.vec_epilogue && N <= 1024 &&
      (M >= 4096 || pv == 5)) {
    // a few scalars per row: stream the output from the vector ALU (project_narrow_kernel)
    const int L = N / 4, RP = kBlock / L;
    const int ktot = Kc * nterms;
    int iters = (40 * 1024 / 4 - ktot * N) / (ktot * RP * 4);    // A values of a block: about 40 KB of LDS with the weights
    iters = iters < 1 ? 1 : (iters > 16 ? 16 : iters);
    const int rows_per_block = RP * 4 * iters;
    const int64_t nb = (M + rows_per_block - 1) / rows_per_block;
    if (nb > (int64_t)INT32_MAX) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "project: M too large");
    const size_t lds = ((size_t)ktot * N + (size_t)ktot * rows_per_block) * sizeof(float);
    allow_large_lds((const void*)project_narrow_kernel, 160 * 1024);
    ProfScope ps(TGCN_PROF_PROJECT, st);
    hipLaunchKernelGGL(project_narrow_kernel, dim3((unsigned)nb), dim3(kBlock), lds, st, p, iters);
    TGCN_CHECK_LAUNCH("tgcn_cheb_project_f32 (narrow)");
    return TGCN_OK;
  }
  const bool use_x3 = pv == 3 || (pv == 0 && M >= 8192 && (int64_t)Kc * nterms >= 64);
  if (wbytes <= (size_t)kResMaxWBytes && pv != 1 && !use_x3) {
    const int rt = g_proj_variant.load() == 2 ? 1 : 2;                  // 8 waves x 32 rows (variant 2: 16 waves x 16 rows)
    const int res_rows = 16 * rt, res_waves = 1024 / rt / 64;
    const size_t lds = wbytes + (size_t)kResScratchFloats * sizeof(float);
    const int64_t ntiles = (M + res_rows - 1) / res_rows;
    int64_t gx = (ntiles + res_waves - 1) / res_waves;
    if (gx > 256) gx = 256;    // one persistent workgroup per CU (LDS-limited residency)
    const dim3 grid((unsigned)gx, gy);
    ProfScope ps(TGCN_PROF_PROJECT, st);
#define TGCN_PROJ_R(NTV, V4)                                                                                  \
  {                                                                                                           \
    allow_large_lds((const void*)project_resident_kernel<NTV, V4, 1>, kResMaxWBytes + kResScratchFloats * (int)sizeof(float)); \
    allow_large_lds((const void*)project_resident_kernel<NTV, V4, 2>, kResMaxWBytes + kResScratchFloats * (int)sizeof(float)); \
    if (rt == 1) hipLaunchKernelGGL((project_resident_kernel<NTV, V4, 1>), grid, dim3(1024), lds, st, p, kc4, ntiles); \
    else hipLaunchKernelGGL((project_resident_kernel<NTV, V4, 2>), grid, dim3(512), lds, st, p, kc4, ntiles);          \
  }
    if (nt == 1) { if (vec4) TGCN_PROJ_R(1, true) else TGCN_PROJ_R(1, false) }
    else if (nt == 2) { if (vec4) TGCN_PROJ_R(2, true) else TGCN_PROJ_R(2, false) }
    else { if (vec4) TGCN_PROJ_R(4, true) else TGCN_PROJ_R(4, false) }
#undef TGCN_PROJ_R
    TGCN_CHECK_LAUNCH("tgcn_cheb_project_f32 (resident)");
    return TGCN_OK;
  }
  const int64_t mb = (M + 127) / 128;
  if (mb > (int64_t)INT32_MAX) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "project: M too large");
  // streaming-W kernel: widest column tile that keeps padding low, so A is read once per block and no MFMA works
  // on padding (N = 160 -> one block of 10 tiles instead of three of 4)
  const int tiles = (N + 15) / 16;
  int nts = tiles <= 1 ? 1 : tiles <= 2 ? 2 : tiles <= 4 ? 4 : tiles <= 6 ? 6 : tiles <= 8 ? 8 : 10;
  if (tiles > 10) {   // several column blocks: the width with the least padded tiles
    int best = 10, waste = (10 - tiles % 10) % 10;
    for (int c : {8, 6, 4}) { const int w = (c - tiles % c) % c; if (w < waste) { waste = w; best = c; } }
    nts = best;
  }
  const dim3 grid((unsigned)mb, (unsigned)((tiles + nts - 1) / nts));
  ProfScope ps(TGCN_PROF_PROJECT, st);
#define TGCN_PROJ(NTV)                                                                               \
  if (vec4) hipLaunchKernelGGL((project_kernel<NTV, true>), grid, dim3(kBlock), 0, st, p);             \
  else hipLaunchKernelGGL((project_kernel<NTV, false>), grid, dim3(kBlock), 0, st, p);
  if (use_x3) {      // bf16x3 products on the bf16 matrix pipe
    const dim3 grid3((unsigned)((M + 255) / 256), grid.y);
#define TGCN_PROJ3(NTV)                                                                              \
  if (vec4 && NTV >= 6 && g_x3_form.load() == 2) hipLaunchKernelGGL((project_x3v2_kernel<NTV>), grid3, dim3(512), 0, st, p); /* wide outputs: compute-bound */ \
  else if (vec4) hipLaunchKernelGGL((project_x3_kernel<NTV, true>), grid3, dim3(512), 0, st, p);      \
  else hipLaunchKernelGGL((project_x3_kernel<NTV, false>), grid3, dim3(512), 0, st, p);
    switch (nts) {
      case 1: TGCN_PROJ3(1) break;
      case 2: TGCN_PROJ3(2) break;
      case 4: TGCN_PROJ3(4) break;
      case 6: TGCN_PROJ3(6) break;
      case 8: TGCN_PROJ3(8) break;
      default: TGCN_PROJ3(10) break;
    }
#undef TGCN_PROJ3
    TGCN_CHECK_LAUNCH("tgcn_cheb_project_f32 (bf16x3)");
    return TGCN_OK;
  }
  switch (nts) {
    case 1: TGCN_PROJ(1) break;
    case 2: TGCN_PROJ(2) break;
    case 4: TGCN_PROJ(4) break;
    case 6: TGCN_PROJ(6) break;
    case 8: TGCN_PROJ(8) break;
    default: TGCN_PROJ(10) break;
  }
#undef TGCN_PROJ
  TGCN_CHECK_LAUNCH("tgcn_cheb_project_f32");
  return TGCN_OK;
}

static int64_t wgrad_rows_per_block(int64_t M) {
  int64_t rpb = (M + 1023) / 1024;            // at most 1024 row blocks (partials to fold) ...
  if (rpb < 64) rpb = 64;                     // ... of at least 64 rows
  return (rpb + 15) / 16 * 16;
}
static int wgrad_blocks(int64_t M) {
  const int64_t rpb = wgrad_rows_per_block(M);
  return (int)((M + rpb - 1) / rpb);
}

size_t tgcn_cheb_wgrad_workspace_bytes(int64_t M, int32_t Kc, int32_t N, int32_t nterms) {
  if (M <= 0 || Kc <= 0 || N <= 0 || nterms <= 0) return 0;
  return (size_t)wgrad_blocks(M) * nterms * Kc * N * sizeof(float);
}

int tgcn_cheb_wgrad_f32(void* stream, int64_t M, int32_t Kc, int32_t N, int32_t nterms, const float* const* a,
                        const int64_t* lda, const float* G, int64_t ldg, float* dW, void* workspace, size_t workspace_bytes) {
  if (M <= 0 || Kc <= 0 || N <= 0 || nterms <= 0 || !a || !lda || !G || !dW) TGCN_FAIL(TGCN_ERR_INVALID, "wgrad: bad argument");
  if (nterms > kMaxTerms) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "wgrad: nterms %d > %d", nterms, kMaxTerms);
  const size_t need = tgcn_cheb_wgrad_workspace_bytes(M, Kc, N, nterms);
  if (!workspace || workspace_bytes < need) TGCN_FAIL(TGCN_ERR_WORKSPACE, "wgrad: workspace %zu < %zu", workspace_bytes, need);
  WgradParams p;
  memset(&p, 0, sizeof(p));
  for (int t = 0; t < nterms; ++t) {
    if (!a[t]) TGCN_FAIL(TGCN_ERR_INVALID, "wgrad: null term %d", t);
    p.a[t] = a[t];
    p.lda[t] = lda[t];
  }
  p.G = G; p.partial = (float*)workspace; p.dW = dW; p.M = M; p.ldg = ldg;
  p.Kc = Kc; p.N = N; p.nterms = nterms; p.nblocks = wgrad_blocks(M);
  p.rows_per_block = wgrad_rows_per_block(M);
  const int ctiles = (Kc + 15) / 16;
  if ((N + 63) / 64 > 65535 || ctiles > 65535) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "wgrad: Kc=%d N=%d too wide", Kc, N);
  hipStream_t st = (hipStream_t)stream;
  { ProfScope ps(TGCN_PROF_WGRAD, st);
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3(p.nblocks, (N + 63) / 64, ctiles), dim3(64), 0, st, p); }
  { ProfScope ps(TGCN_PROF_WGRAD, st);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((int64_t)nterms * Kc * N + 63) / 64)), dim3(1024), 0, st, p); }
  TGCN_CHECK_LAUNCH("tgcn_cheb_wgrad_f32");
  return TGCN_OK;
}

int tgcn_relayout_qnc_to_nqc_f32(void* stream, const float* in, float* out, int64_t Q, int64_t n, int32_t C) {
  if (!in || !out || Q <= 0 || n <= 0 || C <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "relayout: bad argument");
  if (C > 32) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "relayout: C=%d > 32", C);
  const int64_t gy = (Q + kRelT - 1) / kRelT;
  if (gy > 65535) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "relayout: Q too large");
  const dim3 grid((unsigned)((n + kRelT - 1) / kRelT), (unsigned)gy);
  ProfScope ps(TGCN_PROF_RELAYOUT, (hipStream_t)stream);
  hipLaunchKernelGGL(relayout_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, in, out, Q, n, (int)C);
  TGCN_CHECK_LAUNCH("tgcn_relayout_qnc_to_nqc_f32");
  return TGCN_OK;
}

// Workspace layout of the layer forward (all offsets 256-byte aligned):
//   [xt]        n*q*C floats           (layout 1 only: re-laid input)
//   [hop 1..K-1] (K-1) * qc*n*C floats
//   [partial]   long-row segment scratch for one hop
static int fwd_nsets(int64_t q, int64_t qc, int32_t layout) {
  // two sets of hop tensors when there are several passes: the projection of pass i (side stream, MFMA-bound)
  // overlaps the hops of pass i+1 (memory-bound)
  return (layout == 0 && q > qc && g_overlap.load() != 0) ? 2 : 1;
}

static void fwd_ws_layout(const tgcn_csr_sched* S, int32_t K, int64_t q, int64_t n, int32_t C, int32_t layout,
                          int64_t qc, size_t* off_xt, size_t* off_hops, size_t* hop_bytes, size_t* off_part, size_t* total) {
  size_t o = 0;
  *off_xt = o;
  if (layout == 1) o += align_up((size_t)q * n * C * sizeof(float), 256);
  *off_hops = o;
  // consecutive hop tensors are staggered by an odd multiple of 256 B on top of their size: the projection streams
  // all K of them at once and equally aligned streams collide on the same DRAM channels (measured: -6 %)
  *hop_bytes = align_up((size_t)qc * n * C * sizeof(float), 256) + 65 * 256;
  o += (size_t)fwd_nsets(q, qc, layout) * (size_t)(K > 1 ? K - 1 : 0) * *hop_bytes;
  *off_part = o;
  const int32_t nb = layout == 1 ? 1 : (int32_t)qc;
  const int32_t Crow = layout == 1 ? (int32_t)(q * C) : C;
  o += align_up(tgcn_csr_hop_workspace_bytes(S, nb, Crow, 1), 256);
  *total = o;
}

int tgcn_cheb_forward_small_pool_supported(int64_t n, int64_t nnz, int32_t C, int32_t mode) {
  int dense = 0;
  return small_config(n, nnz, C, mode, &dense);
}

int tgcn_cheb_forward_small_supported(int64_t n, int64_t nnz, int32_t C, int32_t mode) {
  int dense = 0;
  const int ntc = small_config(n, nnz, C, mode, &dense);
  if (ntc) return ntc;
  return (g_small_dense.load() && dense_mfma_config(n, nnz, C, mode, 1, 1, false)) ? 16 : 0;
}

int tgcn_cheb_forward_small_f32(void* stream, const tgcn_csr* A, int32_t mode, int32_t K, int64_t q, int32_t C, int32_t N,
                                const float* x, const float* W, const float* fold, const float* bias, int32_t bias_kind,
                                float* out) {
  return tgcn_cheb_forward_small_pool_f32(stream, A, mode, K, q, C, N, x, W, fold, bias, bias_kind, 0, 0, out, nullptr);
}

int tgcn_cheb_forward_small_pool_f32(void* stream, const tgcn_csr* A, int32_t mode, int32_t K, int64_t q, int32_t C, int32_t N,
                                     const float* x, const float* W, const float* fold, const float* bias, int32_t bias_kind,
                                     int32_t relu, int32_t pool, float* out, uint8_t* pool_idx) {
  if (!A || !x || !W || !out || K < 1 || q < 1 || N < 1) TGCN_FAIL(TGCN_ERR_INVALID, "forward_small: bad argument");
  if (pool < 0 || pool > 255 || (pool > 0 && A->n % pool != 0) || (pool == 0 && relu)) TGCN_FAIL(TGCN_ERR_INVALID, "forward_small: pool=%d relu=%d n=%lld", pool, relu, (long long)A->n);
  if (bias_kind < 0 || bias_kind > 2 || (bias_kind && !bias)) TGCN_FAIL(TGCN_ERR_INVALID, "forward_small: bias_kind %d", bias_kind);
  if (fold && mode != 0) TGCN_FAIL(TGCN_ERR_INVALID, "forward_small: fold is for mode 0");
  if (q > 2147483647LL) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "forward_small: grid too large");
  int dense = 0;
  SmallParams p;
  p.rowptr = A->rowptr; p.ev = A->edges; p.x = x; p.W = W; p.fold = fold; p.bias = bias; p.out = out;
  p.n = (int32_t)A->n; p.nnz = (int32_t)A->nnz; p.q = (int32_t)q; p.K = K; p.C = C; p.N = N; p.mode = mode; p.bias_kind = bias_kind; p.dense = 0; p.relu = relu; p.pool = pool; p.pool_idx = pool_idx;
  p.npad = 0; p.spw = 1;
  p.Ld = A->dense;
  if (pool == 0 && A->dense && g_small_dense.load()) {   // dense operand (e.g. the 148-parcel DTI graph): fp32 matrix pipe
    const int64_t tiles16 = (N + 15) / 16;
    const int S = dense_mfma_config(A->n, A->nnz, C, mode, q, tiles16, false);
    if (S && tiles16 <= 65535) {
      p.npad = 0; p.spw = S;
      hipStream_t st = (hipStream_t)stream;
      ProfScope ps(TGCN_PROF_SMALL, st);
      launch_small_dense<false>(st, p, S, tiles16);
      TGCN_CHECK_LAUNCH("tgcn_cheb_forward_small_f32 (dense)");
      return TGCN_OK;
    }
  }
  const int ntc = small_config(A->n, A->nnz, C, mode, &dense);
  if (!ntc) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "forward_small: n=%lld nnz=%lld C=%d does not fit in LDS", (long long)A->n, (long long)A->nnz, C);
  if ((N + ntc - 1) / ntc > 65535) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "forward_small: grid too large");
  p.dense = dense;
  if (C <= 4 && g_small_narrow.load() && (pool == 0 || (pool <= 64 && (64 % pool) == 0))) {
    // first layers: recursion on the 4-float input side, all of NT output channels per workgroup (small_narrow_kernel)
    p.npad = (p.n + 63) / 64 * 64;
    int nt = N > 32 ? 64 : (N > 16 ? 32 : 16);
    while (nt > 16 && q * ((N + nt - 1) / nt) < 256) nt /= 2;            // fill the chip before widening the tile
    int spw = 1;
    const int64_t tiles = (N + nt - 1) / nt;
    while ((spw + 1) * p.npad <= kSmallMaxN && narrow_lds_bytes(p.n, p.nnz, nt, mode, dense, spw + 1) <= 160 * 1024 &&
           (q + spw) / (spw + 1) * tiles >= 512)
      ++spw;
    if (narrow_lds_bytes(p.n, p.nnz, nt, mode, dense, spw) <= 160 * 1024) {
      p.spw = spw;
      const size_t lds = narrow_lds_bytes(p.n, p.nnz, nt, mode, dense, spw);
      const dim3 grid((unsigned)((q + spw - 1) / spw), (unsigned)tiles);
      hipStream_t st = (hipStream_t)stream;
      const unsigned nthreads = (unsigned)(p.npad * spw);
      ProfScope ps(TGCN_PROF_SMALL, st);
#define TGCN_NARROW(NTV)                                                                     \
  {                                                                                          \
    allow_large_lds((const void*)small_narrow_kernel<NTV>, 160 * 1024);                      \
    hipLaunchKernelGGL((small_narrow_kernel<NTV>), grid, dim3(nthreads), lds, st, p);        \
  }
      if (nt == 64) TGCN_NARROW(64) else if (nt == 32) TGCN_NARROW(32) else TGCN_NARROW(16)
#undef TGCN_NARROW
      TGCN_CHECK_LAUNCH("tgcn_cheb_forward_small_f32 (narrow input)");
      return TGCN_OK;
    }
  }
  // samples per workgroup: as many as fit the 1024-thread / 160 KB budget, but keep >= 512 workgroups in the grid
  p.npad = (p.n + 63) / 64 * 64;
  int spw = 1;
  const int64_t col_tiles = (N + ntc - 1) / ntc;
  while ((spw + 1) * p.npad <= kSmallMaxN && small_lds_bytes(p.n, p.nnz, C, ntc, mode, dense, spw + 1) <= 160 * 1024 &&
         (q + spw) / (spw + 1) * col_tiles >= 512)
    ++spw;
  p.spw = spw;
  const size_t lds = small_lds_bytes(p.n, p.nnz, C, ntc, mode, dense, spw);
  const dim3 grid((unsigned)((q + spw - 1) / spw), (unsigned)col_tiles);
  hipStream_t st = (hipStream_t)stream;
  const unsigned nthreads = (unsigned)(p.npad * p.spw);
  ProfScope ps(TGCN_PROF_SMALL, st);
#define TGCN_SMALL(NTCV, CPV)                                                                                     \
  {                                                                                                               \
    allow_large_lds((const void*)small_forward_kernel<NTCV, CPV>, 160 * 1024);                                      \
    hipLaunchKernelGGL((small_forward_kernel<NTCV, CPV>), grid, dim3(nthreads), lds, st, p);                      \
  }
  if (ntc == 16) { if (C <= 4) TGCN_SMALL(16, 4) else if (C <= 16) TGCN_SMALL(16, 16) else TGCN_SMALL(16, 32) }
  else { if (C <= 4) TGCN_SMALL(8, 4) else if (C <= 16) TGCN_SMALL(8, 16) else TGCN_SMALL(8, 32) }
#undef TGCN_SMALL
  TGCN_CHECK_LAUNCH("tgcn_cheb_forward_small_f32");
  return TGCN_OK;
}

int tgcn_cheb_basis_small_supported(int64_t n, int64_t nnz, int32_t C, int32_t mode) {
  int dense = 0;
  const int ct = basis_config(n, nnz, C, mode, &dense);
  if (ct) return ct;
  return (g_small_dense.load() && dense_mfma_config(n, nnz, C, mode, 1, 1, true)) ? 16 : 0;
}

int tgcn_cheb_basis_small_f32(void* stream, const tgcn_csr* A, int32_t mode, int32_t K, int64_t q, int32_t C,
                              const float* x, float* stack) {
  if (!A || !x || !stack || K < 1 || q < 1 || C < 1) TGCN_FAIL(TGCN_ERR_INVALID, "basis_small: bad argument");
  if (!tgcn_cheb_basis_small_supported(A->n, A->nnz, C, mode))
    TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "basis_small: n=%lld nnz=%lld does not fit in LDS", (long long)A->n, (long long)A->nnz);
  if (K == 1) return TGCN_OK;
  if (q > 2147483647LL || C > 16 * 65535) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "basis_small: grid too large");
  SmallParams p;
  memset(&p, 0, sizeof(p));
  p.rowptr = A->rowptr; p.ev = A->edges; p.x = x; p.out = stack;
  p.n = (int32_t)A->n; p.nnz = (int32_t)A->nnz; p.q = (int32_t)q; p.K = K; p.C = C; p.mode = mode;
  p.Ld = A->dense;
  if (A->dense && g_small_dense.load()) {
    const int64_t tiles16 = (C + 15) / 16;
    const int S = dense_mfma_config(A->n, A->nnz, C, mode, q, tiles16, true);
    if (S) {
      p.spw = S;
      hipStream_t st = (hipStream_t)stream;
      ProfScope ps(TGCN_PROF_SMALL_BASIS, st);
      launch_small_dense<true>(st, p, S, tiles16);
      TGCN_CHECK_LAUNCH("tgcn_cheb_basis_small_f32 (dense)");
      return TGCN_OK;
    }
  }
  int dense = 0;
  const int ct = basis_config(A->n, A->nnz, C, mode, &dense);
  if (!ct) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "basis_small: n=%lld nnz=%lld does not fit in LDS", (long long)A->n, (long long)A->nnz);
  const int64_t col_tiles = (C + ct - 1) / ct;
  p.dense = dense;
  p.npad = (p.n + 63) / 64 * 64;
  int spw = 1;
  while ((spw + 1) * p.npad <= kSmallMaxN && basis_lds_bytes(p.n, p.nnz, ct, mode, dense, spw + 1) <= 160 * 1024 &&
         (q + spw) / (spw + 1) * col_tiles >= 512)
    ++spw;
  p.spw = spw;
  const size_t lds = basis_lds_bytes(p.n, p.nnz, ct, mode, dense, spw);
  const dim3 grid((unsigned)((q + spw - 1) / spw), (unsigned)col_tiles);
  hipStream_t st = (hipStream_t)stream;
  const unsigned nthreads = (unsigned)(p.npad * p.spw);
  ProfScope ps(TGCN_PROF_SMALL_BASIS, st);
#define TGCN_BASIS(CTV)                                                                    \
  {                                                                                        \
    allow_large_lds((const void*)small_basis_kernel<CTV>, 160 * 1024);                     \
    hipLaunchKernelGGL((small_basis_kernel<CTV>), grid, dim3(nthreads), lds, st, p);       \
  }
  if (ct == 16) TGCN_BASIS(16) else if (ct == 8) TGCN_BASIS(8) else TGCN_BASIS(4)
#undef TGCN_BASIS
  TGCN_CHECK_LAUNCH("tgcn_cheb_basis_small_f32");
  return TGCN_OK;
}

size_t tgcn_cheb_forward_workspace_bytes(const tgcn_csr_sched* S, int32_t K, int64_t q, int64_t n, int32_t C,
                                         int32_t layout, int64_t q_chunk) {
  if (!S || K < 1 || q < 1 || n < 1 || C < 1) return 0;
  const int64_t qc = (layout == 1 || q_chunk <= 0 || q_chunk > q) ? q : q_chunk;
  size_t a, b, c, d, total;
  fwd_ws_layout(S, K, q, n, C, layout, qc, &a, &b, &c, &d, &total);
  return total;
}

int tgcn_cheb_forward_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t mode, int32_t K,
                          int64_t q, int64_t n, int32_t C, int32_t N, const float* x, const float* W,
                          const float* bias, int32_t bias_kind, float* out, int32_t layout, int64_t q_chunk,
                          void* workspace, size_t workspace_bytes) {
  if (!A || !S || !x || !W || !out) TGCN_FAIL(TGCN_ERR_INVALID, "forward: null operand");
  if (K < 1 || q < 1 || n < 1 || C < 1 || N < 1 || n != A->n) TGCN_FAIL(TGCN_ERR_INVALID, "forward: bad shape (n=%lld, L is %lld)", (long long)n, (long long)A->n);
  if (mode != 0 && mode != 1) TGCN_FAIL(TGCN_ERR_INVALID, "forward: mode %d", mode);
  if (layout != 0 && layout != 1) TGCN_FAIL(TGCN_ERR_INVALID, "forward: layout %d", layout);
  if (layout == 1 && (C > 32 || q * C > (int64_t)INT32_MAX)) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "forward: layout 1 needs C <= 32");
  if (((uintptr_t)x & 15) || ((uintptr_t)workspace & 15)) TGCN_FAIL(TGCN_ERR_INVALID, "forward: x/workspace must be 16-byte aligned");
  const int64_t qc = (layout == 1 || q_chunk <= 0 || q_chunk > q) ? q : q_chunk;
  size_t off_xt, off_hops, hop_bytes, off_part, total;
  fwd_ws_layout(S, K, q, n, C, layout, qc, &off_xt, &off_hops, &hop_bytes, &off_part, &total);
  if (total > 0 && (!workspace || workspace_bytes < total)) TGCN_FAIL(TGCN_ERR_WORKSPACE, "forward: workspace %zu < %zu", workspace_bytes, total);
  char* ws = (char*)workspace;
  float* part = (float*)(ws + off_part);
  const size_t part_bytes = total - off_part;
  const float* terms[kMaxTerms];
  int64_t ldas[kMaxTerms];
  int rc;

  const int nsets = fwd_nsets(q, qc, layout);
  SideStream* side = nsets == 2 ? side_stream() : nullptr;
  hipStream_t main_st = (hipStream_t)stream;
  const size_t set_bytes = (size_t)(K > 1 ? K - 1 : 0) * hop_bytes;
  int pass = 0;
  for (int64_t q0 = 0; q0 < q; q0 += qc, ++pass) {
    const int64_t qn = (q - q0 < qc) ? (q - q0) : qc;
    const int set = side ? (pass & 1) : 0;
    // operand view of this pass
    int32_t nb, Crow;
    const float* x0;
    if (layout == 1) {
      float* xt = (float*)(ws + off_xt);
      if ((rc = tgcn_relayout_qnc_to_nqc_f32(stream, x, xt, q, n, C)) != TGCN_OK) return rc;
      x0 = xt; nb = 1; Crow = (int32_t)(q * C);
    } else {
      x0 = x + q0 * n * C; nb = (int32_t)qn; Crow = C;
    }
    const int64_t bs = (int64_t)n * Crow;
    auto hop_ptr = [&](int k) -> float* {
      return k == 0 ? const_cast<float*>(x0) : (float*)(ws + off_hops + (size_t)set * set_bytes + (size_t)(k - 1) * hop_bytes);
    };
    // this set was last read by the projection of pass-2: wait for it before overwriting
    if (side && pass >= 2 && hipStreamWaitEvent(main_st, side->proj_done[set], 0) != hipSuccess) TGCN_FAIL(TGCN_ERR_LAUNCH, "forward: stream wait failed");
    for (int k = 1; k < K; ++k) {
      tgcn_dense X = {hop_ptr(k - 1), bs, Crow};
      tgcn_dense Y = {hop_ptr(k), bs, Crow};
      if (mode == 0 || k == 1) {
        rc = tgcn_csr_hop_f32(stream, A, S, nb, Crow, &X, nullptr, 1.f, 0.f, &Y, nullptr, part, part_bytes);
      } else {
        tgcn_dense Zd = {hop_ptr(k - 2), bs, Crow};
        rc = tgcn_csr_hop_f32(stream, A, S, nb, Crow, &X, &Zd, 2.f, -1.f, &Y, nullptr, part, part_bytes);
      }
      if (rc != TGCN_OK) return rc;
    }
    void* proj_stream = stream;
    if (side) {
      if (hipEventRecord(side->hops_done[set], main_st) != hipSuccess || hipStreamWaitEvent(side->st, side->hops_done[set], 0) != hipSuccess)
        TGCN_FAIL(TGCN_ERR_LAUNCH, "forward: stream fork failed");
      proj_stream = side->st;
    }
    // projection, in chunks of <= 32 terms
    const int64_t M = (layout == 1) ? n * q : qn * n;
    float* o0 = (layout == 1) ? out : out + q0 * n * N;
    for (int k0 = 0; k0 < K; k0 += kMaxTerms) {
      const int nt = (K - k0 < kMaxTerms) ? K - k0 : kMaxTerms;
      for (int t = 0; t < nt; ++t) { terms[t] = hop_ptr(k0 + t); ldas[t] = C; }
      const bool last = (k0 + nt >= K);
      rc = tgcn_cheb_project_f32(proj_stream, M, C, N, nt, terms, ldas, W + (size_t)k0 * C * N, last ? bias : nullptr,
                                 last ? bias_kind : 0, n, layout == 1 ? q : 1, k0 > 0 ? 1 : 0, o0, N);
      if (rc != TGCN_OK) return rc;
    }
    if (side && hipEventRecord(side->proj_done[set], side->st) != hipSuccess) TGCN_FAIL(TGCN_ERR_LAUNCH, "forward: event record failed");
  }
  if (side) {  // join: everything the side stream did is ordered before whatever the caller enqueues next
    for (int i = 0; i < 2 && i < pass; ++i)
      if (hipStreamWaitEvent(main_st, side->proj_done[i], 0) != hipSuccess) TGCN_FAIL(TGCN_ERR_LAUNCH, "forward: stream join failed");
  }
  return TGCN_OK;
}

int tgcn_relu_pool_f32(void* stream, const float* x, float* out, uint8_t* idx, int64_t q, int64_t n, int32_t f, int32_t p) {
  if (!x || !out || q <= 0 || n <= 0 || f <= 0 || p <= 0 || p > 255 || n % p != 0) TGCN_FAIL(TGCN_ERR_INVALID, "relu_pool: bad argument");
  const int64_t total = q * (n / p) * f;
  hipLaunchKernelGGL(relu_pool_kernel, dim3(grid_1d(total)), dim3(kBlock), 0, (hipStream_t)stream, x, out, idx, total, (int)f, (int)p);
  TGCN_CHECK_LAUNCH("tgcn_relu_pool_f32");
  return TGCN_OK;
}

int tgcn_relu_pool_bwd_f32(void* stream, const float* grad_z, const float* z, const uint8_t* idx, float* grad_y, int64_t q,
                           int64_t n, int32_t f, int32_t p) {
  if (!grad_z || !z || !idx || !grad_y || q <= 0 || n <= 0 || f <= 0 || p <= 0 || n % p != 0) TGCN_FAIL(TGCN_ERR_INVALID, "relu_pool_bwd: bad argument");
  const int64_t total = q * (n / p) * f;
  hipLaunchKernelGGL(relu_pool_bwd_kernel, dim3(grid_1d(total)), dim3(kBlock), 0, (hipStream_t)stream, grad_z, z, idx, grad_y, total, (int)f, (int)p);
  TGCN_CHECK_LAUNCH("tgcn_relu_pool_bwd_f32");
  return TGCN_OK;
}

// Workspace of the project-first path: Z (q*n x K*N) + 3 result buffers (q*n x N) + hop scratch.
static void pf_ws_layout(const tgcn_csr_sched* S, int32_t K, int64_t q, int64_t n, int32_t N, size_t* z_bytes, size_t* y_bytes,
                         size_t* off_part, size_t* total) {
  *z_bytes = align_up((size_t)q * n * K * N * sizeof(float), 256);
  *y_bytes = align_up((size_t)q * n * N * sizeof(float), 256);
  *off_part = *z_bytes + 3 * *y_bytes;
  *total = *off_part + align_up(tgcn_csr_hop_workspace_bytes(S, (int32_t)q, N, N % 4 == 0), 256);
}

size_t tgcn_cheb_forward_pf_workspace_bytes(const tgcn_csr_sched* S, int32_t K, int64_t q, int64_t n, int32_t N) {
  if (!S || K < 1 || q < 1 || n < 1 || N < 1) return 0;
  size_t a, b, c, t;
  pf_ws_layout(S, K, q, n, N, &a, &b, &c, &t);
  return t;
}

int tgcn_cheb_forward_pf_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t mode, int32_t K, int64_t q,
                             int64_t n, int32_t C, int32_t N, const float* x, const float* Wcat, const float* bias,
                             int32_t bias_kind, float* out, void* workspace, size_t workspace_bytes) {
  if (!A || !S || !x || !Wcat || !out) TGCN_FAIL(TGCN_ERR_INVALID, "forward_pf: null operand");
  if (K < 1 || q < 1 || n < 1 || C < 1 || N < 1 || n != A->n || q > 65535) TGCN_FAIL(TGCN_ERR_INVALID, "forward_pf: bad shape");
  if (mode != 0 && mode != 1) TGCN_FAIL(TGCN_ERR_INVALID, "forward_pf: mode %d", mode);
  size_t z_bytes, y_bytes, off_part, total;
  pf_ws_layout(S, K, q, n, N, &z_bytes, &y_bytes, &off_part, &total);
  if (!workspace || workspace_bytes < total || ((uintptr_t)workspace & 15)) TGCN_FAIL(TGCN_ERR_WORKSPACE, "forward_pf: workspace %zu < %zu", workspace_bytes, total);
  char* ws = (char*)workspace;
  const int64_t M = q * n, KN = (int64_t)K * N;
  float* Zb = (K == 1) ? out : (float*)ws;      // K == 1: the projection IS the layer
  const float* a1[1] = {x};
  const int64_t lda1[1] = {C};
  int rc = project_impl(stream, M, C, (int32_t)KN, 1, a1, lda1, Wcat, bias, bias_kind, n, 1, 0, Zb, KN, 0, 0, N);
  if (rc != TGCN_OK || K == 1) return rc;
  float* part = (float*)(ws + off_part);
  const size_t part_bytes = total - off_part;
  auto zview = [&](int j) { return tgcn_dense{Zb + (int64_t)j * N, n * KN, KN}; };
  auto ybuf = [&](int i) { return tgcn_dense{(float*)(ws + z_bytes + (size_t)i * y_bytes), n * (int64_t)N, N}; };
  const tgcn_dense outd = {out, n * (int64_t)N, N};
  if (mode == 0) {            // Horner: Y_j = Z_j + L Y_{j+1}
    tgcn_dense cur = zview(K - 1);
    for (int j = K - 2; j >= 0; --j) {
      const tgcn_dense zj = zview(j);
      const tgcn_dense dst = (j == 0) ? outd : ybuf(j & 1);
      rc = tgcn_csr_hop2_f32(stream, A, S, (int32_t)q, N, &cur, &zj, 1.f, 1.f, nullptr, 0.f, &dst, nullptr, part, part_bytes);
      if (rc != TGCN_OK) return rc;
      cur = dst;
    }
  } else {                    // Clenshaw: b_k = Z_k + 2 L b_{k+1} - b_{k+2};  out = Z_0 + L b_1 - b_2
    tgcn_dense b1 = zview(K - 1), b2 = {nullptr, 0, 0};
    for (int k = K - 2; k >= 0; --k) {
      const tgcn_dense zk = zview(k);
      const tgcn_dense dst = (k == 0) ? outd : ybuf(k % 3);
      rc = tgcn_csr_hop2_f32(stream, A, S, (int32_t)q, N, &b1, b2.ptr ? &b2 : nullptr, k == 0 ? 1.f : 2.f, -1.f, &zk, 1.f, &dst,
                             nullptr, part, part_bytes);
      if (rc != TGCN_OK) return rc;
      b2 = b1;
      b1 = dst;
    }
  }
  return TGCN_OK;
}

int tgcn_pool_max_f32(void* stream, const float* x, float* out, int32_t* idx, int64_t q, int64_t n, int32_t f, int32_t p) {
  if (!x || !out || q <= 0 || n <= 0 || f <= 0 || p <= 0 || n % p != 0) TGCN_FAIL(TGCN_ERR_INVALID, "pool: bad argument (n=%lld p=%d)", (long long)n, p);
  const int64_t total = q * (n / p) * f;
  hipLaunchKernelGGL(pool_max_kernel, dim3(grid_1d(total)), dim3(kBlock), 0, (hipStream_t)stream, x, out, idx, total, (int)f, (int)p);
  TGCN_CHECK_LAUNCH("tgcn_pool_max_f32");
  return TGCN_OK;
}

int tgcn_pool_max_bwd_f32(void* stream, const float* grad_out, const int32_t* idx, float* grad_in, int64_t q, int64_t n, int32_t f, int32_t p) {
  if (!grad_out || !idx || !grad_in || q <= 0 || n <= 0 || f <= 0 || p <= 0 || n % p != 0) TGCN_FAIL(TGCN_ERR_INVALID, "pool_bwd: bad argument");
  const int64_t total = q * (n / p) * f;
  hipLaunchKernelGGL(pool_max_bwd_kernel, dim3(grid_1d(total)), dim3(kBlock), 0, (hipStream_t)stream, grad_out, idx, grad_in, total, (int)f, (int)p);
  TGCN_CHECK_LAUNCH("tgcn_pool_max_bwd_f32");
  return TGCN_OK;
}

}  // extern "C"
