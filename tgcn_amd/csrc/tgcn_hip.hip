// tgcn_hip.hip -- gfx950 (MI355X) kernels + C ABI for the Chebyshev (time-)graph convolution.
// See include/tgcn_hip.h for the contract and DESIGN.md for layout / roofline notes.
//
// One translation unit; the kernels live in topic headers included once inside the anonymous namespace below:
//   common.h         error reporting, launch timing (tgcn_profile_*), tuning switches, LDS attribute bookkeeping
//   hop.h            hop_kernel<LPR,VEC,U,R> / hop_fixup_kernel: row-block + column-ordered-segment CSR x dense rows,
//                    fused Y = alpha (L X) + beta Z (+ gamma Z2) (+ P = L X); deterministic segment fold
//   project.h        out = sum_t A_t W_t + bias: exact fp32 MFMA (streaming / W-resident), bf16x3 (two forms), narrow VALU
//   wgrad.h          dW_t = A_t^T G, two deterministic stages on the fp32 MFMA
//   small_graph.h    graphs that fit in LDS: whole layer / basis in ONE launch (sparse, first-layer, dense matrix-pipe)
//   pool_relayout.h  (Q,n,C) -> (n,Q,C), gcn_pool / gcn_pool_4, relu + pool pass
//   device_build.h   operand / schedule construction on the device: prefix sums, stable radix sort, CSR build, schedule kernels
//   graph_build.h    tgcn_graph_* / tgcn_sched_* / tgcn_csr_build_f32: host-side orchestration of device_build.h (one-off per operand)
// This file: the extern "C" entry points (argument checks, workspace carving, launches) declared in tgcn_hip.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <math.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <numeric>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "tgcn_hip.h"

namespace {

#include "common.h"
#include "hop.h"
#include "project.h"
#include "wgrad.h"
#include "small_graph.h"
#include "pool_relayout.h"
#include "device_build.h"
#include "graph_build.h"

}  // namespace

// ==================================================================================================
// C ABI
// ==================================================================================================
extern "C" {

const char* tgcn_last_error(void) { return g_err; }
int tgcn_abi_version(void) { return TGCN_ABI_VERSION; }

void tgcn_reset_tuning(void) {
  g_hop_variant.store(0); g_hop_remap.store(1); g_hop_seg_remap.store(0); g_hop_mix.store(0); g_hop_stream.store(1); g_hop_lds_pad.store(0);
  g_proj_variant.store(0); g_small_dense.store(2); g_small_narrow.store(1); g_x3_form.store(2); g_compact_proj.store(0); g_x3_tail.store(1); g_fuse_last.store(0);
  g_overlap.store(0);
}

int tgcn_set_tuning(const char* key, int32_t value) {
  if (key && strcmp(key, "hop_variant") == 0) { g_hop_variant.store(value); return TGCN_OK; }
  if (key && strcmp(key, "hop_xcd_remap") == 0) { g_hop_remap.store(value != 0); return TGCN_OK; }
  if (key && strcmp(key, "hop_seg_remap") == 0) { g_hop_seg_remap.store(value != 0); return TGCN_OK; }
  if (key && strcmp(key, "hop_mix") == 0) { g_hop_mix.store(value); return TGCN_OK; }
  if (key && strcmp(key, "hop_stream") == 0) { g_hop_stream.store(value != 0); return TGCN_OK; }
  if (key && strcmp(key, "hop_lds_pad") == 0) { if (value < 0 || value > 160 * 1024) TGCN_FAIL(TGCN_ERR_INVALID, "set_tuning: hop_lds_pad %d", value); g_hop_lds_pad.store(value); return TGCN_OK; }
  if (key && strcmp(key, "project_variant") == 0) { g_proj_variant.store(value); return TGCN_OK; }
  if (key && strcmp(key, "small_dense") == 0) { g_small_dense.store(value); return TGCN_OK; }
  if (key && strcmp(key, "small_narrow") == 0) { g_small_narrow.store(value); return TGCN_OK; }
  if (key && strcmp(key, "x3_form") == 0) { g_x3_form.store(value); return TGCN_OK; }
  if (key && strcmp(key, "compact_proj") == 0) { g_compact_proj.store(value); return TGCN_OK; }
  if (key && strcmp(key, "x3_tail") == 0) { g_x3_tail.store(value != 0); return TGCN_OK; }
  if (key && strcmp(key, "fuse_last_hop") == 0) { g_fuse_last.store(value != 0); return TGCN_OK; }
  if (key && strcmp(key, "overlap") == 0) { g_overlap.store(value); return TGCN_OK; }
  TGCN_FAIL(TGCN_ERR_INVALID, "set_tuning: unknown key");
}

int tgcn_profile_start(int32_t capacity) {
  if (capacity <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "profile: capacity %d", capacity);
  for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof.clear();
  g_prof.reserve(capacity);
  g_prof_cap = capacity;
  return TGCN_OK;
}

int tgcn_profile_stop(int32_t* kinds, float* ms, int32_t capacity, int32_t* count) {
  g_prof_cap = 0;
  int n = 0;
  for (auto& r : g_prof) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) t = -1.f;
    if (n < capacity && kinds && ms) { kinds[n] = r.kind; ms[n] = t; ++n; }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
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

static int hop_impl(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t nb, int32_t C,
                    const tgcn_dense* X, const tgcn_dense* Z, float alpha, float beta, const tgcn_dense* Z2, float gamma,
                    const tgcn_dense* Y, const tgcn_dense* P, void* workspace, size_t workspace_bytes, int long_rows_only);

int tgcn_csr_hop2_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t nb, int32_t C,
                      const tgcn_dense* X, const tgcn_dense* Z, float alpha, float beta, const tgcn_dense* Z2, float gamma,
                      const tgcn_dense* Y, const tgcn_dense* P, void* workspace, size_t workspace_bytes) {
  return hop_impl(stream, A, S, nb, C, X, Z, alpha, beta, Z2, gamma, Y, P, workspace, workspace_bytes, 0);
}

// long_rows_only: the rows of more than row_thresh entries only (whole-row wave segments, lane-group segments + fix-up) -- the others are
// left to the caller (the projection with the fused last hop gathers them itself); the rows it skips are not written.
static int hop_impl(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t nb, int32_t C,
                    const tgcn_dense* X, const tgcn_dense* Z, float alpha, float beta, const tgcn_dense* Z2, float gamma,
                    const tgcn_dense* Y, const tgcn_dense* P, void* workspace, size_t workspace_bytes, int long_rows_only) {
  if (!A || !S || !X || !X->ptr) TGCN_FAIL(TGCN_ERR_INVALID, "hop: null operand");
  if (int drc = check_pointer_device(X->ptr, (hipStream_t)stream, "hop")) return drc;
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
  if (S->seg_mode != 0 && !(S->seg_mode == 1 && g.lpr < 64)) TGCN_FAIL(TGCN_ERR_INVALID, "hop: seg_mode %d with %d lanes per row", S->seg_mode, g.lpr);
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
  p.C = C; p.nb = nb; p.nchunks = g.nchunks; p.cpad = g.cpad; p.remap = g_hop_remap.load();
  p.seg_mode = S->seg_mode; p.seg_remap = g_hop_seg_remap.load();
  if (S->nwseg < 0 || S->nwseg > S->nseg || (S->nwseg > 0 && (S->seg_mode != 0 || g.lpr >= 64))) TGCN_FAIL(TGCN_ERR_INVALID, "hop: nwseg %d of %d segments", S->nwseg, S->nseg);
  p.nwseg = S->nwseg;
  {
    // the REQUEST only (1: row blocks dealt among the segment blocks, -1: segment blocks first); launch_hop turns it into the
    // period once it knows the real number of segment blocks of the kernel it launches (rows interleaved per group differ by variant)
    const int mix = g_hop_mix.load();
    p.mix_period = (mix == 1 || (mix == 0 && S->row_mix)) ? 1 : (mix == 2 ? -1 : 0);
  }
  p.stream_out = ((int64_t)A->n * C * (int64_t)sizeof(float) * nb > ((int64_t)256 << 20)) && g_hop_stream.load();
  if (long_rows_only) {
    if (S->nseg == 0) return TGCN_OK;       // no row above the threshold: nothing to do
    p.nblk = 0;                             // no row blocks: every workgroup of the launch is a segment block
    p.mix_period = 0;
    p.long_rows_only = 1;                   // (a FULL hop on a schedule without row blocks is still a TGCN_PROF_HOP record: ADVICE r04)
  }
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
                        int32_t win_n, int32_t win_t, int32_t bias_cols = -1, const int32_t* rowmap = nullptr, uint32_t mapped = 0,
                        int32_t nbatch = 1, const int64_t* a_bs = nullptr, int64_t out_bs = 0, int32_t pool = 0, uint8_t* pool_idx = nullptr,
                        const struct ProjGather* gather = nullptr);

// The last hop fused into the projection (ProjParams.g_*): operand, gather source and the term it produces.
struct ProjGather {
  const tgcn_csr* A;      // rows = the projection's tile rows (compact rows), columns index rows of X
  const float* X;         // the previous hop tensor (row stride = the row length Kc), sample stride xbs floats
  int64_t xbs;
  int32_t term, thresh;   // term whose tiles are gathered for rows of at most `thresh` entries
};

// THE dispatch of the projection: which kernel a shape takes.  project_impl launches what this returns and project_pool_fusable asks the
// same function, so the fused relu + pool epilogue can never be requested from a kernel that does not have it.
// project_variant: 0 auto (vector-ALU kernel for a few scalars per row, bf16x3 on large problems, else exact fp32: W-resident when the
// weight fits, streaming otherwise), 1 exact-fp32 streaming, 2 exact-fp32 (W-resident with 16-row wave tiles when it fits), 3 bf16x3 always,
// 4 exact fp32 auto, 5 vector-ALU kernel whenever it applies, 6 the streaming bf16x3 kernel whenever the shape has it (also below its row threshold).
enum ProjKernel { kProjNarrow, kProjResident, kProjX3, kProjX3Wide, kProjStream, kProjX3Stream };
constexpr int64_t kX3StreamMinRows = 32768;          // fewer rows: the tiled kernels (a persistent grid of 4096 waves wants >= a few tiles each)
constexpr size_t kX3StreamMaxLds = 150 * 1024;       // the three bf16 planes of the whole weight in fragment order
static inline int x3_stream_nt(int32_t N) { return N <= 16 ? 1 : (N <= 32 ? 2 : 4); }
static inline size_t x3_stream_lds(int32_t Kc, int32_t N, int32_t nterms) { return (size_t)nterms * (Kc / 32) * x3_stream_nt(N) * 3 * 1024; }
struct ProjChoice {
  ProjKernel kernel;
  int nt;            // 16-column tiles per workgroup of the W-resident kernel
  int nts;           // 16-column tiles per workgroup of the streaming / bf16x3 kernels
  size_t wbytes;     // LDS image of the weight for the W-resident kernel
  bool pool_epilogue;   // the kernel can end in relu + max over consecutive rows (through its vector epilogue)
  int pool_max;         // ... over groups whose size divides this: 16 (rows of a wave's tile in LDS scratch), 4 in the wide bf16x3 kernel (a lane's four accumulator rows)
};
// stream_ok: the call has nothing the streaming kernel lacks (pool epilogue, fused last hop, accumulate, interleave, windows) -- project_impl knows,
// the shape-only queries (pool / gather fusability) pass false: those forms live in project_x3_kernel.
static ProjChoice project_choose(int64_t M, int32_t Kc, int32_t N, int32_t nterms, bool vec4, bool vec_epilogue, bool has_rowmap, bool windows,
                                 bool stream_ok = false, int32_t nbatch = 1) {
  ProjChoice c;
  const int pv = g_proj_variant.load();
  c.nt = N <= 16 ? 1 : (N <= 32 ? 2 : 4);
  c.wbytes = (size_t)nterms * ((Kc + 3) / 4 * 4) * c.nt * 16 * sizeof(float);
  // streaming-W / bf16x3 kernels: widest column tile that keeps padding low, so A is read once per block and no MFMA works
  // on padding (N = 160 -> one block of 10 tiles instead of three of 4)
  const int tiles = (N + 15) / 16;
  c.nts = tiles <= 1 ? 1 : tiles <= 2 ? 2 : tiles <= 4 ? 4 : tiles <= 6 ? 6 : tiles <= 8 ? 8 : 10;
  if (tiles > 10) {   // several column blocks: the width with the least padded tiles
    int best = 10, waste = (10 - tiles % 10) % 10;
    for (int w : {8, 6, 4}) { const int ws = (w - tiles % w) % w; if (ws < waste) { waste = ws; best = w; } }
    c.nts = best;
  }
  c.pool_epilogue = false;
  c.pool_max = 16;
  (void)has_rowmap;     // every kernel reads mapped terms / writes mapped rows through proj_arow / proj_orow
  if ((pv == 0 || pv == 5) && (int64_t)Kc * nterms <= kNarrowMaxK && !windows && vec_epilogue && N <= 1024 && (M >= 4096 || pv == 5)) {
    c.kernel = kProjNarrow;           // a few scalars per row: the output streams from the vector ALU
    return c;
  }
  const bool use_x3 = pv == 3 || (pv == 0 && M >= 8192 && (int64_t)Kc * nterms >= 64);
  // rows of 32 / 64 floats, <= 64 output columns, whole weight resident as bf16 planes: the barrier-free streaming form
  if (stream_ok && !windows && vec4 && vec_epilogue && (Kc == 32 || Kc == 64) && N <= 64 && x3_stream_lds(Kc, N, nterms) <= kX3StreamMaxLds &&
      x3_stream_lds(Kc, N, nterms) <= (size_t)lds_optin_limit() && M < (int64_t)INT32_MAX &&
      /* n_vertices is checked by the caller of this function where it matters: project_impl passes it through stream_ok */
      (pv == 6 || (pv == 0 && use_x3 && M * (int64_t)nbatch >= kX3StreamMinRows))) {
    c.kernel = kProjX3Stream;
    return c;
  }
  if (c.wbytes <= (size_t)kResMaxWBytes && pv != 1 && !use_x3) {
    c.kernel = kProjResident;
    c.pool_epilogue = vec_epilogue;
    return c;
  }
  if (use_x3) {
    c.kernel = (vec4 && c.nts >= 6 && g_x3_form.load() == 2) ? kProjX3Wide : kProjX3;      // wide outputs: A fragments from registers
    c.pool_epilogue = (c.kernel == kProjX3 && c.nts <= 4 && vec_epilogue) || c.kernel == kProjX3Wide;
    if (c.kernel == kProjX3Wide) c.pool_max = 4;
    return c;
  }
  c.kernel = kProjStream;
  return c;
}

// Whether the projection of this shape (aligned operands) takes the kernel that has the gathering form (fused last hop).
static bool project_gather_fusable(int64_t M, int32_t Kc, int32_t N, int32_t nterms, bool has_rowmap) {
  if (Kc % 4 != 0 || nterms > kMaxTerms) return false;
  const ProjChoice c = project_choose(M, Kc, N, nterms, true, N % 4 == 0, has_rowmap, false);
  return c.kernel == kProjX3 && c.nts <= 4;
}

// Whether the projection of this shape (aligned operands, no row map) ends in a kernel with the fused relu + pool epilogue.
static bool project_pool_fusable(int64_t M, int32_t Kc, int32_t N, int32_t nterms, int32_t pool) {
  if (pool < 2 || 16 % pool != 0 || M % pool != 0 || N % 4 != 0 || nterms > kMaxTerms) return false;
  const ProjChoice c = project_choose(M, Kc, N, nterms, Kc % 4 == 0, true, false, false);
  return c.pool_epilogue && c.pool_max % pool == 0;
}

int tgcn_cheb_project_f32(void* stream, int64_t M, int32_t Kc, int32_t N, int32_t nterms, const float* const* a,
                          const int64_t* lda, const float* W, const float* bias, int32_t bias_kind,
                          int64_t n_vertices, int64_t interleave, int32_t accumulate, float* out, int64_t ldo) {
  return project_impl(stream, M, Kc, N, nterms, a, lda, W, bias, bias_kind, n_vertices, interleave, accumulate, out, ldo, 0, 0);
}

int tgcn_cheb_project_mapped_f32(void* stream, int64_t M, int32_t Kc, int32_t N, int32_t nterms, const float* const* a, const int64_t* lda,
                                 const float* W, const float* bias, int32_t bias_kind, int64_t n_vertices, int64_t interleave, const int32_t* rowmap,
                                 uint32_t mapped_terms, int32_t nbatch, const int64_t* a_bs, int64_t out_bs, float* out, int64_t ldo) {
  if (!rowmap || nbatch < 1 || (nbatch > 1 && !a_bs) || (mapped_terms & kProjMapTermsOnly) || interleave < 1)
    TGCN_FAIL(TGCN_ERR_INVALID, "project_mapped: bad argument");
  int64_t zero_bs[kMaxTerms] = {0};
  return project_impl(stream, M, Kc, N, nterms, a, lda, W, bias, bias_kind, n_vertices, interleave, 0, out, ldo, 0, 0, -1, rowmap, mapped_terms, nbatch,
                      a_bs ? a_bs : zero_bs, out_bs);
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
                        int32_t win_n, int32_t win_t, int32_t bias_cols, const int32_t* rowmap, uint32_t mapped,
                        int32_t nbatch, const int64_t* a_bs, int64_t out_bs, int32_t pool, uint8_t* pool_idx, const ProjGather* gather) {
  if (nbatch < 1 || (nbatch > 1 && !a_bs)) TGCN_FAIL(TGCN_ERR_INVALID, "project: nbatch %d", nbatch);
  if (M <= 0 || Kc <= 0 || N <= 0 || nterms <= 0 || !a || !lda || !W || !out) TGCN_FAIL(TGCN_ERR_INVALID, "project: bad argument");
  if (int drc = check_pointer_device(out, (hipStream_t)stream, "project")) return drc;
  if (nterms > kMaxTerms) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "project: nterms %d > %d (chunk with accumulate=1)", nterms, kMaxTerms);
  if (bias_kind < 0 || bias_kind > 2 || (bias_kind && !bias)) TGCN_FAIL(TGCN_ERR_INVALID, "project: bias_kind %d", bias_kind);
  if (interleave < 1 || n_vertices < 1) TGCN_FAIL(TGCN_ERR_INVALID, "project: interleave/n_vertices");
  if (interleave > 1 && !rowmap && M != interleave * n_vertices) TGCN_FAIL(TGCN_ERR_INVALID, "project: M != interleave*n_vertices");
  if (rowmap && win_n != 0) TGCN_FAIL(TGCN_ERR_INVALID, "project: a row map excludes windows");
  if (rowmap && interleave != 1 && (M % interleave != 0 || (mapped & kProjMapTermsOnly) || nbatch != 1))
    TGCN_FAIL(TGCN_ERR_INVALID, "project: row map with interleave %lld: M must be mapped vertices x interleave, one sample batch", (long long)interleave);
  ProjParams p;
  memset(&p, 0, sizeof(p));
  p.rowmap = rowmap; p.mapped = rowmap ? mapped : 0u;
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
  bool strides4 = (out_bs % 4 == 0);
  if (nbatch > 1) for (int t = 0; t < nterms; ++t) strides4 = strides4 && (a_bs[t] % 4 == 0);
  const bool stream_ok = pool <= 1 && !gather && !accumulate && interleave == 1 && win_n == 0 && strides4 && bias_cols < 0 && n_vertices < (int64_t)INT32_MAX;
  const ProjChoice choice = project_choose(M, Kc, N, nterms, vec4, p.vec_epilogue != 0, rowmap != nullptr, win_n != 0, stream_ok, nbatch);
  if (rowmap && interleave != 1 && choice.kernel != kProjNarrow)
    TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "project: a row map together with interleave is the vector-ALU kernel's form (nterms*Kc <= %d, N %% 4 == 0, M >= 4096)", kNarrowMaxK);
  if (pool > 1) {     // fused relu + max-pool epilogue: only where the dispatch takes a kernel that has it
    if (rowmap || interleave != 1 || accumulate || win_n != 0 || nbatch != 1 || pool < 2 || choice.pool_max % pool != 0 || M % pool != 0 || !choice.pool_epilogue ||
        n_vertices % pool != 0)
      TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "project: no fused pool epilogue for this shape (M=%lld Kc=%d N=%d terms=%d pool=%d)", (long long)M, Kc, N, nterms, pool);
    p.pool = pool; p.pool_idx = pool_idx;
  }
  if (gather) {       // fused last hop: only the bf16x3 kernel with at most 4 column tiles has the gathering form
    if (choice.kernel != kProjX3 || !vec4 || choice.nts > 4 || pool > 1 || accumulate || interleave != 1 || win_n != 0 || !gather->A || !gather->X ||
        gather->term < 0 || gather->term >= nterms || gather->A->n != M || ((uintptr_t)gather->X & 15) || (gather->xbs % 4) != 0 ||
        (rowmap && ((mapped >> gather->term) & 1u)))
      TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "project: no gathering form for this shape (M=%lld Kc=%d N=%d terms=%d)", (long long)M, Kc, N, nterms);
    p.g_rowptr = gather->A->rowptr; p.g_edges = gather->A->edges; p.g_X = gather->X; p.g_xbs = gather->xbs;
    p.g_term = gather->term; p.g_thresh = gather->thresh;
  } else {
    p.g_term = -1;
  }
  p.nbatch = 1;
  if (nbatch > 1) {
    // samples sharing the tile rows: inside project_x3_kernel<NT, true>, a host loop otherwise
    if ((choice.kernel == kProjX3 && vec4) || choice.kernel == kProjX3Stream) {
      p.nbatch = nbatch; p.out_bs = out_bs;
      for (int t = 0; t < nterms; ++t) {
        p.a_bs[t] = a_bs[t];
        if (a_bs[t] % 4 != 0) TGCN_FAIL(TGCN_ERR_INVALID, "project: sample stride of term %d not a multiple of 4 floats", t);
      }
      if (out_bs % 4 != 0) p.vec_epilogue = 0;
    } else {
      const float* ab[kMaxTerms];
      for (int b = 0; b < nbatch; ++b) {
        for (int t = 0; t < nterms; ++t) ab[t] = a[t] + (int64_t)b * a_bs[t];
        const int rc = project_impl(stream, M, Kc, N, nterms, ab, lda, W, bias, bias_kind, n_vertices, interleave, accumulate,
                                    out + (int64_t)b * out_bs, ldo, win_n, win_t, bias_cols, rowmap, mapped);
        if (rc != TGCN_OK) return rc;
      }
      return TGCN_OK;
    }
  }
  const int nt = choice.nt;
  hipStream_t st = (hipStream_t)stream;
  const int kc4 = (Kc + 3) / 4 * 4;
  const size_t wbytes = choice.wbytes;
  const unsigned gy = (unsigned)((N + nt * 16 - 1) / (nt * 16));
  if (choice.kernel == kProjNarrow) {
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
  if (choice.kernel == kProjX3Stream) {
    const int snt = x3_stream_nt(N);
    const size_t lds = x3_stream_lds(Kc, N, nterms);
    const int64_t ntiles = (M + 15) / 16;
    int64_t gx = (ntiles + 15) / 16;
    if (gx > cu_count()) gx = cu_count();           // persistent: one 1024-thread workgroup per CU, waves take tiles round robin
    ProfScope ps(TGCN_PROF_PROJECT, st);
#define TGCN_PROJ_S(NTV, KTV)                                                                                              \
  {                                                                                                                        \
    allow_large_lds((const void*)project_x3_stream_kernel<NTV, KTV>, (int)kX3StreamMaxLds);                                \
    hipLaunchKernelGGL((project_x3_stream_kernel<NTV, KTV>), dim3((unsigned)gx), dim3(1024), lds, st, p, ntiles);          \
  }
    if (Kc == 32) { if (snt == 1) TGCN_PROJ_S(1, 1) else if (snt == 2) TGCN_PROJ_S(2, 1) else TGCN_PROJ_S(4, 1) }
    else { if (snt == 1) TGCN_PROJ_S(1, 2) else if (snt == 2) TGCN_PROJ_S(2, 2) else TGCN_PROJ_S(4, 2) }
#undef TGCN_PROJ_S
    TGCN_CHECK_LAUNCH("tgcn_cheb_project_f32 (bf16x3 streaming)");
    return TGCN_OK;
  }
  const bool use_x3 = choice.kernel == kProjX3 || choice.kernel == kProjX3Wide;
  if (choice.kernel == kProjResident) {
    const int rt = g_proj_variant.load() == 2 ? 1 : 2;                  // 8 waves x 32 rows (variant 2: 16 waves x 16 rows)
    const int res_rows = 16 * rt, res_waves = 1024 / rt / 64;
    const size_t lds = wbytes + (size_t)kResScratchFloats * sizeof(float);
    const int64_t ntiles = (M + res_rows - 1) / res_rows;
    int64_t gx = (ntiles + res_waves - 1) / res_waves;
    if (gx > cu_count()) gx = cu_count();    // one persistent workgroup per CU (LDS-limited residency)
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
  const int tiles = (N + 15) / 16, nts = choice.nts;
  const dim3 grid((unsigned)mb, (unsigned)((tiles + nts - 1) / nts));
  ProfScope ps(gather ? TGCN_PROF_PROJECT_GATHER : TGCN_PROF_PROJECT, st);
#define TGCN_PROJ(NTV)                                                                               \
  if (vec4) hipLaunchKernelGGL((project_kernel<NTV, true>), grid, dim3(kBlock), 0, st, p);             \
  else hipLaunchKernelGGL((project_kernel<NTV, false>), grid, dim3(kBlock), 0, st, p);
  if (use_x3) {      // bf16x3 products on the bf16 matrix pipe
    const int64_t gx3 = (M + 255) / 256 * p.nbatch;                   // sample-fastest: the nbatch workgroups of a tile are neighbours
    if (gx3 > (int64_t)INT32_MAX) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "project: grid too large");
    const dim3 grid3((unsigned)gx3, grid.y);
    // project_x3v2_kernel runs one workgroup per CU: when the last round of 256-row tiles would fill at most ~60 % of the CUs,
    // its rows go out as 128-row tiles (twice the workgroups, about half the duration each)
    const int64_t B = (M + 255) / 256, rem = B % cu_count();
    int64_t main_blocks = B, tail_blocks = 0;
    if (rem != 0 && rem * 8 <= (int64_t)cu_count() * 5 && g_x3_tail.load()) {
      main_blocks = B - rem;
      tail_blocks = (M - main_blocks * 256 + 127) / 128;
    }
    const dim3 grid3v2((unsigned)(main_blocks + tail_blocks), grid.y);
    if (gather) {
      switch (nts) {
        case 1: hipLaunchKernelGGL((project_x3_gather_kernel<1>), grid3, dim3(512), 0, st, p); break;
        case 2: hipLaunchKernelGGL((project_x3_gather_kernel<2>), grid3, dim3(512), 0, st, p); break;
        default: hipLaunchKernelGGL((project_x3_gather_kernel<4>), grid3, dim3(512), 0, st, p); break;
      }
      TGCN_CHECK_LAUNCH("tgcn_cheb_project_f32 (bf16x3 + fused last hop)");
      return TGCN_OK;
    }
#define TGCN_PROJ3(NTV)                                                                              \
  if (NTV >= 6 && choice.kernel == kProjX3Wide) hipLaunchKernelGGL((project_x3v2_kernel<NTV>), grid3v2, dim3(512), 0, st, p, (int)main_blocks); /* wide outputs: compute-bound */ \
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
  const int tgroups = (nterms + kWgTerms - 1) / kWgTerms;
  if ((N + 63) / 64 > 65535 || (int64_t)ctiles * tgroups > 65535) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "wgrad: Kc=%d N=%d too wide", Kc, N);
  hipStream_t st = (hipStream_t)stream;
  { ProfScope ps(TGCN_PROF_WGRAD, st);
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3(p.nblocks, (N + 63) / 64, ctiles * tgroups), dim3(64), 0, st, p); }
  { ProfScope ps(TGCN_PROF_WGRAD, st);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((int64_t)nterms * Kc * N + 63) / 64)), dim3(1024), 0, st, p); }
  TGCN_CHECK_LAUNCH("tgcn_cheb_wgrad_f32");
  return TGCN_OK;
}

int tgcn_relayout_qnc_to_nqc_f32(void* stream, const float* in, float* out, int64_t Q, int64_t n, int32_t C) {
  if (!in || !out || Q <= 0 || n <= 0 || C <= 0) TGCN_FAIL(TGCN_ERR_INVALID, "relayout: bad argument");
  if (C > 32) {       // wide rows: a coalesced row copy
    const bool v4 = (C % 4 == 0) && (((uintptr_t)in & 15) == 0) && (((uintptr_t)out & 15) == 0);
    const int64_t units = Q * n * (v4 ? C / 4 : C);
    ProfScope ps(TGCN_PROF_RELAYOUT, (hipStream_t)stream);
    if (v4) hipLaunchKernelGGL((relayout_rows_kernel<4>), dim3(grid_1d(units)), dim3(kBlock), 0, (hipStream_t)stream, in, out, Q, n, C);
    else hipLaunchKernelGGL((relayout_rows_kernel<1>), dim3(grid_1d(units)), dim3(kBlock), 0, (hipStream_t)stream, in, out, Q, n, C);
    TGCN_CHECK_LAUNCH("tgcn_relayout_qnc_to_nqc_f32 (wide rows)");
    return TGCN_OK;
  }
  const int64_t gy = (Q + kRelT - 1) / kRelT;
  if (gy > 65535) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "relayout: Q too large");
  const int vt = relayout_vertex_tile(C);
  const dim3 grid((unsigned)((n + vt - 1) / vt), (unsigned)gy);
  ProfScope ps(TGCN_PROF_RELAYOUT, (hipStream_t)stream);
  hipLaunchKernelGGL(relayout_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, in, out, Q, n, (int)C, vt);
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
  if (int drc = check_pointer_device(x, (hipStream_t)stream, "forward_small")) return drc;
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
      hipStream_t st = (hipStream_t)stream;
      ProfScope ps(TGCN_PROF_SMALL, st);
      const int S3 = g_small_dense.load() >= 2 ? dense_x3_config(A->n, A->nnz, C, mode, q, tiles16, false) : 0;
      p.npad = 0; p.spw = S3 ? S3 : S;
      if (S3) launch_small_dense_x3<false>(st, p, S3, tiles16);     // L . Y on the bf16 matrix pipe, three-way split
      else launch_small_dense<false>(st, p, S, tiles16);
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
      hipStream_t st = (hipStream_t)stream;
      ProfScope ps(TGCN_PROF_SMALL_BASIS, st);
      const int S3 = g_small_dense.load() >= 2 ? dense_x3_config(A->n, A->nnz, C, mode, q, tiles16, true) : 0;
      p.spw = S3 ? S3 : S;
      if (S3) launch_small_dense_x3<true>(st, p, S3, tiles16);
      else launch_small_dense<true>(st, p, S, tiles16);
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

static int forward_impl(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t mode, int32_t K,
                        int64_t q, int64_t n, int32_t C, int32_t N, const float* x, const float* W,
                        const float* bias, int32_t bias_kind, float* out, int32_t layout, int64_t q_chunk,
                        void* workspace, size_t workspace_bytes, int32_t pool, uint8_t* pool_idx);

int tgcn_cheb_forward_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t mode, int32_t K,
                          int64_t q, int64_t n, int32_t C, int32_t N, const float* x, const float* W,
                          const float* bias, int32_t bias_kind, float* out, int32_t layout, int64_t q_chunk,
                          void* workspace, size_t workspace_bytes) {
  return forward_impl(stream, A, S, mode, K, q, n, C, N, x, W, bias, bias_kind, out, layout, q_chunk, workspace, workspace_bytes, 0, nullptr);
}

// shapes whose layer forward can end in the fused relu + pool epilogue: (sample, vertex) row order, one projection call per pass
static bool forward_pool_fusable(int32_t K, int64_t q, int64_t n, int32_t C, int32_t N, int32_t layout, int64_t q_chunk, int32_t pool) {
  if (layout != 0 || K > kMaxTerms || n % pool != 0) return false;
  const int64_t qc = (q_chunk <= 0 || q_chunk > q) ? q : q_chunk;
  // every pass must take the same kind of kernel: check the full and the last (shorter) pass
  const int64_t last = q % qc ? q % qc : qc;
  return project_pool_fusable(qc * n, C, N, K, pool) && project_pool_fusable(last * n, C, N, K, pool);
}

size_t tgcn_cheb_forward_pool_workspace_bytes(const tgcn_csr_sched* S, int32_t K, int64_t q, int64_t n, int32_t C, int32_t N,
                                              int32_t layout, int64_t q_chunk, int32_t pool) {
  if (!S || K < 1 || q < 1 || n < 1 || C < 1 || N < 1 || pool < 1) return 0;
  // the base figure is 0 for K = 1 on a schedule without partial rows (no hop tensors, no scratch for partial sums): such a layer
  // still needs the scratch for its output when the shape cannot take the fused epilogue
  const size_t base = tgcn_cheb_forward_workspace_bytes(S, K, q, n, C, layout, q_chunk);
  return align_up(base, 256) + (forward_pool_fusable(K, q, n, C, N, layout, q_chunk, pool) ? 0 : align_up((size_t)q * n * N * sizeof(float), 256));
}

int tgcn_cheb_forward_pool_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t mode, int32_t K, int64_t q,
                               int64_t n, int32_t C, int32_t N, const float* x, const float* W, const float* bias, int32_t bias_kind,
                               int32_t pool, float* out, uint8_t* pool_idx, int32_t layout, int64_t q_chunk, void* workspace,
                               size_t workspace_bytes) {
  if (pool < 1 || pool > 255 || n % pool != 0 || !out) TGCN_FAIL(TGCN_ERR_INVALID, "forward_pool: pool=%d n=%lld", pool, (long long)n);
  const size_t base = align_up(tgcn_cheb_forward_workspace_bytes(S, K, q, n, C, layout, q_chunk), 256);
  const bool aligned = (((uintptr_t)out & 15) == 0) && (!bias || ((uintptr_t)bias & 15) == 0) && (!pool_idx || ((uintptr_t)pool_idx & 3) == 0);
  const bool fusable = pool > 1 && forward_pool_fusable(K, q, n, C, N, layout, q_chunk, pool);
  if (fusable && aligned)       // the (q, n, N) layer output is never written
    return forward_impl(stream, A, S, mode, K, q, n, C, N, x, W, bias, bias_kind, out, layout, q_chunk, workspace, workspace_bytes, pool, pool_idx);
  // other shapes: the layer into scratch, then the relu + pool pass
  const size_t need = base + align_up((size_t)q * n * N * sizeof(float), 256);
  if (fusable && (!workspace || workspace_bytes < need))
    // the query sized the workspace for the fused epilogue (it cannot see the pointers): say what is wrong instead of "workspace"
    TGCN_FAIL(TGCN_ERR_INVALID, "forward_pool: the fused relu + pool epilogue of this shape needs out and bias 16-byte aligned and pool_idx 4-byte aligned "
                                "(out %p, bias %p, pool_idx %p); align them or pass %zu bytes of workspace for the two-pass form", (void*)out, (const void*)bias, (void*)pool_idx, need);
  if (!workspace || workspace_bytes < need) TGCN_FAIL(TGCN_ERR_WORKSPACE, "forward_pool: workspace %zu < %zu", workspace_bytes, need);
  float* y = (float*)((char*)workspace + base);
  int rc = forward_impl(stream, A, S, mode, K, q, n, C, N, x, W, bias, bias_kind, y, layout, q_chunk, workspace, base, 0, nullptr);
  if (rc != TGCN_OK) return rc;
  return tgcn_relu_pool_f32(stream, y, out, pool_idx, q, n, N, pool);
}

static int forward_impl(void* stream, const tgcn_csr* A, const tgcn_csr_sched* S, int32_t mode, int32_t K,
                        int64_t q, int64_t n, int32_t C, int32_t N, const float* x, const float* W,
                        const float* bias, int32_t bias_kind, float* out, int32_t layout, int64_t q_chunk,
                        void* workspace, size_t workspace_bytes, int32_t pool, uint8_t* pool_idx) {
  if (!A || !S || !x || !W || !out) TGCN_FAIL(TGCN_ERR_INVALID, "forward: null operand");
  if (int drc = check_pointer_device(x, (hipStream_t)stream, "forward")) return drc;
  if (K < 1 || q < 1 || n < 1 || C < 1 || N < 1 || n != A->n) TGCN_FAIL(TGCN_ERR_INVALID, "forward: bad shape (n=%lld, L is %lld)", (long long)n, (long long)A->n);
  if (mode != 0 && mode != 1) TGCN_FAIL(TGCN_ERR_INVALID, "forward: mode %d", mode);
  if (layout != 0 && layout != 1) TGCN_FAIL(TGCN_ERR_INVALID, "forward: layout %d", layout);
  if (layout == 1 && (C > 32 || q * C > (int64_t)INT32_MAX)) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "forward: layout 1 needs C <= 32");
  // rows of a multiple of 4 floats are read with 16-byte loads; other widths take the scalar forms and may start anywhere
  if ((C % 4 == 0 && ((uintptr_t)x & 15)) || ((uintptr_t)workspace & 15)) TGCN_FAIL(TGCN_ERR_INVALID, "forward: x/workspace must be 16-byte aligned");
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
    const int pl = pool > 1 ? pool : 1;              // pooled epilogue (layout 0, K <= 32 terms): output rows shrink by the pool
    float* o0 = (layout == 1) ? out : out + q0 * (n / pl) * N;
    for (int k0 = 0; k0 < K; k0 += kMaxTerms) {
      const int nt = (K - k0 < kMaxTerms) ? K - k0 : kMaxTerms;
      for (int t = 0; t < nt; ++t) { terms[t] = hop_ptr(k0 + t); ldas[t] = C; }
      const bool last = (k0 + nt >= K);
      rc = project_impl(proj_stream, M, C, N, nt, terms, ldas, W + (size_t)k0 * C * N, last ? bias : nullptr,
                        last ? bias_kind : 0, n, layout == 1 ? q : 1, k0 > 0 ? 1 : 0, o0, N, 0, 0, -1, nullptr, 0, 1, nullptr, 0,
                        pool > 1 ? pool : 0, pool_idx ? pool_idx + q0 * (n / pl) * N : nullptr);
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

// Workspace of the compacted layer: the hop tensors -- K-1 (mode 0: terms 1..K-1) or K (mode 1: T_0 = x packed to the kept rows, then
// T_1..T_{K-1}) buffers of qc x (n_c + 1) x C floats (row n_c of every sample is the zero row that entries pointing at a left-out
// vertex gather from) -- unless the caller keeps the terms in its own memory, then the long-row scratch of one hop.
static int compact_nterm_bufs(int32_t mode, int32_t K) { return mode == 1 ? K : (K > 1 ? K - 1 : 0); }

static void cfwd_ws_layout(const tgcn_csr_sched* S, int32_t mode, int32_t K, int64_t n_c, int32_t C, int64_t qc, int keep, size_t* hop_bytes,
                           size_t* off_part, size_t* total) {
  *hop_bytes = align_up((size_t)qc * (size_t)(n_c + 1) * C * sizeof(float), 256) + 65 * 256;   // staggered like fwd_ws_layout
  *off_part = keep ? 0 : (size_t)compact_nterm_bufs(mode, K) * *hop_bytes;
  *total = *off_part + align_up(tgcn_csr_hop_workspace_bytes(S, 1, C, 1), 256);      // hops run one time step per launch
}

size_t tgcn_cheb_compact_layer_workspace_bytes(const tgcn_csr_sched* S, int32_t mode, int32_t K, int64_t q, int64_t n_c, int32_t C, int64_t q_chunk,
                                               int32_t keep_terms) {
  if (!S || K < 2 || q < 1 || n_c < 1 || C < 1 || (mode != 0 && mode != 1)) return 0;
  const int64_t qc = (keep_terms || q_chunk <= 0 || q_chunk > q) ? q : q_chunk;
  size_t a, b, total;
  cfwd_ws_layout(S, mode, K, n_c, C, qc, keep_terms != 0, &a, &b, &total);
  return total;
}

size_t tgcn_cheb_forward_compact_workspace_bytes(const tgcn_csr_sched* S, int32_t K, int64_t q, int64_t n_c, int32_t C, int64_t q_chunk) {
  return tgcn_cheb_compact_layer_workspace_bytes(S, 0, K, q, n_c, C, q_chunk, 0);
}

int tgcn_cheb_forward_compact_f32(void* stream, const tgcn_csr* A_first, const tgcn_csr* A_rest, const tgcn_csr_sched* S, int32_t K,
                                  int64_t q, int64_t n, int32_t C, int32_t N, const float* x, const float* W, const float* bias,
                                  int32_t bias_kind, float* out, const int32_t* rows, const int32_t* empty_rows, int64_t n_empty,
                                  const int32_t* compact_id, int64_t q_chunk, void* workspace, size_t workspace_bytes) {
  return tgcn_cheb_compact_layer_f32(stream, A_first, A_rest, S, 0, K, q, n, C, N, x, W, nullptr, bias, bias_kind, out, rows, empty_rows, n_empty,
                                     compact_id, q_chunk, nullptr, workspace, workspace_bytes);
}

int tgcn_cheb_compact_layer_f32(void* stream, const tgcn_csr* A_first, const tgcn_csr* A_rest, const tgcn_csr_sched* S, int32_t mode, int32_t K,
                                int64_t q, int64_t n, int32_t C, int32_t N, const float* x, const float* W, const float* W_left, const float* bias,
                                int32_t bias_kind, float* out, const int32_t* rows, const int32_t* empty_rows, int64_t n_empty,
                                const int32_t* compact_id, int64_t q_chunk, float* keep_terms, void* workspace, size_t workspace_bytes) {
  if (!A_first || !A_rest || !S || !x || !W || !out || !rows) TGCN_FAIL(TGCN_ERR_INVALID, "compact_layer: null operand");
  if (int drc = check_pointer_device(x, (hipStream_t)stream, "compact_layer")) return drc;
  const int64_t n_c = A_first->n;
  if (mode != 0 && mode != 1) TGCN_FAIL(TGCN_ERR_INVALID, "compact_layer: mode %d", mode);
  if (K < 2 || K > kMaxTerms || q < 1 || n < 1 || C < 1 || N < 1) TGCN_FAIL(TGCN_ERR_INVALID, "compact_layer: bad shape (K=%d)", K);
  if (A_rest->n != n_c || A_rest->nnz != A_first->nnz || n_c < 1 || n_empty < 0 || n_c + n_empty != n || (n_empty > 0 && !empty_rows))
    TGCN_FAIL(TGCN_ERR_INVALID, "compact_layer: %lld compact + %lld left-out rows for n=%lld", (long long)n_c, (long long)n_empty, (long long)n);
  if (mode == 1 && n_empty > 0 && !W_left) TGCN_FAIL(TGCN_ERR_INVALID, "compact_layer: mode 1 needs W_left (W_0 - W_2 + W_4 - ...) for the left-out vertices");
  if ((C % 4 == 0 && ((uintptr_t)x & 15)) || ((uintptr_t)workspace & 15) || ((uintptr_t)keep_terms & 15))
    TGCN_FAIL(TGCN_ERR_INVALID, "compact_layer: x / workspace / keep_terms must be 16-byte aligned");
  const int64_t qc = (keep_terms || q_chunk <= 0 || q_chunk > q) ? q : q_chunk;      // kept terms: every sample's hop tensors survive the call
  if (qc > 65535) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "compact_layer: %lld samples per pass", (long long)qc);
  size_t hop_bytes, off_part, total;
  cfwd_ws_layout(S, mode, K, n_c, C, qc, keep_terms != nullptr, &hop_bytes, &off_part, &total);
  if (!workspace || workspace_bytes < total) TGCN_FAIL(TGCN_ERR_WORKSPACE, "compact_layer: workspace %zu < %zu", workspace_bytes, total);
  char* ws = (char*)workspace;
  float* part = (float*)(ws + off_part);
  const size_t part_bytes = total - off_part;
  const int64_t bs_c = (n_c + 1) * (int64_t)C;            // sample stride of a compact hop tensor
  const int k_first = mode == 1 ? 0 : 1;                  // first term that lives in a compact buffer
  // term k of the pass: in the caller's buffer the terms are contiguous ([term][q][n_c + 1][C]), in the workspace staggered
  auto hop_ptr = [&](int k) -> float* {
    const int i = k - k_first;
    return keep_terms ? keep_terms + (int64_t)i * q * bs_c : (float*)(ws + (size_t)i * hop_bytes);
  };
  hipStream_t st = (hipStream_t)stream;
  // the zero row of every hop tensor (gathered from by the next hop and, with compact_id, by the projection for the left-out
  // vertices); no hop writes it, so once per call
  for (int k = k_first; k < K; ++k)
    if (hipMemset2DAsync(hop_ptr(k) + n_c * (int64_t)C, (size_t)bs_c * sizeof(float), 0, (size_t)C * sizeof(float), (size_t)qc, st) != hipSuccess)
      TGCN_FAIL(TGCN_ERR_LAUNCH, "compact_layer: memset failed");
  const float* terms[kMaxTerms];
  int64_t ldas[kMaxTerms];
  for (int k = 0; k < K; ++k) ldas[k] = C;
  int rc;
  int64_t a_bs[kMaxTerms];
  a_bs[0] = mode == 1 ? bs_c : n * (int64_t)C;
  for (int k = 1; k < K; ++k) a_bs[k] = bs_c;
  // tgcn_set_tuning("fuse_last_hop", 1): last hop fused into the projection of the compact rows, where that projection is the bf16x3 kernel
  // with <= 64 output columns (with compact_proj = 1 the projection runs over all vertices instead).  Bitwise the unfused result; OFF by
  // default: measured slower on cfg5 (docs/EXPERIMENTS.md A.4).  Mode 0 only.
  const bool one_proj = mode == 0 && compact_id && g_compact_proj.load() == 1;
  const bool fuse_last = mode == 0 && g_fuse_last.load() && !one_proj && ((uintptr_t)x & 15) == 0 &&
                         project_gather_fusable(n_c, C, N, K, true) && (N % 4 == 0) && (((uintptr_t)out & 15) == 0) &&
                         (!bias || ((uintptr_t)bias & 15) == 0);
  const bool vec_rows = (C % 4 == 0) && (((uintptr_t)x & 15) == 0);
  for (int64_t q0 = 0; q0 < q; q0 += qc) {
    const int64_t qn = (q - q0 < qc) ? (q - q0) : qc;
    const float* x0 = x + q0 * n * C;
    // hops: one launch per hop and time step (a launch's gather working set stays one (n_c, C) slab).
    // mode 0 (monomials of the folded weight): P_1 = A_first x (columns in the caller's labels), P_k = A_rest P_{k-1}.
    // mode 1 (Chebyshev): T_0 = the kept rows of x, T_1 = A_rest T_0, T_k = 2 A_rest T_{k-1} - T_{k-2}.
    // fuse_last: the rows of at most row_thresh entries of the LAST hop are gathered inside the projection (project_x3_gather_kernel); the
    // hop launch then covers the longer rows only, and the last hop tensor is neither written nor read for the others.
    for (int64_t b = 0; b < qn; ++b) {
      if (mode == 1) {
        const int64_t units = vec_rows ? n_c * (C / 4) : n_c * (int64_t)C;
        if (vec_rows) hipLaunchKernelGGL((gather_rows_i32_kernel<4>), dim3(grid_1d(units)), dim3(kBlock), 0, st, x0 + b * n * C, rows, hop_ptr(0) + b * bs_c, n_c, C, (int64_t)C);
        else hipLaunchKernelGGL((gather_rows_i32_kernel<1>), dim3(grid_1d(units)), dim3(kBlock), 0, st, x0 + b * n * C, rows, hop_ptr(0) + b * bs_c, n_c, C, (int64_t)C);
        TGCN_CHECK_LAUNCH("compact_layer (pack the kept rows of x)");
      }
      for (int k = 1; k < K; ++k) {
        tgcn_dense X = {(mode == 0 && k == 1) ? const_cast<float*>(x0) + b * n * C : hop_ptr(k - 1) + b * bs_c, 0, C};
        tgcn_dense Y = {hop_ptr(k) + b * bs_c, 0, C};
        if (mode == 1 && k >= 2) {
          tgcn_dense Zd = {hop_ptr(k - 2) + b * bs_c, 0, C};
          rc = hop_impl(stream, A_rest, S, 1, C, &X, &Zd, 2.f, -1.f, nullptr, 0.f, &Y, nullptr, part, part_bytes, 0);
        } else {
          rc = hop_impl(stream, (mode == 0 && k == 1) ? A_first : A_rest, S, 1, C, &X, nullptr, 1.f, 0.f, nullptr, 0.f, &Y, nullptr, part, part_bytes,
                        (fuse_last && k == K - 1) ? 1 : 0);
        }
        if (rc != TGCN_OK) return rc;
      }
    }
    // projection of the pass's qn time steps in one launch per row class: the per-vertex bias is read once per pass
    terms[0] = mode == 1 ? hop_ptr(0) : x0;
    for (int k = 1; k < K; ++k) terms[k] = hop_ptr(k);
    float* o = out + q0 * n * N;
    if (one_proj) {
      // ONE launch over all vertices in order: x, bias and out stream contiguously; terms 1..K-1 are read through the vertex ->
      // compact id map (empty vertices read the zero row): twice the tile work of the split form, every byte in whole DRAM pages
      uint32_t bits = kProjMapTermsOnly;
      for (int k = 1; k < K; ++k) bits |= (1u << k);
      rc = project_impl(stream, n, C, N, K, terms, ldas, W, bias, bias_kind, n, 1, 0, o, N, 0, 0, -1, compact_id, bits, (int32_t)qn, a_bs, n * (int64_t)N);
      if (rc != TGCN_OK) return rc;
      continue;
    }
    // kept vertices: all K terms (mode 0: x through the row map, hop tensors in compact rows; mode 1: every term in compact rows)
    ProjGather gat;
    gat.A = (K == 2) ? A_first : A_rest;                                // the last hop's operand and its gather source (hop K-2, or x for K = 2)
    gat.X = (K == 2) ? x0 : hop_ptr(K - 2);
    gat.xbs = (K == 2) ? n * (int64_t)C : bs_c;
    gat.term = K - 1; gat.thresh = S->row_thresh;
    rc = project_impl(stream, n_c, C, N, K, terms, ldas, W, bias, bias_kind, n, 1, 0, o, N, 0, 0, -1, rows, mode == 1 ? 0u : 1u, (int32_t)qn, a_bs,
                      n * (int64_t)N, 0, nullptr, fuse_last ? &gat : nullptr);
    if (rc != TGCN_OK) return rc;
    // the others.  mode 0: P_k = 0 for k >= 1, so out = x W'_0 + bias.  mode 1: the left-out vertices are ISOLATED (no entries, never pointed at):
    // T_k = x, 0, -x, 0, ... so out = x (W_0 - W_2 + W_4 - ...) + bias = x W_left + bias.
    if (n_empty > 0) {
      const float* xt[1] = {x0};
      const int64_t xbs[1] = {n * (int64_t)C};
      rc = project_impl(stream, n_empty, C, N, 1, xt, ldas, (mode == 1 || W_left) ? W_left : W, bias, bias_kind, n, 1, 0, o, N, 0, 0, -1, empty_rows, 1u, (int32_t)qn,
                        xbs, n * (int64_t)N);
      if (rc != TGCN_OK) return rc;
    }
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
  if (int drc = check_pointer_device(x, (hipStream_t)stream, "forward_pf")) return drc;
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

// The first step of the project-first form on its own: the vertex-sharded layer (tgcn_amd/dist.py) issues its hops itself -- each one behind
// an exchange of the previous result's cut rows -- so it needs Z without the recursion that tgcn_cheb_forward_pf_f32 runs behind it.
int tgcn_cheb_project_first_f32(void* stream, int64_t q, int64_t rows, int32_t C, int32_t K, int32_t N, const float* x, const float* Wcat,
                                const float* bias, int32_t bias_kind, const int32_t* rowmap, float* Z) {
  if (!x || !Wcat || !Z) TGCN_FAIL(TGCN_ERR_INVALID, "project_first: null operand");
  if (q < 1 || rows < 1 || C < 1 || K < 1 || N < 1 || (int64_t)K * N > (int64_t)INT32_MAX) TGCN_FAIL(TGCN_ERR_INVALID, "project_first: bad shape");
  if (int drc = check_pointer_device(x, (hipStream_t)stream, "project_first")) return drc;
  const int64_t KN = (int64_t)K * N;
  const float* a1[1] = {x};
  const int64_t lda1[1] = {C};
  if (!rowmap) return project_impl(stream, q * rows, C, (int32_t)KN, 1, a1, lda1, Wcat, bias, bias_kind, rows, 1, 0, Z, KN, 0, 0, N);
  const int64_t a_bs[1] = {rows * (int64_t)C};
  return project_impl(stream, rows, C, (int32_t)KN, 1, a1, lda1, Wcat, bias, bias_kind, rows, 1, 0, Z, KN, 0, 0, N, rowmap, 0u, (int32_t)q, a_bs, rows * KN);
}

static int windows_chunks(int64_t M) { const int64_t c = (M + 16383) / 16384; return (int)(c < 1 ? 1 : (c > 256 ? 256 : c)); }

size_t tgcn_cheb_windows_wgrad_workspace_bytes(int64_t S, int64_t n_vertices, int32_t T, int32_t H, int32_t N, int32_t K) {
  if (S < 1 || n_vertices < 1 || H < 1 || T < H || N < 1 || K < 1) return 0;
  return (size_t)windows_chunks(S * (T - H + 1) * n_vertices) * K * H * N * sizeof(float);
}

int tgcn_cheb_windows_backward_f32(void* stream, int64_t S, int64_t n_vertices, int32_t T, int32_t H, int32_t N, int32_t K,
                                   const float* stack, const float* g, const float* W, float* G, float* dW, void* workspace,
                                   size_t workspace_bytes) {
  if (S < 1 || n_vertices < 1 || H < 1 || T < H || N < 1 || K < 1 || !g) TGCN_FAIL(TGCN_ERR_INVALID, "windows_backward: bad argument");
  if ((int64_t)K * H > 65535) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "windows_backward: K*H > 65535");
  hipStream_t st = (hipStream_t)stream;
  if (G) {
    if (!W) TGCN_FAIL(TGCN_ERR_INVALID, "windows_backward: the input gradient needs W");
    hipLaunchKernelGGL(windows_dgrad_kernel, dim3(grid_1d((int64_t)K * S * n_vertices * T)), dim3(kBlock), 0, st, g, W, G, S, n_vertices, T, H, N, K);
  }
  if (dW) {
    if (!stack) TGCN_FAIL(TGCN_ERR_INVALID, "windows_backward: the weight gradient needs the hop tensors");
    const int nchunks = windows_chunks(S * (T - H + 1) * n_vertices);
    const size_t need = (size_t)nchunks * K * H * N * sizeof(float);
    if (!workspace || workspace_bytes < need) TGCN_FAIL(TGCN_ERR_WORKSPACE, "windows_backward: workspace %zu < %zu", workspace_bytes, need);
    hipLaunchKernelGGL(windows_wgrad_partial_kernel, dim3((unsigned)(K * H), (unsigned)((N + 63) / 64), (unsigned)nchunks), dim3(kBlock), 0, st, stack, g,
                       (float*)workspace, S, n_vertices, T, H, N, nchunks);
    const int64_t count = (int64_t)K * H * N;
    hipLaunchKernelGGL(windows_wgrad_reduce_kernel, dim3(grid_1d(count)), dim3(kBlock), 0, st, (const float*)workspace, dW, count, nchunks);
  }
  TGCN_CHECK_LAUNCH("tgcn_cheb_windows_backward_f32");
  return TGCN_OK;
}

int tgcn_fold_weight_f32(void* stream, int32_t K, int64_t CN, const float* fold, const float* W, float* out, int32_t transpose) {
  if (K < 1 || K > 4096 || CN < 1 || !fold || !W || !out || W == out) TGCN_FAIL(TGCN_ERR_INVALID, "fold_weight: bad argument");
  hipLaunchKernelGGL(fold_weight_kernel, dim3(grid_1d(CN)), dim3(kBlock), 0, (hipStream_t)stream, fold, W, out, (int)K, CN, (int)transpose);
  TGCN_CHECK_LAUNCH("tgcn_fold_weight_f32");
  return TGCN_OK;
}

int tgcn_weight_layout_f32(void* stream, int32_t K, int32_t C, int32_t N, const float* W, float* out, int32_t kind) {
  if (K < 1 || C < 1 || N < 1 || !W || !out || W == out || kind < 0 || kind > 2) TGCN_FAIL(TGCN_ERR_INVALID, "weight_layout: bad argument");
  hipLaunchKernelGGL(weight_layout_kernel, dim3(grid_1d((int64_t)K * C * N)), dim3(kBlock), 0, (hipStream_t)stream, W, out, (int)K, (int)C, (int)N, (int)kind);
  TGCN_CHECK_LAUNCH("tgcn_weight_layout_f32");
  return TGCN_OK;
}

int tgcn_csr_hop_f64(void* stream, int64_t n, const int32_t* rowptr, const int32_t* col, const double* val, int64_t F,
                     const double* X, const double* Z, double alpha, double beta, double* Y, double* P) {
  if (n <= 0 || F <= 0 || !rowptr || !X || (!Y && !P)) TGCN_FAIL(TGCN_ERR_INVALID, "hop_f64: bad argument");
  const int64_t gx = (F + 63) / 64, gy = (n + 3) / 4;
  if (gx > (int64_t)INT32_MAX || gy > 65535 * 1024LL) TGCN_FAIL(TGCN_ERR_UNSUPPORTED, "hop_f64: grid too large");
  for (int64_t y0 = 0; y0 < gy; y0 += 65535) {       // grid.y limit: slices of 65535 row tiles
    const int64_t ny = gy - y0 < 65535 ? gy - y0 : 65535;
    const int64_t r0 = y0 * 4, rows = (n - r0 < ny * 4) ? n - r0 : ny * 4;
    hipLaunchKernelGGL(hop_f64_kernel, dim3((unsigned)gx, (unsigned)ny), dim3(kBlock), 0, (hipStream_t)stream, rows, rowptr + r0, col, val, F, X,
                       Z ? Z + r0 * F : nullptr, alpha, beta, Y ? Y + r0 * F : nullptr, P ? P + r0 * F : nullptr);
  }
  TGCN_CHECK_LAUNCH("tgcn_csr_hop_f64");
  return TGCN_OK;
}

int tgcn_csr_sddmm_f32(void* stream, const tgcn_csr* A, int64_t n_cols, int32_t nb, int32_t C, const tgcn_dense* rows, const tgcn_dense* cols,
                       float alpha, float* dval, int32_t accumulate) {
  if (!A || !rows || !cols || !rows->ptr || !cols->ptr || !dval || nb < 1 || C < 1 || n_cols < 1 || !A->rowptr || (A->nnz > 0 && !A->edges))
    TGCN_FAIL(TGCN_ERR_INVALID, "sddmm: bad argument");
  if (A->nnz == 0) return TGCN_OK;
  const bool v4 = (C % 4 == 0) && aligned4(rows) && aligned4(cols);
  const unsigned grid = grid_1d(A->nnz * 16);
  if (v4) hipLaunchKernelGGL((sddmm_kernel<4>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, A->n, A->nnz, A->rowptr, A->edges, nb, C, rows->ptr,
                             rows->batch_stride, rows->row_stride, cols->ptr, cols->batch_stride, cols->row_stride, alpha, dval, (int)accumulate);
  else hipLaunchKernelGGL((sddmm_kernel<1>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, A->n, A->nnz, A->rowptr, A->edges, nb, C, rows->ptr,
                          rows->batch_stride, rows->row_stride, cols->ptr, cols->batch_stride, cols->row_stride, alpha, dval, (int)accumulate);
  TGCN_CHECK_LAUNCH("tgcn_csr_sddmm_f32");
  return TGCN_OK;
}

int tgcn_pack_rows_f32(void* stream, const float* src, int64_t ld_src, const int64_t* idx, int64_t nrows, int32_t C, float* out) {
  if (!src || !idx || !out || nrows < 0 || C <= 0 || ld_src < C) TGCN_FAIL(TGCN_ERR_INVALID, "pack_rows: bad argument");
  if (nrows == 0) return TGCN_OK;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(grid_1d(nrows * C)), dim3(kBlock), 0, (hipStream_t)stream, src, idx, out, nrows, C, ld_src);
  TGCN_CHECK_LAUNCH("tgcn_pack_rows_f32");
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
