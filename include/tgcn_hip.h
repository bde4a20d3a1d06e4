/* tgcn_hip.h -- C ABI of libtgcn_hip.so: the Chebyshev (time-)graph convolution hot path of
 * cassianobecker/tgcn as hand-written HIP for gfx950 (MI355X).
 *
 * The reference has no native layer: its "operator API" for this path is the Python class surface of
 * tgcn/nn/gcn.py (TGCNCheb :8, TGCNCheb_H :82, GCNCheb :158, ChebConv :348, ChebTimeConv :445) and
 * gcn/graph.py::chebyshev (:241).  The entry points below are what a maintainer of the reference would
 * bind (ctypes) to replace the stock-torch ops inside those forwards; every entry
 * point names the reference lines it replaces.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller; fp32, row-major, last dim contiguous
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); NULL = default
 *   - nothing allocates, frees or synchronises; scratch comes from the caller (see *_workspace_bytes)
 *   - returns 0 on success, a negative TGCN_ERR_* code otherwise; tgcn_last_error() gives the text
 *     (thread-local).  Nothing throws across the ABI.  Launches go to the calling thread's current device.
 */
#ifndef TGCN_HIP_H
#define TGCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGCN_OK 0
#define TGCN_ERR_INVALID (-1)     /* bad argument (shape, alignment, null pointer) */
#define TGCN_ERR_LAUNCH (-2)      /* HIP launch error */
#define TGCN_ERR_WORKSPACE (-3)   /* caller workspace too small */
#define TGCN_ERR_UNSUPPORTED (-4) /* shape outside what the kernels were built for */

#define TGCN_ABI_VERSION 7

/* One stored entry of the sparse operand: 8 bytes, read with a single load. */
typedef struct tgcn_edge {
  int32_t col;
  float val;
} tgcn_edge;

/* CSR operand L-hat (n x n).  Replaces the dense (n,n) `self.L` of gcn.py:18,92,168 and the per-call
 * (edge_index, lap) pair of gcn.py:413,510. */
typedef struct tgcn_csr {
  int64_t n;
  int64_t nnz;           /* < 2^31 */
  const int32_t* rowptr; /* [n+1] */
  const tgcn_edge* edges; /* [nnz], rows in order */
  const float* dense;    /* optional (null if absent): the same operand as a row-major n x n fp32 matrix, duplicates
                            summed.  Only read for small dense operands (n <= 256 with >= n*n/4 stored entries), which
                            then run on the matrix pipe (tgcn_cheb_forward_small_f32 / tgcn_cheb_basis_small_f32). */
} tgcn_csr;

/* Work schedule of one operand for one lane-group width (built once by the host side, tgcn_amd/graph.py).
 *   rows with <= row_thresh stored entries: nnz-balanced row blocks, one workgroup each, one lane group per row;
 *   longer rows: segments of bounded length, STORED IN ORDER OF THEIR FIRST COLUMN so that the groups in flight
 *   gather from the same region of the dense operand at about the same time (hub columns then hit in L2).
 *   16-lane schedules: rows of more than row_thresh but at most 128 entries are one whole-row segment handled by ONE WAVE (nwseg);
 *   A segment that is its whole row writes the row directly (seg_slot = -1); rows cut into several segments
 *   ("long rows") sum them through numbered scratch slots that a fix-up launch folds in slot order, so results
 *   do not depend on timing.  The first nhuge long rows (most slots) get a whole workgroup each in the fix-up.
 *   seg_mode 0: one lane group per segment (segments of <= 32 entries).  seg_mode 1 (lanes_per_row < 64): one WAVE per
 *   segment of <= 32 * (64 / lanes_per_row) entries -- its lane groups take consecutive pieces and the pieces are folded
 *   inside the wave, so rows up to that length are written directly and longer rows leave one partial row per wave. */
typedef struct tgcn_csr_sched {
  int32_t lanes_per_row; /* lane-group width the schedule was balanced for: 1,2,4,...,64 */
  int32_t row_thresh;
  int32_t nblk;     /* row blocks */
  int32_t nseg;     /* segments */
  int32_t nlong;    /* rows cut into more than one segment */
  int32_t nhuge;    /* leading entries of long_row folded by a whole workgroup */
  int32_t npartial; /* scratch slots (= segments of long rows) */
  int32_t seg_mode; /* 0 lane-group segments, 1 wave segments */
  int32_t row_mix;  /* 1: row blocks dealt evenly among the segment blocks instead of all in front -- set by the builders for operands with
                       >= 1/8 structurally empty rows, whose row blocks are mostly streaming zero writes that fill the gaps of the
                       gather-bound segments (uncompacted R-MAT: 4.25 -> 3.95 ms per launch); 0 otherwise (it costs 2.5 % there) */
  int32_t nwseg;    /* leading entries of the seg_* arrays that are WHOLE rows of row_thresh < entries <= 32 * (64 / lanes_per_row) (the builders
                       use them for 16-lane schedules only), in
                       order of their first column: one wave each (its lane groups take consecutive pieces, folded inside the wave), written
                       directly -- no partial rows, no fix-up for them.  The remaining nseg - nwseg entries are the lane-group segments
                       of the longer rows.  0 for 64-lane schedules */
  const int32_t* blk_row;   /* [nblk+1] first row of each block; blk_row[nblk] == n */
  const int32_t* seg_row;   /* [nseg] */
  const int32_t* seg_e0;    /* [nseg] first entry */
  const int32_t* seg_e1;    /* [nseg] one past last entry */
  const int32_t* seg_slot;  /* [nseg] scratch slot, or -1: whole row, written directly */
  const int32_t* long_row;  /* [nlong] */
  const int32_t* long_slot; /* [nlong+1] first slot of each long row (slots of a row are consecutive, in column order) */
} tgcn_csr_sched;

/* Batched dense operand: element (b, i, c) lives at ptr[b*batch_stride + i*row_stride + c]. */
typedef struct tgcn_dense {
  float* ptr;
  int64_t batch_stride;
  int64_t row_stride;
} tgcn_dense;

const char* tgcn_last_error(void);
int tgcn_abi_version(void);

/* Operand and schedule construction inside the library (tgcn_graph_create_from_coo/csr, tgcn_graph_destroy),
 * so that a caller of the C ABI needs nothing from the Python package.  The work runs ON THE DEVICE in kernels of this library
 * (csrc/device_build.h: stable LSD radix sort, prefix sums, binary-search marks -- no host round trip of the index arrays);
 * these are the ONLY entry points that allocate device memory and synchronise (a few 8-byte read-backs that size the next
 * allocation): call them once per graph, outside the forward path.  All index / value pointers are DEVICE pointers.
 *   from_coo         entries in any order; duplicates of one (row, col) stay separate entries in their given order (their
 *                    sum is what scatter_add computes, gcn.py:308,343)
 *   from_csr         the same from row pointers (int64, rowptr[n] = nnz)
 *   from_edge_index  the operand ChebConv / ChebTimeConv build on every forward (gcn.py:398-413 == :495-510) from the
 *                    caller's (2, E) int64 edge list and optional weights: self loops removed, source-degree normalised
 *   tgcn_sched_build the row-block + column-ordered-segment schedule of tgcn_csr_sched for rows of C floats (what
 *                    tgcn_amd/graph.py::Schedule builds) */
typedef struct tgcn_graph tgcn_graph; /* opaque: owns its device arrays */
typedef struct tgcn_sched tgcn_sched; /* opaque: owns its device arrays */
int tgcn_graph_create_from_coo(int64_t n, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col, const float* val,
                               tgcn_graph** out);
int tgcn_graph_create_from_csr(int64_t n, int64_t n_cols, const int64_t* rowptr, const int32_t* col, const float* val,
                               tgcn_graph** out);
int tgcn_graph_create_from_edge_index(int64_t n, int64_t E, const int64_t* edge_index, const float* edge_weight,
                                      tgcn_graph** out);
const tgcn_csr* tgcn_graph_csr(const tgcn_graph* g);
int64_t tgcn_graph_n_cols(const tgcn_graph* g);
void tgcn_graph_destroy(tgcn_graph* g);
int tgcn_sched_build(const tgcn_graph* g, int32_t C, int aligned16, tgcn_sched** out);
/* The same schedule for a CSR the caller owns (n_cols = number of columns of the operand). */
int tgcn_sched_build_csr(const tgcn_csr* A, int64_t n_cols, int32_t C, int aligned16, tgcn_sched** out);
/* The arrays of a library-built schedule copied (device to device) into arrays the caller owns, sized from the counts of
 * tgcn_sched_get: blk_row [nblk+1], seg_row / seg_e0 / seg_e1 / seg_slot [max(nseg, 1)], long_row [max(nlong, 1)],
 * long_slot [max(nlong + 1, 2)].  Lets a host side that manages its own memory (tgcn_amd/graph.py) drop the handle afterwards. */
int tgcn_sched_copy(const tgcn_sched* s, void* stream, int32_t* blk_row, int32_t* seg_row, int32_t* seg_e0, int32_t* seg_e1,
                    int32_t* seg_slot, int32_t* long_row, int32_t* long_slot);
/* COO -> CSR into CALLER memory on `stream` (what tgcn_graph_create_from_coo does into memory of its own): rowptr [n+1] int32 and
 * packed entries [nnz], rows sorted by (row, col), duplicates in their given order.  Device kernels of this library (two stable
 * radix sorts, a prefix sum); synchronises the stream once (range check of the indices).  workspace: 256-byte aligned,
 * tgcn_csr_build_workspace_bytes(n, nnz). */
size_t tgcn_csr_build_workspace_bytes(int64_t n, int64_t nnz);
int tgcn_csr_build_f32(void* stream, int64_t n, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col, const float* val,
                       int32_t* rowptr, tgcn_edge* edges, void* workspace, size_t workspace_bytes);

/* ABI v5 -- the operand VALUES the callers compute before their layers, on caller memory (device pointers, `stream`), so that a host side
 * that owns its arrays (tgcn_amd/graph.py) has no arithmetic of its own: one builder, the library's.  Both synchronise the stream once
 * (range flag / count) and write a HOST count; workspace 256-byte aligned.
 *   tgcn_edge_normalise_f32       edge list (2, E) int64 [+ weights, nullable] -> COO of the ChebConv / ChebTimeConv operand
 *                                 (tgcn/nn/gcn.py:398-413 == :495-510): self loops removed, lap_e = -deg^-1/2[row] * w_e * deg^-1/2[col] with
 *                                 deg = UNWEIGHTED edge count per source vertex (integer atomics: order-independent), deg^-1/2 = 0 at degree 0.
 *                                 row / col / val: E slots; *kept entries are written, in the given order.  (tgcn_graph_create_from_edge_index
 *                                 is this followed by tgcn_graph_create_from_coo.)
 *   tgcn_adjacency_normalise_f32  COO of the weight matrix W (m entries, any order) -> COO of rescale_L(laplacian(W, normalized=True), lmax)
 *                                 (gcn/graph.py:117-136, 232-238): d = colsum(W) + eps -- a stable sort by column, then one wave per column in a
 *                                 fixed summation order, no float atomics --, L-hat = (2/lmax) (I - D^-1/2 W D^-1/2) - I.  row_out / col_out /
 *                                 val_out: m + n slots; *count = m, plus n diagonal entries when lmax != 2. */
size_t tgcn_edge_normalise_workspace_bytes(int64_t n, int64_t E);
int tgcn_edge_normalise_f32(void* stream, int64_t n, int64_t E, const int64_t* edge_index, const float* edge_weight, int64_t* row, int64_t* col,
                            float* val, int64_t* kept, void* workspace, size_t workspace_bytes);
size_t tgcn_adjacency_normalise_workspace_bytes(int64_t n, int64_t m);
int tgcn_adjacency_normalise_f32(void* stream, int64_t n, int64_t m, const int64_t* row, const int64_t* col, const float* weight, float lmax,
                                 int64_t* row_out, int64_t* col_out, float* val_out, int64_t* count, void* workspace, size_t workspace_bytes);

/* One level of the reference's Graclus / METIS-style coarsening (gcn/coarsening.py:119-165: a Python loop over vertices and
 * entries) as host code: HOST arrays in and out -- coarsening is one-off preprocessing of the caller's graph (tgcn_amd/
 * coarsening.py mirrors coarsen / metis / compute_perm / perm_data / perm_adjacency around it).  Entries sorted by row;
 * `order` = visiting sequence; an unmatched vertex v joins the unmatched neighbour u with the largest
 * vv * (1/weight[v] + 1/weight[u]) (first one in entry order on ties), in float / double arithmetic like the reference's
 * dtype; cluster[v] = cluster index in visiting order. */
int tgcn_graclus_match_f32(int64_t nnz, const int64_t* rr, const int64_t* cc, const float* vv, int64_t n, const int64_t* order,
                           const float* weight, int32_t* cluster);
int tgcn_graclus_match_f64(int64_t nnz, const int64_t* rr, const int64_t* cc, const double* vv, int64_t n, const int64_t* order,
                           const double* weight, int32_t* cluster);
const tgcn_csr_sched* tgcn_sched_get(const tgcn_sched* s);
void tgcn_sched_destroy(tgcn_sched* s);

/* Optional launch timing for benchmarks: between start and stop every kernel launch THE CALLING THREAD makes through this
 * library is bracketed by a hipEvent pair on its own stream.  stop() (same thread) synchronises those events and returns
 * (kind, milliseconds) per launch in launch order.  Thread-local since ABI v7: launches of other threads (DataParallel replicas,
 * autograd workers running a backward) are not recorded and never touch the record.  Not for use under hipGraph capture. */
#define TGCN_PROF_HOP 0
#define TGCN_PROF_HOP_FIXUP 1
#define TGCN_PROF_PROJECT 2
#define TGCN_PROF_RELAYOUT 3
#define TGCN_PROF_SMALL 4
#define TGCN_PROF_WGRAD 5
#define TGCN_PROF_SMALL_BASIS 6
#define TGCN_PROF_HOP_LONG 7     /* hop_kernel over the rows above the threshold only (last hop fused into the projection) */
#define TGCN_PROF_PROJECT_GATHER 8 /* projection with the last hop's short rows gathered inside (project_x3_gather_kernel) */
int tgcn_profile_start(int32_t capacity);
int tgcn_profile_stop(int32_t* kinds, float* ms, int32_t capacity, int32_t* count);

/* Developer switches for A/B runs (tools/hop_bench.py, tools/proj_bench.py; every one is checked against the oracle in
 * tests/test_fuzz_parity.py).  State of the CALLING THREAD since ABI v7 (thread_local in the library, as is the launch-timing record of
 * tgcn_profile_*): a switch changes the launches the calling thread issues afterwards and nothing else -- the threads of an
 * nn.DataParallel process (examples/pytorch_based/pytorch_hcp_tgcn.py:270-273) and the autograd engine's workers keep the defaults,
 * which are what ships; the library holds no process-global mutable state, only caches of immutable per-device facts
 * (CU count, the LDS attribute of a kernel, one helper stream per device).  Same arithmetic, another kernel; TGCN_ERR_INVALID for unknown keys.
 *   "hop_variant"     0 shipped hop kernel; 1.. alternative unroll / row-interleave shapes of hop.h
 *   "hop_xcd_remap"   1 (default): each XCD gets a contiguous range of row blocks; 0: row blocks round robin over the XCDs
 *   "hop_seg_remap"   1: each XCD gets a contiguous range of the column-ordered segment blocks; 0 (default): round robin
 *   "hop_mix"         0 (default): row blocks in front unless the schedule asks for the mix (tgcn_csr_sched.row_mix); 1: always dealt among
 *                     the segment blocks; 2: segment blocks first
 *   "hop_stream"      1 (default): outputs larger than the Infinity Cache (256 MB) take the form with non-temporal entry loads, row
 *                     stores and partial-row stores (16-lane groups); 0: plain accesses always
 *   "hop_lds_pad"     bytes of unused dynamic LDS per hop_kernel workgroup: limits the workgroups per CU to 160 KB / pad (0: none)
 *   "project_variant" 0 auto; 1 exact-fp32 streaming-W; 2 exact-fp32 W-resident with 16-row wave tiles; 3 bf16x3 always;
 *                     4 exact-fp32 auto; 5 vector-ALU narrow kernel wherever it applies
 *   "x3_form"         2 (default) bf16x3 with A fragments from registers for >= 96 output columns; 1 both operands via LDS
 *   "x3_tail"         1 (default): the rows of a thinly filled last round of the wide bf16x3 kernel go out as 128-row tiles; 0: never
 *   "fuse_last_hop"   1: compacted forward with the last hop's rows of <= 32 entries gathered inside the projection kernel (the last hop tensor is
 *                     neither written nor read for them; the hop launch covers the longer rows only); bitwise the same result, measured slower
 *                     on cfg5 (290 -> 329 ms per forward), so default 0: hop + projection
 *   "overlap"         1: projection of pass i on a side stream under the hops of pass i+1 (default 0)
 *   "small_dense"     dense small operands: 2 (default) bf16x3 on the matrix pipe when the batch fills the chip, 1 exact
 *                     fp32 MFMA only, 0 vector-ALU one-launch kernels
 *   "small_narrow"    0: C <= 4 inputs use the output-side one-launch kernel (default 1: input-side recursion) */
int tgcn_set_tuning(const char* key, int32_t value);
/* Every switch above back to its default, for the calling thread (ABI v5; thread-local since v7).  A harness that sets one restores
 * them with this call in its teardown, whatever happened in between (tests/conftest.py does after every test). */
void tgcn_reset_tuning(void);

/* Geometry the host needs to build a schedule / size scratch for a row length C (floats).
 * `aligned16` != 0 when every operand base, row stride and batch stride is a multiple of 4 floats. */
int tgcn_hop_vec_width(int32_t C, int aligned16);      /* floats per lane: 4 or 1 */
int tgcn_hop_lanes_per_row(int32_t C, int aligned16);  /* 1,2,...,64 */
int tgcn_hop_groups_per_block(int32_t C, int aligned16); /* 256 / lanes_per_row */
size_t tgcn_csr_hop_workspace_bytes(const tgcn_csr_sched* sched, int32_t nb, int32_t C, int aligned16);

/* One hop of the recursion:  S = L-hat . X ;  P = S (optional) ;  Y = alpha*S + beta*Z (Z optional).
 *   reference_power (gcn.py:72-78,147-153,230-236):  P_k = L P_{k-1};  Xt[k] = 2 P_k - Xt[k-2]
 *   chebyshev       (gcn.py:423-431,521-527):        Tx_k = 2 L Tx_{k-1} - Tx_{k-2}
 *   plain SpMM      (gcn.py:258-345 spmm*):          alpha=1, Z=NULL
 * nb batches of (n x C) rows share the CSR. */
int tgcn_csr_hop_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* sched, int32_t nb, int32_t C,
                     const tgcn_dense* X, const tgcn_dense* Z, float alpha, float beta, const tgcn_dense* Y,
                     const tgcn_dense* P, void* workspace, size_t workspace_bytes);

/* tgcn_csr_hop_f32 with a second addend:  Y = alpha*S + beta*Z + gamma*Z2  (Clenshaw step of the project-first path). */
int tgcn_csr_hop2_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* sched, int32_t nb, int32_t C,
                      const tgcn_dense* X, const tgcn_dense* Z, float alpha, float beta, const tgcn_dense* Z2, float gamma,
                      const tgcn_dense* Y, const tgcn_dense* P, void* workspace, size_t workspace_bytes);

/* The same hop in fp64 for the numpy twin gcn.graph.chebyshev(L, X, K) with a float64 operand: the reference computes in
 * L.dtype (gcn/graph.py:247, 256-265).  Plain CSR (int32 rowptr / col, fp64 val), X / Z / Y / P: rows of F contiguous doubles;
 * Y = alpha * (L X) + beta * Z (Z nullable), P = L X (nullable).  Entries are summed in stored order. */
int tgcn_csr_hop_f64(void* stream, int64_t n, const int32_t* rowptr, const int32_t* col, const double* val, int64_t F,
                     const double* X, const double* Z, double alpha, double beta, double* Y, double* P);

/* Change of basis of a (K, CN) weight between the dense-L classes' recursion Xt[k] = 2 L^k x - Xt[k-2] (gcn.py:75-78) and the
 * monomials L^j x the kernels work in:  out[j, :] = sum_k fold[k, j] W[k, :]  (transpose != 0: sum_k fold[j, k] W[k, :], the
 * adjoint, for the weight gradient).  fold: K x K device matrix (tgcn_amd/functional.py::power_fold_matrix). */
int tgcn_fold_weight_f32(void* stream, int32_t K, int64_t CN, const float* fold, const float* W, float* out, int32_t transpose);

/* Re-layouts of a (K, C, N) layer weight for the drivers (ABI v6; the host side issues no torch permute on the path):
 *   kind 0: (C, K*N), out[c, k*N + n] = W[k, c, n]  -- Wcat of tgcn_cheb_forward_pf_f32;
 *   kind 1: (K, N, C), W_k^T -- the input gradient as a layer on (L^T, g, W^T);   kind 2: (N, K*C) -- G = g [W_0^T | ... | W_{K-1}^T]. */
int tgcn_weight_layout_f32(void* stream, int32_t K, int32_t C, int32_t N, const float* W, float* out, int32_t kind);

/* Stacked-hop dense projection (gcn.py:39,113,194 einsum; :420-431 / :519-527 per-hop matmul):
 *   out[r(m), :] (+)= sum_t A_t[m, 0:Kc] . W[t*Kc:(t+1)*Kc, 0:N] + bias
 * A_t: M x Kc with row stride lda[t]; W: (nterms*Kc) x N contiguous; fp32 MFMA, fp32 accumulate.
 * Row map r(m) = (m % interleave) * n_vertices + m / interleave   (interleave = 1: identity).
 * bias_kind: 0 none | 1 per channel [N] | 2 per vertex and channel [n_vertices*N] (vertex = r % n_vertices).
 * `a` and `lda` are HOST arrays of length nterms (<= 32). */
int tgcn_cheb_project_f32(void* stream, int64_t M, int32_t Kc, int32_t N, int32_t nterms, const float* const* a,
                          const int64_t* lda, const float* W, const float* bias, int32_t bias_kind,
                          int64_t n_vertices, int64_t interleave, int32_t accumulate, float* out, int64_t ldo);

/* The same contraction through a ROW MAP (ABI v5; the building block of the compacted layers): tile row m is the caller's
 * vertex rowmap[m] -- its output row, its bias row, and its row in every term t whose bit t is set in `mapped_terms` (typically term 0 = x in
 * the caller's labels); the other terms (compact hop tensors) are read at row m.  nbatch samples share the tile rows: sample b reads term t
 * at a[t] + b * a_bs[t] floats and writes out + b * out_bs (a per-vertex bias row then reaches HBM once per pass).  a_bs: HOST array of
 * nterms strides (nullable for nbatch = 1).  M = number of mapped rows (one sample); n_vertices = rows of the bias / output per sample.
 * interleave > 1 (vertex-major operands of the layout-1 driver, nbatch = 1): tile row m = (mapped vertex m / interleave, sample m % interleave),
 * M = mapped vertices x interleave; a mapped term is read at row rowmap[v] * interleave + s, the output row is s * n_vertices + rowmap[v].
 * This form exists in the vector-ALU kernel only (nterms * Kc <= 16 scalars per row, N % 4 == 0, M >= 4096): TGCN_ERR_UNSUPPORTED otherwise. */
int tgcn_cheb_project_mapped_f32(void* stream, int64_t M, int32_t Kc, int32_t N, int32_t nterms, const float* const* a, const int64_t* lda,
                                 const float* W, const float* bias, int32_t bias_kind, int64_t n_vertices, int64_t interleave, const int32_t* rowmap,
                                 uint32_t mapped_terms, int32_t nbatch, const int64_t* a_bs, int64_t out_bs, float* out, int64_t ldo);

/* Streaming time windows (replaces materialising the T-H+1 overlapping windows of
 * load/data_hcp.py:116-154 and running TGCNCheb_H on each): series[t] are the hop tensors of ONE recording,
 * (n_vertices, T) contiguous (term t = L^t applied to the raw series); window w of vertex i is series[t][i, w:w+H].
 *   out[w, i, :] = sum_t series[t][i, w:w+H] . W[t*H:(t+1)*H, :] + bias       out: (T-H+1, n_vertices, N)
 * i.e. exactly what the layer returns for the windowed batch, with the K-1 hops done once on T columns instead
 * of (T-H+1)*H. */
int tgcn_cheb_project_windows_f32(void* stream, int64_t n_vertices, int32_t T, int32_t H, int32_t N, int32_t nterms,
                                  const float* const* series, const float* W, const float* bias, int32_t bias_kind,
                                  float* out);

/* Backward of tgcn_cheb_project_windows_f32 for S recordings at once: stack (K, S, n_vertices, T) hop tensors, g the gradient
 * of the (S*(T-H+1), n_vertices, N) output, W (K*H, N).
 *   G  (nullable, (K, S, n_vertices, T)):  G[k, s, i, t] = sum_h sum_c g[(s, t-h), i, c] W[k*H + h, c]  -- per-term input
 *      gradients, which the caller folds with the hop on L^T (Horner / Clenshaw, as for the layer)
 *   dW (nullable, (K*H, N)):  dW[k*H + h, c] = sum_{s,w,i} stack[k, s, i, w+h] g[(s, w), i, c]; fixed reduction order. */
size_t tgcn_cheb_windows_wgrad_workspace_bytes(int64_t S, int64_t n_vertices, int32_t T, int32_t H, int32_t N, int32_t K);
int tgcn_cheb_windows_backward_f32(void* stream, int64_t S, int64_t n_vertices, int32_t T, int32_t H, int32_t N, int32_t K,
                                   const float* stack, const float* g, const float* W, float* G, float* dW, void* workspace,
                                   size_t workspace_bytes);

/* Weight gradient of the projection (backward of gcn.py:39,113,194 w.r.t. weight):
 *   dW[t*Kc + c, n] = sum_m A_t[m, c] * G[m, n]
 * A_t as in tgcn_cheb_project_f32 (host arrays of nterms <= 32 pointers / strides), G: M x N with row stride ldg,
 * dW: (nterms*Kc) x N contiguous.  fp32 MFMA, two-stage reduction in fixed order (deterministic). */
size_t tgcn_cheb_wgrad_workspace_bytes(int64_t M, int32_t Kc, int32_t N, int32_t nterms);
int tgcn_cheb_wgrad_f32(void* stream, int64_t M, int32_t Kc, int32_t N, int32_t nterms, const float* const* a,
                        const int64_t* lda, const float* G, int64_t ldg, float* dW, void* workspace, size_t workspace_bytes);

/* (Q, n, C) -> (n, Q, C) re-layout so that short per-sample rows become one long row per vertex (LDS-tiled transpose for C <= 32, a
 * coalesced row copy for wider rows). */
int tgcn_relayout_qnc_to_nqc_f32(void* stream, const float* in, float* out, int64_t Q, int64_t n, int32_t C);

/* Whole layer forward: K-1 hops + projection, enqueued on `stream` (capturable in a hipGraph).
 *   mode 0 "reference_power": W must be the monomial-folded weight (see tgcn_amd/functional.py);
 *          out = sum_j (L^j x) W_j + bias          (TGCNCheb / TGCNCheb_H / GCNCheb)
 *   mode 1 "chebyshev": out = sum_k T_k(L) x W_k + bias   (ChebConv / ChebTimeConv)
 * x: (q, n, C) contiguous, C = H*f; W: (K*C) x N; out: (q, n, N) contiguous.
 * layout 0: hops run on the (q, n, C) layout as is; layout 1: x is first re-laid to (n, q*C).
 * q_chunk: samples per pass (layout 0 only; 0 = all).  Workspace: tgcn_cheb_forward_workspace_bytes. */
size_t tgcn_cheb_forward_workspace_bytes(const tgcn_csr_sched* sched, int32_t K, int64_t q, int64_t n, int32_t C,
                                         int32_t layout, int64_t q_chunk);
int tgcn_cheb_forward_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* sched, int32_t mode, int32_t K,
                          int64_t q, int64_t n, int32_t C, int32_t N, const float* x, const float* W,
                          const float* bias, int32_t bias_kind, float* out, int32_t layout, int64_t q_chunk,
                          void* workspace, size_t workspace_bytes);

/* The same layer (mode 0 only) for operands with structurally EMPTY rows (R-MAT: 5.27 M of 10 M vertices; the isolated
 * fake vertices the reference's coarsening pads its graphs with, gcn/coarsening.py:167-217).  For such a vertex i every hop
 * tensor row P_k[i], k >= 1, is zero, so out[i] = x[i] W_0 + bias[i]; the hop tensors are kept for the n_c vertices that do
 * have entries only (compact ids = rank among them):
 *   A_first  n_c rows, columns in the caller's labels (hop 1 gathers from x);  A_rest  the same rows and entry order with
 *            columns in compact ids, entries whose column is an empty vertex pointing at the zero row n_c  (hops 2..K-1);
 *   rows[n_c] / empty_rows[n_empty]  caller's label of every compact / empty row, ascending;  sched: shared by both operands;
 *   compact_id[n] (nullable)  compact id of every vertex, n_c for the empty ones: lets the projection run as ONE launch over all
 *            vertices in order (tgcn_set_tuning("compact_proj", 1)) instead of one launch per row class.
 * The projection reads x, bias and writes out through the row maps; hop tensors, hop writes and four of the five projection
 * terms shrink by n_empty / n.  Bitwise equal to tgcn_cheb_forward_f32 (layout 0) on the same operand.  K >= 2. */
size_t tgcn_cheb_forward_compact_workspace_bytes(const tgcn_csr_sched* sched, int32_t K, int64_t q, int64_t n_c, int32_t C,
                                                 int64_t q_chunk);
int tgcn_cheb_forward_compact_f32(void* stream, const tgcn_csr* A_first, const tgcn_csr* A_rest, const tgcn_csr_sched* sched,
                                  int32_t K, int64_t q, int64_t n, int32_t C, int32_t N, const float* x, const float* W,
                                  const float* bias, int32_t bias_kind, float* out, const int32_t* rows, const int32_t* empty_rows,
                                  int64_t n_empty, const int32_t* compact_id, int64_t q_chunk, void* workspace, size_t workspace_bytes);

/* The compacted layer in general form (ABI v6): BOTH recurrences, and optionally the hop tensors handed back to the caller.
 *   mode 0  as tgcn_cheb_forward_compact_f32 (which is this function with mode 0, W_left = NULL, keep_terms = NULL).
 *   mode 1  true Chebyshev recurrence (ChebConv / ChebTimeConv, tgcn/nn/gcn.py:420-432, :519-528) on compact hop tensors: the kept
 *           vertices must be CLOSED -- every vertex with entries and every vertex an entry points at -- so that a left-out vertex is
 *           isolated and T_k[i] = x[i], 0, -x[i], 0, ...: out[i] = x[i] W_left + bias with W_left = W_0 - W_2 + W_4 - ... (C x N, the
 *           caller folds it: tgcn_fold_weight_f32 with the sign column).  T_0 = the kept rows of x (packed by the driver), T_1 = A_rest T_0,
 *           T_k = 2 A_rest T_{k-1} - T_{k-2}; A_first is not used.  W: (K*C) x N in the reference basis.
 *   W_left  (nullable for mode 0: the first C rows of W, i.e. W'_0)  the C x N matrix of the left-out vertices.
 *   keep_terms (nullable)  caller memory for the hop tensors, [T][q][n_c + 1][C] floats contiguous, T = K-1 (mode 0: terms 1..K-1) or
 *           K (mode 1: terms 0..K-1); row n_c of every sample is zeroed by the driver.  They are exactly the basis the weight gradient
 *           contracts with g (the training forward keeps them instead of recomputing K-1 hops in backward); all q samples are then one
 *           pass and the workspace holds the long-row scratch only.
 * One C call per layer forward replaces the host-side pipeline of K hops + 2 projections (functional.compact_forward before round 5). */
size_t tgcn_cheb_compact_layer_workspace_bytes(const tgcn_csr_sched* sched, int32_t mode, int32_t K, int64_t q, int64_t n_c, int32_t C,
                                               int64_t q_chunk, int32_t keep_terms);
int tgcn_cheb_compact_layer_f32(void* stream, const tgcn_csr* A_first, const tgcn_csr* A_rest, const tgcn_csr_sched* sched, int32_t mode,
                                int32_t K, int64_t q, int64_t n, int32_t C, int32_t N, const float* x, const float* W, const float* W_left,
                                const float* bias, int32_t bias_kind, float* out, const int32_t* rows, const int32_t* empty_rows,
                                int64_t n_empty, const int32_t* compact_id, int64_t q_chunk, float* keep_terms, void* workspace,
                                size_t workspace_bytes);

/* "Project first" form of the same layer for wide inputs and narrow outputs (N well below C = H*f, e.g.
 * TGCNCheb_H(L, 1, 32, K, 1200)): Z = x . Wcat for all K terms in ONE projection (Wcat: C x (K*N), column block j =
 * W_j, folded for mode 0), then the recursion runs on the (q, n, N) results -- Horner  Y_j = Z_j + L Y_{j+1}  for
 * mode 0, Clenshaw for mode 1 -- so the K-1 hops move N instead of C floats per row.  Same result as
 * tgcn_cheb_forward_f32 up to fp32 re-association.  `sched` must be the schedule for rows of N floats. */
size_t tgcn_cheb_forward_pf_workspace_bytes(const tgcn_csr_sched* sched, int32_t K, int64_t q, int64_t n, int32_t N);
int tgcn_cheb_forward_pf_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* sched, int32_t mode, int32_t K, int64_t q,
                             int64_t n, int32_t C, int32_t N, const float* x, const float* Wcat, const float* bias,
                             int32_t bias_kind, float* out, void* workspace, size_t workspace_bytes);

/* The first step of that form on its own (ABI v7): Z[b, r(m), 0:K*N] = x[b, m, 0:C] . Wcat, bias added to the first N columns (Z_0).
 * x: (q, rows, C) contiguous; Z: (q, rows, K*N) contiguous; bias: [N] (kind 1) or [rows, N] (kind 2), read at the OUTPUT row.
 * rowmap (nullable, device, int32[rows]): output row r(m) = rowmap[m] -- a vertex shard keeps its rows in the order
 * [interior | boundary] and writes Z in that order while reading x in the caller's.  The caller then runs the Horner / Clenshaw
 * recursion itself with tgcn_csr_hop2_f32 on strided views of Z: the vertex-sharded layer exchanges the cut rows of every
 * intermediate result between its hops (tgcn_amd/dist.py; reference call shape examples/pytorch_based/pytorch_hcp_tgcn.py:103-104). */
int tgcn_cheb_project_first_f32(void* stream, int64_t q, int64_t rows, int32_t C, int32_t K, int32_t N, const float* x, const float* Wcat,
                                const float* bias, int32_t bias_kind, const int32_t* rowmap, float* Z);

/* Small graphs (n <= 1024, C <= 128, CSR + activations fit in 160 KB of LDS -- the reference's own MNIST / coarsened
 * graphs): the whole layer in ONE launch, recursion run on the output side in LDS (Horner for mode 0, Clenshaw for
 * mode 1).  W: (K, C, N) contiguous; `fold` (nullable, device, K x K): the reference_power -> monomial fold matrix,
 * applied while the weight is staged so the caller passes the layer's raw weight.
 * tgcn_cheb_forward_small_supported returns the channel tile (16 / 8) or 0 when the shape does not fit.
 * Operands that store at least a quarter of their entries (n <= 256, C <= 32: the 148-parcel DTI graph of load/res) run
 * the same recursion on the matrix pipe, L held as A-fragments in registers: exact fp32 (v_mfma_f32_16x16x4_f32), or -- when
 * the batch fills the chip -- the three-way bf16 split of the projection kernels (fp32-accurate, 2.7x fewer MFMA cycles);
 * tgcn_set_tuning("small_dense", 0) keeps them on the vector-ALU kernel. */
int tgcn_cheb_forward_small_supported(int64_t n, int64_t nnz, int32_t C, int32_t mode);   /* counts on A->dense for dense operands */
int tgcn_cheb_forward_small_pool_supported(int64_t n, int64_t nnz, int32_t C, int32_t mode);   /* ... with the fused relu + pool epilogue */
int tgcn_cheb_forward_small_f32(void* stream, const tgcn_csr* A, int32_t mode, int32_t K, int64_t q, int32_t C, int32_t N,
                                const float* x, const float* W, const float* fold, const float* bias, int32_t bias_kind,
                                float* out);

/* Backward of the same shapes.  The input gradient is the forward kernel itself on the transposed operand
 * (dx = sum_j (L^T)^j g W_j^T: call tgcn_cheb_forward_small_f32 with A = L^T, x = g, W = the (K, N, C) transposed
 * working-basis weight, no bias).  The weight gradient needs the basis autograd would recompute through
 * _chebyshev / _time_chebyshev (gcn.py:52-79,126-154,208-237; true recurrence gcn.py:420-432,519-528):
 * tgcn_cheb_basis_small_f32 writes terms
 * k = 1 .. K-1 of the (K, q, n, C) stack in ONE launch (mode 0: monomials L^k x, the basis of the folded weight;
 * mode 1: Chebyshev T_k x); term 0 is x itself and is not copied.  Feed the terms to tgcn_cheb_wgrad_f32.
 * _supported returns the channel tile (16 / 8 / 4) or 0 when the operand does not fit in LDS. */
int tgcn_cheb_basis_small_supported(int64_t n, int64_t nnz, int32_t C, int32_t mode);
int tgcn_cheb_basis_small_f32(void* stream, const tgcn_csr* A, int32_t mode, int32_t K, int64_t q, int32_t C,
                              const float* x, float* stack);

/* Fused epilogue of the callers' pattern  x = gcn_pool_4(F.relu(layer(x)))  (examples/pytorch_based/
 * pytorch_hcp_tgcn.py:134-141): out (q, n/pool, N) = max over `pool` consecutive vertices of
 * relu(layer output); pool_idx (nullable, uint8) receives the arg-max offset for the backward.
 * _small_pool: inside the one-launch small-graph kernel (the layer output never reaches HBM);
 * tgcn_relu_pool_f32 / _bwd: the same epilogue as its own pass for shapes the small-graph kernel does not take. */
int tgcn_cheb_forward_small_pool_f32(void* stream, const tgcn_csr* A, int32_t mode, int32_t K, int64_t q, int32_t C, int32_t N,
                                     const float* x, const float* W, const float* fold, const float* bias, int32_t bias_kind,
                                     int32_t relu, int32_t pool, float* out, uint8_t* pool_idx);
int tgcn_relu_pool_f32(void* stream, const float* x, float* out, uint8_t* idx, int64_t q, int64_t n, int32_t f, int32_t p);

/* The layer of tgcn_cheb_forward_f32 followed by relu + max over `pool` consecutive vertices for graphs that do NOT fit in LDS:
 * out (q, n/pool, N), pool_idx (nullable) as above.  On the (q, n, C) layout with up to 32 terms the epilogue runs inside the projection
 * kernel -- N <= 64 output columns: bias, relu and the max over 2 / 4 / 8 / 16 rows of the finished tile in the wave's LDS scratch;
 * N >= 96 on the bf16x3 path (ABI v7): groups of 2 / 4 rows folded in registers, a lane's four accumulators of a column being four
 * consecutive rows -- so the (q, n, N) layer output is never written; other shapes (the vertex-major layout 1 of rows shorter than 32
 * floats, groups of 8 / 16 with wide outputs) run the layer into workspace scratch followed by tgcn_relu_pool_f32: measured 0 - 1.3 % of the
 * fused call on the layout-1 shapes of the reference's HCP model at mesh size (profiles/r06_pool_epilogue_cost.jsonl), final as it is
 * (the workspace query accounts for it; it returns 0 only for invalid arguments -- a K = 1 layer on a schedule without partial rows has
 * a base figure of 0 and still gets its output scratch).  pool = 1: relu only.
 * Alignment: the fused epilogue needs `out` and `bias` 16-byte aligned and `pool_idx` 4-byte aligned.  The query decides fused vs two-pass
 * from the shape alone; a call with a fusable shape and a misaligned pointer takes the two-pass form when the workspace also holds
 * q*n*N floats of scratch behind the queried size, and fails with TGCN_ERR_INVALID (message names the pointers) otherwise. */
size_t tgcn_cheb_forward_pool_workspace_bytes(const tgcn_csr_sched* sched, int32_t K, int64_t q, int64_t n, int32_t C, int32_t N,
                                              int32_t layout, int64_t q_chunk, int32_t pool);
int tgcn_cheb_forward_pool_f32(void* stream, const tgcn_csr* A, const tgcn_csr_sched* sched, int32_t mode, int32_t K, int64_t q,
                               int64_t n, int32_t C, int32_t N, const float* x, const float* W, const float* bias, int32_t bias_kind,
                               int32_t pool, float* out, uint8_t* pool_idx, int32_t layout, int64_t q_chunk, void* workspace,
                               size_t workspace_bytes);
int tgcn_relu_pool_bwd_f32(void* stream, const float* grad_z, const float* z, const uint8_t* idx, float* grad_y, int64_t q,
                           int64_t n, int32_t f, int32_t p);

/* Gradient of a hop w.r.t. the VALUES of its sparse operand (ABI v5): the sampled dense-dense product over the stored pattern
 *   dval[e] (+)= alpha * sum_b sum_c rows[b, row(e), c] * cols[b, col(e), c]      for every stored entry e, in CSR order
 * -- for S = L X: dL/dval_e = <dL/dS[row(e)], X[col(e)]>.  What makes `edge_weight` of ChebConv / ChebTimeConv and `value` of spmm* learnable as
 * in the reference, whose gather / scale / scatter_add form is differentiable in them (tgcn/nn/gcn.py:296-308, 413, 510).  rows: (nb, A->n, C),
 * cols: (nb, n_cols, C); one lane group per entry, fixed summation order (no atomics); accumulate != 0 adds to dval. */
int tgcn_csr_sddmm_f32(void* stream, const tgcn_csr* A, int64_t n_cols, int32_t nb, int32_t C, const tgcn_dense* rows, const tgcn_dense* cols,
                       float alpha, float* dval, int32_t accumulate);

/* Vertex sharding (replaces nn.DataParallel's batch split, examples/pytorch_based/pytorch_hcp_tgcn.py:270-273):
 * out[i, 0:C] = src[idx[i], 0:C] -- the rows of a hop tensor that a neighbouring shard needs, packed into one message.
 * idx: int64 device array; src rows ld_src floats apart; out contiguous. */
int tgcn_pack_rows_f32(void* stream, const float* src, int64_t ld_src, const int64_t* idx, int64_t nrows, int32_t C, float* out);

/* gcn_pool / gcn_pool_4 (gcn.py:246-255): max over p consecutive vertices; idx (nullable) receives the
 * arg-max offset 0..p-1 for the backward. */
int tgcn_pool_max_f32(void* stream, const float* x, float* out, int32_t* idx, int64_t q, int64_t n, int32_t f,
                      int32_t p);
int tgcn_pool_max_bwd_f32(void* stream, const float* grad_out, const int32_t* idx, float* grad_in, int64_t q,
                          int64_t n, int32_t f, int32_t p);

#ifdef __cplusplus
}
#endif
#endif /* TGCN_HIP_H */
