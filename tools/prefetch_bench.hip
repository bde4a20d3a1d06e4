// prefetch_bench.hip -- developer micro-benchmark (not part of the product path), round 4.
// Question: hop_kernel's gathers are bound by the vector L1's queue of pending misses (TCP_PENDING_STALL 53 % of
// its active cycles, data returned in order) times the latency of the slowest lines in the queue.  Scalar loads reach the L2 through
// the scalar cache, i.e. NOT through that queue: can a wave warm the L2 for the rows it is about to gather with s_load_dword
// "prefetches" and so drain the vector queue at L2-hit speed?
//   table   : n rows x 64 floats (default 4,727,102 rows = 1.21 GB: the compacted hop tensor of cfg5)
//   requests: E (default 160 M) in three tiers like the R-MAT's columns: p1 from H1 rows (L2-resident in every XCD), p2 from H2 rows
//             (Infinity-Cache resident), the rest uniform over the table
//   kernel  : hop_kernel's access shape -- 16-lane groups, 16 indices per coalesced load, 8 row loads in flight per lane, one 256-byte
//             result row per 16 requests; every lane group walks a CONTIGUOUS range of result rows so that it can look D requests ahead
//   variants: PF = 0 plain; PF = 1 / 2: one / two s_load_dword per row (first line / both 128-byte lines), D = look-ahead in requests
// Usage: prefetch_bench [n] [E] [H1] [H2] [p1] [p2]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_table(float* x, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * (1.0f / 65536.0f) - 0.5f;
}

__device__ __forceinline__ uint32_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}

__global__ void fill_indices(int32_t* idx, size_t E, uint32_t n, uint32_t H1, uint32_t H2, uint32_t t1, uint32_t t2) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < E; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t a = mix(2 * i) & 0xffff, b = mix(2 * i + 1);
    // tier rows are spread over the table (stride n/H) so that they do not share DRAM pages
    idx[i] = a < t1 ? (int32_t)((uint64_t)(b % H1) * (n / H1)) : a < t2 ? (int32_t)((uint64_t)(b % H2) * (n / H2) + 1) : (int32_t)(b % n);
  }
}

__device__ __forceinline__ int bcast16(int v, int lane) { return __shfl(v, lane, 16); }

// One dword of 32 look-ahead rows (lanes LANES of `off`, byte offsets) through the scalar cache: each line (TWO: and its neighbour at
// +128) is in the XCD's L2 afterwards.  s62 / s63 are scratch registers far above what the compiler allocates for this kernel (checked
// in the disassembly: they appear in these blocks only); the loaded dwords are never read.
#define PF_LANES_LO "0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31"
#define PF_LANES_HI "32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63"
#define PF_BLOCK(LANES, SECOND)                                              \
  asm volatile(".irp l," LANES "\n"                                          \
               "v_readlane_b32 s62, %1, \\l\n"                               \
               "s_nop 0\n"                                                    \
               "s_load_dword s63, %0, s62\n" SECOND ".endr\n"                 \
               : : "s"(base), "v"(off) : "s62", "s63", "scc", "memory")
#define PF_SECOND "s_add_u32 s62, s62, 0x80\ns_load_dword s63, %0, s62\n"
template <int PF, int HALF>
__device__ __forceinline__ void scalar_touch32(const float* base, uint32_t off) {
  if constexpr (PF == 1 && HALF == 0) PF_BLOCK(PF_LANES_LO, "");
  if constexpr (PF == 1 && HALF == 1) PF_BLOCK(PF_LANES_HI, "");
  if constexpr (PF == 2 && HALF == 0) PF_BLOCK(PF_LANES_LO, PF_SECOND);
  if constexpr (PF == 2 && HALF == 1) PF_BLOCK(PF_LANES_HI, PF_SECOND);
}

template <int PF>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ X, const int32_t* __restrict__ idx, float* __restrict__ Y,
                                                     int64_t rows_per_group, int64_t nout, int ahead_rows) {
  const int t = threadIdx.x & 15;
  const int64_t g = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t r0 = g * rows_per_group, r1 = min(nout, r0 + rows_per_group);
  for (int64_t r = r0; r < r1; ++r) {
    const int my = idx[r * 16 + t];
    uint32_t off = 0;
    if (PF > 0) {
      // the requests `ahead_rows` result rows further along this group's own stream (past its end: the next group's, harmless)
      const int64_t rp = min(nout - 1, r + ahead_rows);
      off = (uint32_t)idx[rp * 16 + t] << 8;                          // byte offset of the row: the table is < 4 GB
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(X + (int64_t)bcast16(my, j0 + u) * 64 + t * 4);
      // behind the vector loads of this half (they are in flight): half of the wave's 64 look-ahead rows
      if (j0 == 0) scalar_touch32<PF, 0>(X, off); else scalar_touch32<PF, 1>(X, off);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    *reinterpret_cast<float4*>(Y + r * 64 + t * 4) = acc;
  }
  if (PF > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 4727102u;
  const size_t E = argc > 2 ? (size_t)atoll(argv[2]) : 160000000ull;
  const uint32_t H1 = argc > 3 ? (uint32_t)atol(argv[3]) : 4096u;
  const uint32_t H2 = argc > 4 ? (uint32_t)atol(argv[4]) : 400000u;
  const double p1 = argc > 5 ? atof(argv[5]) : 0.42, p2 = argc > 6 ? atof(argv[6]) : 0.42;
  if ((uint64_t)n * 256 >= (1ull << 32)) { fprintf(stderr, "table must stay below 4 GB (32-bit scalar offsets)\n"); return 1; }
  const int reps = 4;
  float *X, *Y;
  int32_t* idx;
  const int64_t nout = (int64_t)(E / 16);
  CK(hipMalloc(&X, (size_t)n * 256));
  CK(hipMalloc(&Y, (size_t)nout * 256));
  CK(hipMalloc(&idx, E * 4));
  hipLaunchKernelGGL(fill_table, dim3(4096), dim3(256), 0, 0, X, (size_t)n * 64);
  hipLaunchKernelGGL(fill_indices, dim3(8192), dim3(256), 0, 0, idx, E, n, H1, H2, (uint32_t)(p1 * 65536.0 + 0.5), (uint32_t)((p1 + p2) * 65536.0 + 0.5));
  CK(hipDeviceSynchronize());
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("table %u rows (%.2f GB), %zu requests: %.0f %% from %u rows, %.0f %% from %u rows, rest uniform\n", n, n * 256e-9, E, p1 * 100, H1, p2 * 100, H2);
  const int64_t rpgs[] = {16, 64};            // result rows per lane group (x 16 requests): workgroups = nout / (16 * rpg)
  const int aheads[] = {1, 2, 4, 8};
  float* ref = nullptr;
  for (int64_t rpg : rpgs) {
    const int64_t groups = (nout + rpg - 1) / rpg;
    const unsigned blocks = (unsigned)((groups + 15) / 16);
    for (int pf = 0; pf <= 2; ++pf)
      for (int ai = 0; ai < (pf == 0 ? 1 : 4); ++ai) {
        const int ahead = aheads[ai];
        float best = 1e9f;
        for (int r = 0; r < reps + 1; ++r) {
          CK(hipEventRecord(a, 0));
          if (pf == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, X, idx, Y, rpg, nout, ahead);
          else if (pf == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, X, idx, Y, rpg, nout, ahead);
          else hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, X, idx, Y, rpg, nout, ahead);
          CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
          CK(hipGetLastError());
          float ms; CK(hipEventElapsedTime(&ms, a, b));
          if (r > 0 && ms < best) best = ms;
        }
        // the prefetches must not change a result: compare a strided sample with the plain run
        if (!ref) { ref = (float*)malloc(4096 * sizeof(float)); }
        float probe[4096];
        CK(hipMemcpy(probe, Y + (nout / 3) * 64, sizeof(probe), hipMemcpyDeviceToHost));
        bool same = true;
        if (pf == 0 && rpg == rpgs[0]) for (int i = 0; i < 4096; ++i) ref[i] = probe[i];
        else for (int i = 0; i < 4096; ++i) same = same && (ref[i] == probe[i]);
        printf("rows/group %3lld  prefetch %d  ahead %d rows (%3d requests): %.3f ms  %.2f TB/s gathered%s\n", (long long)rpg, pf, pf ? ahead : 0,
               pf ? ahead * 16 : 0, best, (double)nout * 16 * 256e-9 / best, same ? "" : "  RESULT DIFFERS");
        fflush(stdout);
      }
  }
  return 0;
}
