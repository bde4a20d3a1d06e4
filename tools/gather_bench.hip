// gather_bench.hip -- developer micro-benchmark (not part of the product path): the rate at which one MI355X gathers random
// 256-byte rows in the access shape of the hop kernels, as a function of the share of requests the XCD's L2 can serve.
//   table  : n rows x 64 floats (default 10 M rows = 2.56 GB, the cfg5 hop operand of one time step)
//   indices: E requests (default 160 M); a fraction p_hot of them from a hot set of H rows (H*256 B fits every L2),
//            the rest uniform over the table (each such request is an L2 miss at this table size)
//   shape 0: hop_kernel's -- 256-thread workgroups, 16-lane groups, 16 indices per coalesced load, 4 row loads in flight
//            per lane, one 256-byte result row stored per 16 requests
//   shape 1: hop_sweep_kernel's -- 256 persistent 1024-thread workgroups (one per CU), each lane group walks one long index
//            stream and adds its running sum into a private LDS slot every 16 requests
// Prints one line per (shape, p_hot): time, requests/s, TB/s of gathered bytes, and the cfg5-equivalent algorithmic GB/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_table(float* x, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * (1.0f / 65536.0f) - 0.5f;
}

__device__ __forceinline__ uint32_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}

__global__ void fill_indices(int32_t* idx, size_t E, uint32_t n, uint32_t H, uint32_t hot_per_64k) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < E; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t a = mix(2 * i), b = mix(2 * i + 1);
    const bool hot = (a & 0xffff) < hot_per_64k;
    // hot rows are spread over the table (stride n/H) so that they do not share DRAM pages
    idx[i] = hot ? (int32_t)((uint64_t)(b % H) * (n / H)) : (int32_t)(b % n);
  }
}

__device__ __forceinline__ int bcast16(int v, int lane) { return __shfl(v, lane, 16); }

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ X, const int32_t* __restrict__ idx, float* __restrict__ Y, int64_t nout) {
  const int t = threadIdx.x & 15, g = threadIdx.x >> 4;
  for (int64_t r = (int64_t)blockIdx.x * 16 + g; r < nout; r += (int64_t)gridDim.x * 16) {
    const int my = idx[r * 16 + t];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(X + (int64_t)bcast16(my, j0 + u) * 64 + t * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    *reinterpret_cast<float4*>(Y + r * 64 + t * 4) = acc;
  }
}

__global__ __launch_bounds__(1024) void gather_sweep_kernel(const float* __restrict__ X, const int32_t* __restrict__ idx, float* __restrict__ Y, int64_t per_group) {
  extern __shared__ float lds[];                      // 64 groups x 8 slots x 64 floats = 128 KB
  const int t = threadIdx.x & 15, g = threadIdx.x >> 4;
  for (int i = threadIdx.x; i < 64 * 8 * 64; i += 1024) lds[i] = 0.f;
  __syncthreads();
  const int64_t s0 = ((int64_t)blockIdx.x * 64 + g) * per_group;
  for (int64_t e = 0; e < per_group; e += 16) {
    const int my = idx[s0 + e + t];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(X + (int64_t)bcast16(my, j0 + u) * 64 + t * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    float* slot = lds + ((g * 8 + (int)((e >> 4) & 7)) * 64 + t * 4);
    atomicAdd(slot + 0, acc.x); atomicAdd(slot + 1, acc.y); atomicAdd(slot + 2, acc.z); atomicAdd(slot + 3, acc.w);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 8 * 64; i += 1024) Y[(int64_t)blockIdx.x * 64 * 8 * 64 + i] = lds[i];
}

__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 10000000u;
  const size_t E = argc > 2 ? (size_t)atoll(argv[2]) : 160000000ull;
  const uint32_t H = argc > 3 ? (uint32_t)atol(argv[3]) : 4096u;
  const int reps = 5;
  float *X, *Y;
  int32_t* idx;
  const size_t per_group = (E / (256 * 64)) / 16 * 16, Esweep = per_group * 256 * 64;
  const int64_t nout = (int64_t)(E / 16);
  CK(hipMalloc(&X, (size_t)n * 256));
  CK(hipMalloc(&Y, (size_t)nout * 256));
  CK(hipMalloc(&idx, E * 4));
  hipLaunchKernelGGL(fill_table, dim3(4096), dim3(256), 0, 0, X, (size_t)n * 64);
  CK(hipFuncSetAttribute((const void*)gather_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  // copy bandwidth of this box (second denominator of the roofline)
  {
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(copy_kernel, dim3(8192), dim3(256), 0, 0, (const float4*)X, (float4*)Y, (size_t)n * 16);
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("copy %.2f GB read + %.2f GB written: %.3f ms = %.2f TB/s\n", n * 256e-9, n * 256e-9, best, 2.0 * n * 256e-9 / best);
  }
  const double hots[] = {0.0, 0.25, 0.43, 0.60, 0.75, 0.90, 1.0};
  for (double ph : hots) {
    hipLaunchKernelGGL(fill_indices, dim3(8192), dim3(256), 0, 0, idx, E, n, H, (uint32_t)(ph * 65536.0 + 0.5));
    CK(hipDeviceSynchronize());
    for (int shape = 0; shape < 2; ++shape) {
      float best = 1e9f;
      for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(a, 0));
        if (shape == 0) hipLaunchKernelGGL(gather_rows_kernel, dim3(65536), dim3(256), 0, 0, X, idx, Y, nout);
        else hipLaunchKernelGGL(gather_sweep_kernel, dim3(256), dim3(1024), 128 * 1024, 0, X, idx, Y, (int64_t)per_group);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r > 0 && ms < best) best = ms;
      }
      const double req = shape == 0 ? (double)nout * 16 : (double)Esweep;
      const double alg = 8.0 * req / 16 + 4.0 * n / 16 + 8.0 * n * 64;     // cfg5 accounting for the same request count
      printf("shape %d  p_hot %.2f (H=%u rows): %.3f ms  %.1f G requests/s  %.2f TB/s gathered  (cfg5-equivalent algorithmic %.0f GB/s = %.3f of 8 TB/s)\n",
             shape, ph, H, best, req / best * 1e-6, req * 256e-9 / best, alg / best * 1e-6, alg / best * 1e-6 / 8000.0);
    }
  }
  return 0;
}
