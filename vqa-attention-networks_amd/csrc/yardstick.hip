// HBM yardsticks: what a plain streaming kernel of THIS library reaches on the chip, in the units the HBM-bound stages of
// the path (fusion.hip, attention.hip; mfb.py:98-106,116-123) are priced in.  bench.py reports them beside
// `roofline_hbm_kernels`; MI355X_MICROARCH.md measures 6.29 TB/s for a float4 copy and 6.0-6.1 TB/s for read sweeps.
//   copy        dst[i] = src[i]                          16 B per lane, grid-stride
//   read sweep  out[block] = sum of the block's elements (the loads cannot be dropped; one 4-byte store per workgroup)
// nt != 0: non-temporal loads / stores (the streamed bytes do not displace what the caches hold).
// Launch shapes from tools/hbm_probe.hip on 2 GB buffers (far beyond the 256 MiB Infinity Cache; gpurun_out/r04/hbm_probe.log):
// FEW resident waves stream best -- copy: one load in flight per lane, 4 workgroups per CU 5.46 TB/s (8 per CU with four loads
// in flight: 4.4); non-temporal copy: four in flight, 2 per CU 5.2; read sweep: four in flight, 2 per CU 6.1 default policy /
// 6.8 non-temporal (32 per CU: 5.7 / 6.3).  Smaller buffers copy faster (256 MB: 6.5 TB/s), which is where the guide's 6.29 sits.
#include "common.h"

namespace {

constexpr int kThreads = 256;

template <bool NT>
__device__ __forceinline__ f32x4 ld16(const f32x4* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st16(f32x4* p, f32x4 v) {
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <bool NT, int kUnroll>
__global__ __launch_bounds__(kThreads) void hbm_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long n16) {
  const long long stride = (long long)gridDim.x * kThreads;
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  for (; i + (kUnroll - 1) * stride < n16; i += kUnroll * stride) {
    f32x4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = ld16<NT>(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) st16<NT>(dst + i + u * stride, v[u]);
  }
  for (; i < n16; i += stride) st16<NT>(dst + i, ld16<NT>(src + i));
}

template <bool NT>
__global__ __launch_bounds__(kThreads) void hbm_read_kernel(const f32x4* __restrict__ src, long long n16, float* __restrict__ out) {
  constexpr int kUnroll = 4;
  const long long stride = (long long)gridDim.x * kThreads;
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (kUnroll - 1) * stride < n16; i += kUnroll * stride) {
    f32x4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = ld16<NT>(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) acc += v[u];
  }
  for (; i < n16; i += stride) acc += ld16<NT>(src + i);
  float s = wave_sum(acc[0] + acc[1] + acc[2] + acc[3]);
  __shared__ float part[kThreads / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

int grid_for(long long n16, int wg_per_cu) {
  const int cus = vqf_cu_count() > 0 ? vqf_cu_count() : 256;
  long long blocks = (n16 + kThreads - 1) / kThreads;
  const long long cap = (long long)cus * wg_per_cu;
  return (int)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

}  // namespace

extern "C" {

int vqf_hbm_copy(const void* src, void* dst, long long nbytes, int nt, void* stream) {
  if (!src || !dst || nbytes <= 0) return VQF_E_BADARG;
  if (nbytes % 16) return VQF_E_UNSUPPORTED;
  if (!aligned16(src) || !aligned16(dst)) return VQF_E_ALIGN;
  const long long n16 = nbytes / 16;
  if (nt)
    VQF_LAUNCH(KID_HBM_COPY, (hbm_copy_kernel<true, 4>), dim3(grid_for(n16, 2)), dim3(kThreads), 0, (hipStream_t)stream,
               (const f32x4*)src, (f32x4*)dst, n16);
  else
    VQF_LAUNCH(KID_HBM_COPY, (hbm_copy_kernel<false, 1>), dim3(grid_for(n16, 4)), dim3(kThreads), 0, (hipStream_t)stream,
               (const f32x4*)src, (f32x4*)dst, n16);
  return vqf_last_error();
}

int vqf_hbm_read_sweep_blocks(long long nbytes) { return nbytes > 0 ? grid_for(nbytes / 16, 2) : 0; }

int vqf_hbm_read_sweep(const void* src, long long nbytes, int nt, float* block_sums, void* stream) {
  if (!src || !block_sums || nbytes <= 0) return VQF_E_BADARG;
  if (nbytes % 16) return VQF_E_UNSUPPORTED;
  if (!aligned16(src)) return VQF_E_ALIGN;
  const long long n16 = nbytes / 16;
  const int grid = grid_for(n16, 2);
  if (nt)
    VQF_LAUNCH(KID_HBM_READ, hbm_read_kernel<true>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, (const f32x4*)src, n16,
               block_sums);
  else
    VQF_LAUNCH(KID_HBM_READ, hbm_read_kernel<false>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, (const f32x4*)src, n16,
               block_sums);
  return vqf_last_error();
}

}  // extern "C"
