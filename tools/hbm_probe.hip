// HBM streaming probe (MI355X): which launch shape of a plain 16-byte-per-lane copy / read sweep gets closest to the
// memory rate -- the numbers behind csrc/yardstick.hip's choices.   hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool CHUNK>
__global__ __launch_bounds__(256) void copy_k(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long n16) {
  long long i, stride, end;
  if (CHUNK) {                       // every workgroup owns one contiguous chunk
    const long long per = (n16 + gridDim.x - 1) / gridDim.x;
    i = blockIdx.x * per + threadIdx.x; stride = 256; end = min(n16, (long long)(blockIdx.x + 1) * per);
  } else {
    i = (long long)blockIdx.x * 256 + threadIdx.x; stride = (long long)gridDim.x * 256; end = n16;
  }
  for (; i + (U - 1) * stride < end; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
  }
  for (; i < end; i += stride) dst[i] = src[i];
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_k(const f32x4* __restrict__ src, long long n16, float* __restrict__ out) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < n16; i += stride) acc += src[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[blockIdx.x] = 1.f;
}

template <typename F>
float time_ms(F f, int reps = 6) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) f();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const long long sizes[] = {256ll << 20, 1ll << 30, 2007040000ll};
  for (long long nbytes : sizes) {
    nbytes &= ~15ll;
    float *src, *dst, *out;
    hipMalloc(&src, nbytes); hipMalloc(&dst, nbytes); hipMalloc(&out, 1 << 20);
    hipMemset(src, 1, nbytes); hipMemset(dst, 0, nbytes);
    const long long n16 = nbytes / 16;
    printf("---- %lld MB per buffer\n", nbytes >> 20);
    float ms = time_ms([&] { hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpyAsync D2D                      %7.3f ms  %6.0f GB/s\n", ms, 2.0 * nbytes / ms / 1e6);
    for (int bpc : {2, 4, 8, 16, 32}) {
      const int grid = 256 * bpc;
#define RUNC(U, NT, CH, name)                                                                                      \
      ms = time_ms([&] { hipLaunchKernelGGL((copy_k<U, NT, CH>), dim3(grid), dim3(256), 0, 0, (const f32x4*)src, (f32x4*)dst, n16); }); \
      printf("copy %-22s wg/CU %2d  %7.3f ms  %6.0f GB/s\n", name, bpc, ms, 2.0 * nbytes / ms / 1e6);
      RUNC(1, false, false, "U1 stride")
      RUNC(4, false, false, "U4 stride")
      RUNC(8, false, false, "U8 stride")
      RUNC(4, true, false, "U4 stride nt")
      RUNC(4, false, true, "U4 chunk")
      RUNC(4, true, true, "U4 chunk nt")
#define RUNR(U, NT, name)                                                                                          \
      ms = time_ms([&] { hipLaunchKernelGGL((read_k<U, NT>), dim3(grid), dim3(256), 0, 0, (const f32x4*)src, n16, out); });           \
      printf("read %-22s wg/CU %2d  %7.3f ms  %6.0f GB/s\n", name, bpc, ms, 1.0 * nbytes / ms / 1e6);
      RUNR(4, false, "U4")
      RUNR(8, false, "U8")
      RUNR(4, true, "U4 nt")
      RUNR(8, true, "U8 nt")
    }
    hipFree(src); hipFree(dst); hipFree(out);
  }
  return 0;
}
