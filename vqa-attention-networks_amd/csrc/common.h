// Internal helpers shared by the .hip translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vqa_fusion.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// kernel ids for the profiler (keep in sync with prof.hip names[])
enum VqfKernelId {
  KID_GEMM_A0B0 = 0, // ta=0,tb=0   (forward projections); id = KID_GEMM_A0B0 + 2*ta + tb
  KID_GEMM_A0B1,     // ta=0,tb=1   (dgrad)
  KID_GEMM_A1B0,     // ta=1,tb=0
  KID_GEMM_A1B1,     // ta=1,tb=1   (wgrad)
  KID_SPLITK_REDUCE,
  KID_COLSUM,
  KID_GROUP_REDUCE,
  KID_RELU_BWD,
  KID_ATT_LOGITS_FWD,
  KID_ATT_LOGITS_BWD,
  KID_GLIMPSE_FWD,
  KID_GLIMPSE_BWD,
  KID_MFB_FUSE_FWD,
  KID_L2_GROUP_NORM,
  KID_SCALE_ROWS,
  KID_ROWDOT,
  KID_L2_BWD_COEF,
  KID_MFB_FUSE_BWD,
  KID_DROPOUT,
  KID_TANH_DROP_FWD,
  KID_TANH_DROP_BWD,
  KID_SOFTMAX_FWD,
  KID_SOFTMAX_BWD,
  KID_GEMM_BF16,
  KID_CAST_BF16,
  KID_LSTM_FWD,
  KID_LSTM_BWD,
  KID_CE_LOSS,
  KID_KLDIV_LOSS,
  KID_ADAM,
  KID_FEAT_TRANSPOSE,
  KID_LSTM_CELL_FWD,
  KID_LSTM_CELL_BWD,
  KID_EMBED_FWD,
  KID_EMBED_BWD,
  KID_HBM_COPY,
  KID_HBM_READ,
  KID_MULTI_ADD,
  KID_MULTI_COPY,
  KID_HIE_FWD,
  KID_HIE_HEAD,
  KID_HIE_ADD,
  KID_HIE_LEFT,
  KID_HIE_SLABSUM,
  KID_COUNT
};

extern int g_vqf_prof_on;
// library options (include/vqa_fusion.h, VQF_OPT_*): -1 = default; set from the environment once at load, then only by
// vqf_set_option.  The launchers read them as plain loads.
extern int g_vqf_opt[VQF_OPT_COUNT];
static inline int vqf_opt(int id, int dflt) { const int v = g_vqf_opt[id]; return v < 0 ? dflt : v; }
// launch counters per GEMM kernel family (VQF_STAT_*, read by vqf_stat_get)
extern long long g_vqf_stat[VQF_STAT_COUNT];
static inline void vqf_stat_bump(int id) { __atomic_fetch_add(&g_vqf_stat[id], 1LL, __ATOMIC_RELAXED); }
bool vqf_prof_begin(int id, hipStream_t s);   // false: this launch is filtered out (vqf_prof_filter), no events recorded
void vqf_prof_end(int id, hipStream_t s);
void vqf_prof_dims(int d0, int d1, int d2);   // shape tag attached to the next launches of this thread

// Launch with optional event bracketing; evaluates to the launch error code.
#define VQF_LAUNCH(id, kern, grid, block, shmem, stream, ...)                      \
  do {                                                                             \
    (void)hipGetLastError(); /* drop stale errors of unrelated earlier calls */    \
    const bool vqf_bracket_ = g_vqf_prof_on && vqf_prof_begin((id), (stream));     \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);             \
    if (vqf_bracket_) vqf_prof_end((id), (stream));                                \
  } while (0)

// two-stage column reduction (reduce.hip); scratch holds VQF_REDUCE_SPLITS x W floats
#define VQF_REDUCE_SPLITS 32
int vqf_colreduce_2stage(const float* in, int J, int W, float* out, float* scratch, hipStream_t s);

int vqf_splitk_reduce(const float* slab, int splits, int M, int N, float* C, int ldc,
                      const float* bias, int flags, hipStream_t s);   // gemm_f32.hip
int* vqf_splitk_counters(int tiles);   // gemm_f32.hip: `tiles` zero words of the in-launch split-K combine's counter ring, or nullptr

// gemm_bf16_big.hip: 256x256-tile kernel; returns 0 when it does not apply to the shape
int vqf_gemm_bf16_big_try(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C,
                          int ldc, const float* bias, int flags, const float* rowscale, int rps, void* ws, size_t ws_bytes,
                          hipStream_t s, int* rc);
size_t vqf_gemm_bf16_big_ws_bytes(int ta, int tb, int M, int N, int K);
// gemm_f32_big.hip: the same structure for fp32 operands
int vqf_gemm_f32_big_try(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                         int ldc, const float* bias, int flags, const float* rowscale, int rps, void* ws, size_t ws_bytes,
                         hipStream_t s, int* rc, int* rows_done);
int vqf_gemm_f32_big_rows_impl(int ta, int tb, int M, int N, int K, int flags, size_t ws_bytes);
size_t vqf_gemm_f32_big_ws_bytes(int ta, int tb, int M, int N, int K);
// gemm_f32_wave.hip: small-M products, one 32x64 tile per wave, no split-K slabs; returns 0 when it does not apply
int vqf_gemm_f32_wave_try(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                          int ldc, const float* bias, int flags, hipStream_t s, int* rc);

static inline int vqf_last_error() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VQF_OK : (int)e;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: a single process may drive several GPUs (the
// reference wraps the model in nn.DataParallel, solver.py:34-36), so every kernel instantiation that asks for
// more dynamic LDS than the default keeps one flag PER DEVICE.  Idempotent; a race only repeats the same call.
struct VqfDynLdsFlags { bool done[64]; };
static inline int vqf_set_dyn_lds(const void* fn, int bytes, VqfDynLdsFlags& f) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && f.done[dev]) return VQF_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  if (tracked) f.done[dev] = true;
  return VQF_OK;
}

// compute units of the current device (cached per device; 0 when the query fails)
static inline int vqf_cu_count() {
  static int cached[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && cached[dev] > 0) return cached[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
  if (tracked) cached[dev] = n;
  return n;
}

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ bool aligned16_dev(const void* p) { return (((uintptr_t)p) & 15) == 0; }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Philox4x32-10 (Salmon et al.), counter = (ctr_lo, ctr_hi, 0, 0), key = seed.
__device__ __forceinline__ uint4 philox4x32_10(uint64_t ctr, uint64_t seed) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}
// keep iff uniform(0,1) >= p  <=>  u32 >= p * 2^32   (host side; passed to the kernels)
static inline uint32_t drop_threshold_host(float p) {
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
