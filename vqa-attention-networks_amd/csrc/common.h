// Internal helpers shared by the .hip translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vqa_fusion.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
  KID_HIE_AFF,
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
void vqf_splitk_counters_clear(int* words, int tiles, hipStream_t s);   // after a launch that did not return VQF_OK
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
// gemm_f32_n80.hip: one-round 128x80 tiles for the M = 512 forward projections (ta == tb == 0); returns 0 when it does not apply
int vqf_gemm_f32_n80_try(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         const float* bias, int flags, hipStream_t s, int* rc);
// gemm_f32_n80.hip: the fused LSTM step (product + cell) on the same kernel, four gate tiles per workgroup; 0 = not its shape
int vqf_lstm_step16_try(const float* h_prev, const float* w_hh, float* gates, const float* c_prev, int B, int H, float* c_out,
                        float* h_out, hipStream_t s, int* rc);
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

// tanh for the streaming kernels that were VALU-bound on libm's tanhf (hie.hip's fused tanh + Philox pass: 2 TB/s): one v_exp_f32 and
// one v_rcp_f32, tanh(x) = (e - 1) / (e + 1) with e = exp(2x); e - 1 is exact for e in [1/2, 2], so the absolute error stays at the
// exponential's (~1e-7) down to x = 0 (libm's tanhf: 1 ulp relative); e -> 0 gives -1, a huge e is cut off at +1 before inf * 0.
// NOT used where a chain of steps feeds on the value (LSTM cells, the embedding's tanh keep tanhf).
__device__ __forceinline__ float vqf_tanh_fast(float x) {
  const float e = __expf(2.0f * x);
  const float r = __builtin_amdgcn_rcpf(e + 1.0f);
  return e > 16777216.0f ? 1.0f : (e - 1.0f) * r;      // (inf - 1) * 0 would be NaN
}

// ---- device helpers -------------------------------------------------------
// Loads / stores of ONCE-touched streams (the projection P and its gradient in the fusion kernels, the image grid in the glimpse
// pools, the hidden layer in the attention-logit kernels): VQF_STREAM_NT = 1 issues them non-temporal, so that the streamed
// bytes do not displace what the caches hold (tools/hbm_probe.hip: a 2 GB read sweep 6.8 TB/s nt vs 6.1 default policy).
#ifndef VQF_STREAM_NT
#define VQF_STREAM_NT 1
#endif
template <typename T> __device__ __forceinline__ T vqf_ld_stream(const T* p) {
  return VQF_STREAM_NT ? __builtin_nontemporal_load(p) : *p;
}
template <typename T> __device__ __forceinline__ void vqf_st_stream(T* p, T v) {
  if (VQF_STREAM_NT) __builtin_nontemporal_store(v, p); else *p = v;
}

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

// Split-K combined inside the GEMM launch (cdna_hip_programming.md, "In-launch split-K reduction"): every K-slice workgroup
// has written its partial tile to its slab with plain stores; it drains them, the workgroup meets at a barrier, ONE lane
// releases at agent scope (the slabs of a tile may come from different XCDs, whose L2s are not coherent) and draws a ticket
// from the tile's counter.  The LAST arriver acquires and sums the tile's slabs z = 0 .. splits-1 IN THAT ORDER -- whoever
// arrived when -- then bias / accumulate / ReLU exactly as splitk_reduce_kernel (gemm_f32.hip) does: the result has the bits of
// the two-launch form.  Nobody waits for anybody (no spin: every wave reaches its exit); the last arriver leaves the counter at
// zero.  slab: [splits][M][N] fp32.  Returns true in the workgroup that wrote the tile.  `scratch`: 4 bytes of the kernel's ONE
// LDS array (free at this point).  TILE_N / 4 must divide NT.
// pstride != 0: TILE-LOCAL slabs (the stream-K tail of gemm_f32_big.hip): part z of this tile at slab + z * pstride, a
// TILE_M x TILE_N image with row pitch sld -- `slab` then points at the tile's own parts; 0: [splits][M][N] as above.
struct VqfSplitkTile { int* cnt; const float* slab; float* C; const float* bias; int M, N, ldc, flags; long long pstride; int sld; };
template <int TILE_M, int TILE_N, int NT, bool WT = false>      // WT: the slabs were stored write-through (sc1): no release fence
__device__ __forceinline__ bool vqf_splitk_combine(const VqfSplitkTile& g, int tile, int splits, int m0, int n0, int tid,
                                                   float* scratch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    if (!WT) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int ticket = __hip_atomic_fetch_add(g.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *reinterpret_cast<volatile int*>(scratch) = ticket;
  }
  __syncthreads();
  const int ticket = *reinterpret_cast<volatile int*>(scratch);
  __syncthreads();                                          // (scratch may be reused by the caller's next tile)
  if (ticket != splits - 1) return false;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(g.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
  }
  __syncthreads();
  constexpr int CT = TILE_N / 4, RSTEP = NT / CT;
  static_assert(NT % CT == 0, "TILE_N / 4 must divide the workgroup size");
  const bool local = g.pstride != 0;
  const long long total = local ? g.pstride : (long long)g.M * g.N;       // distance between the parts of a tile
  const int sld = local ? g.sld : g.N;                                     // row pitch of a part
  const int srow0 = local ? m0 : 0, scol0 = local ? n0 : 0;
  const bool relu = g.flags & VQF_GEMM_RELU, accum = g.flags & VQF_GEMM_ACCUM;
  const int c4 = tid % CT, r0 = tid / CT;
  const int col = n0 + 4 * c4;
  if ((g.N & 3) == 0 && (sld & 3) == 0 && (g.ldc & 3) == 0 && aligned16_dev(g.C) && aligned16_dev(g.slab) &&
      (!g.bias || aligned16_dev(g.bias))) {
    if (col >= g.N) return true;
    const f32x4 bv = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < TILE_M / RSTEP; k += 2) {
      const int rowA = m0 + r0 + RSTEP * k, rowB = rowA + RSTEP;
      if (rowA >= g.M) break;
      const bool hasB = rowB < g.M && k + 1 < TILE_M / RSTEP;
      const float* pA = g.slab + (long long)(rowA - srow0) * sld + (col - scol0);
      const float* pB = g.slab + (long long)((hasB ? rowB : rowA) - srow0) * sld + (col - scol0);
      f32x4 vA = {0.f, 0.f, 0.f, 0.f}, vB = vA;
      int z = 0;
      for (; z + 3 < splits; z += 4) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a[q] = *reinterpret_cast<const f32x4*>(pA + (z + q) * total);
          b[q] = *reinterpret_cast<const f32x4*>(pB + (z + q) * total);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { vA += a[q]; vB += b[q]; }
      }
      for (; z < splits; ++z) {
        vA += *reinterpret_cast<const f32x4*>(pA + z * total);
        vB += *reinterpret_cast<const f32x4*>(pB + z * total);
      }
      vA += bv; vB += bv;
      float* cA = g.C + (long long)rowA * g.ldc + col;
      float* cB = g.C + (long long)rowB * g.ldc + col;
      if (accum) {
        vA += *reinterpret_cast<const f32x4*>(cA);
        if (hasB) vB += *reinterpret_cast<const f32x4*>(cB);
      }
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { vA[j] = fmaxf(vA[j], 0.f); vB[j] = fmaxf(vB[j], 0.f); }
      }
      *reinterpret_cast<f32x4*>(cA) = vA;
      if (hasB) *reinterpret_cast<f32x4*>(cB) = vB;
    }
    return true;
  }
  for (int k = 0; k < TILE_M / RSTEP; ++k) {               // unaligned shapes: element by element
    const int row = m0 + r0 + RSTEP * k;
    if (row >= g.M) break;
    for (int j = 0; j < 4; ++j) {
      if (col + j >= g.N) break;
      const long long i = (long long)(row - srow0) * sld + (col + j - scol0);
      float v = 0.f;
      for (int z = 0; z < splits; ++z) v += g.slab[z * total + i];
      if (g.bias) v += g.bias[col + j];
      float* pc = g.C + (long long)row * g.ldc + col + j;
      if (accum) v += *pc;
      if (relu) v = fmaxf(v, 0.f);
      *pc = v;
    }
  }
  return true;
}

// Philox4x32-10 (Salmon et al.), counter = (ctr_lo, ctr_hi, 0, 0), key = seed.
__device__ __forceinline__ uint4 philox4x32_10(uint64_t ctr, uint64_t seed) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
#ifdef VQF_PHILOX_MULHI                               // (rounds 1-4: v_mul_hi_u32 + v_mul_lo_u32 per product, both quarter rate)
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
#else
    // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32): the same bits, half the quarter-rate multiplies
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#endif
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
