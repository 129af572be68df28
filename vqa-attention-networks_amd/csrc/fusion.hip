// MFB fusion stage (HBM-bound): broadcast product, dropout, k=5 sum-pool,
// signed square root, and the per-row partial of the per-sample L2 norm.
//
//   mfb.py:98-106  (L = 196: image projection x question projection)
//   mfb.py:128-135 (L = 1:   final block)      mhb_coAtt.py:100-108,126-145,192-211
//
// One thread owns 4 pooled outputs = 20 consecutive projection columns, i.e.
// five 16-byte loads from the (N*L, 5000) projection row; a 256-thread
// workgroup owns one row (1000 pooled outputs -> 250 active lanes), so every
// HBM access is a fully coalesced 16-B-per-lane stream.  The 5-wide pooling
// window never crosses a thread (20 = 4*5), so no cross-lane traffic is needed
// for the pool; the row's sum of squares is a wavefront-shuffle reduction.
// Dropout masks are never stored: forward and backward regenerate them from Philox4x32-10(seed, element index / 8),
// one 16-bit draw per element (keep iff draw >= p * 65536).  Round 1 spent one call per 4 elements on 32-bit draws: the
// 40 quarter-rate 32-bit multiplies of a call made these kernels VALU-bound (0.5 ms of SIMD time per pass over the
// 5e8 elements -- the bf16-P forward took as long as the fp32-P one); 16-bit draws halve that.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int KP = VQF_POOL_K;      // 5
constexpr int TPT = 4;              // pooled outputs per thread
constexpr int CPT = KP * TPT;       // 20 projection columns per thread

__device__ __forceinline__ void load20(const float* __restrict__ p, float (&v)[CPT]) {
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(p + 4 * i);
    v[4 * i] = x[0]; v[4 * i + 1] = x[1]; v[4 * i + 2] = x[2]; v[4 * i + 3] = x[3];
  }
}
// 20 bf16 (40 bytes, 8-byte aligned) -> 20 floats
__device__ __forceinline__ void load20(const __bf16* __restrict__ p, float (&v)[CPT]) {
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const uint2 x = *reinterpret_cast<const uint2*>(p + 4 * i);
    v[4 * i] = __uint_as_float(x.x << 16); v[4 * i + 1] = __uint_as_float(x.x & 0xffff0000u);
    v[4 * i + 2] = __uint_as_float(x.y << 16); v[4 * i + 3] = __uint_as_float(x.y & 0xffff0000u);
  }
}
__device__ __forceinline__ void store20(float* __restrict__ p, const float (&v)[CPT]) {
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    f32x4 x = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    *reinterpret_cast<f32x4*>(p + 4 * i) = x;
  }
}

// 20 values -> 20 bf16 (round-to-nearest-even), 5 x 8-byte stores (e0 * 2 bytes is a multiple of 8)
__device__ __forceinline__ void store20(__bf16* __restrict__ p, const float (&v)[CPT]) {
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    __bf16 h[4] = {(__bf16)v[4 * i], (__bf16)v[4 * i + 1], (__bf16)v[4 * i + 2], (__bf16)v[4 * i + 3]};
    *reinterpret_cast<uint2*>(p + 4 * i) = *reinterpret_cast<const uint2*>(h);
  }
}

// ---- coalesced access to the projection rows through a wave-private LDS transpose --------------------------------
// A thread owns 20 consecutive elements of a row, so a direct load / store instruction touches 16 (8) bytes every 80 (40)
// bytes: each 128-byte line is visited by five different instructions, and with ~100 KB of rows in flight per CU the
// 32 KB L1 no longer holds a line until its fifth visit (the kernels sat at 4.6-4.7 TB/s with the dropout arithmetic
// removed).  Here a wave's 64 threads (1280 consecutive elements) move their span as lane-linear 16-byte pieces
// -- every line is touched by exactly one instruction -- and redistribute it through 5 KB of LDS of their own:
// ds_write_b128 lane-linear, ds_read_b128 at an 80-byte lane stride (both conflict-free), no barrier (the LDS executes
// one wave's instructions in order).
// The exchange is between LANES of one wave, which the compiler's per-thread view does not see: without a fence hipcc
// proves that a thread's own stores cannot alias its reads of other lanes' slots and hoists those reads out of the row
// loop.  A wavefront-scope fence emits no instruction and pins the order.
#ifndef VQF_FUSE_ROWBAR
#define VQF_FUSE_ROWBAR 0      // 1: round 3's forward (the row's sum of squares folded over the four waves behind a barrier per row; A/B)
#endif
#define VQF_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
constexpr int WSPAN = 64 * CPT;                        // elements of a row owned by one wave
constexpr int WLDS = WSPAN * 4;                        // bytes of a wave's transpose buffer
template <typename PT> struct Raw;
template <> struct Raw<float> { f32x4 v[5]; };         // 320 pieces of 4 floats per wave span
template <> struct Raw<__bf16> { uint4 v[3]; };        // 160 pieces of 8 bf16: the third instruction only in lanes 0-31

__device__ __forceinline__ void raw_load(const float* __restrict__ rowp, int W5, int wave, int lane, Raw<float>& r) {
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int el = wave * WSPAN + (64 * k + lane) * 4;
    r.v[k] = el < W5 ? vqf_ld_stream(reinterpret_cast<const f32x4*>(rowp + el)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
__device__ __forceinline__ void raw_load(const __bf16* __restrict__ rowp, int W5, int wave, int lane, Raw<__bf16>& r) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int pc = 64 * k + lane, el = wave * WSPAN + pc * 8;
    if (pc < 160 && el < W5) {
      const u32x4 t = vqf_ld_stream(reinterpret_cast<const u32x4*>(rowp + el));
      r.v[k] = make_uint4(t[0], t[1], t[2], t[3]);
    } else {
      r.v[k] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
}
__device__ __forceinline__ void raw_to_own(const Raw<float>& r, char* wl, int lane, float (&p)[CPT]) {
  VQF_WAVE_FENCE();
#pragma unroll
  for (int k = 0; k < 5; ++k) *reinterpret_cast<f32x4*>(wl + (64 * k + lane) * 16) = r.v[k];
  VQF_WAVE_FENCE();
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(wl + lane * 80 + 16 * i);
    p[4 * i] = x[0]; p[4 * i + 1] = x[1]; p[4 * i + 2] = x[2]; p[4 * i + 3] = x[3];
  }
}
__device__ __forceinline__ void raw_to_own(const Raw<__bf16>& r, char* wl, int lane, float (&p)[CPT]) {
  VQF_WAVE_FENCE();
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (64 * k + lane < 160) *reinterpret_cast<uint4*>(wl + (64 * k + lane) * 16) = r.v[k];
  VQF_WAVE_FENCE();
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const uint2 x = *reinterpret_cast<const uint2*>(wl + lane * 40 + 8 * i);
    p[4 * i] = __uint_as_float(x.x << 16); p[4 * i + 1] = __uint_as_float(x.x & 0xffff0000u);
    p[4 * i + 2] = __uint_as_float(x.y << 16); p[4 * i + 3] = __uint_as_float(x.y & 0xffff0000u);
  }
}
// the reverse: every thread's 20 values -> the wave's span of a row, stored as lane-linear 16-byte pieces
__device__ __forceinline__ void own_store(float* __restrict__ rowp, int W5, int wave, int lane, char* wl,
                                          const float (&v)[CPT]) {
  VQF_WAVE_FENCE();
#pragma unroll
  for (int i = 0; i < 5; ++i)
    *reinterpret_cast<f32x4*>(wl + lane * 80 + 16 * i) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
  VQF_WAVE_FENCE();
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int pc = 64 * k + lane, el = wave * WSPAN + pc * 4;
    if (el < W5) vqf_st_stream(reinterpret_cast<f32x4*>(rowp + el), *reinterpret_cast<const f32x4*>(wl + pc * 16));
  }
}
__device__ __forceinline__ void own_store(__bf16* __restrict__ rowp, int W5, int wave, int lane, char* wl,
                                          const float (&v)[CPT]) {
  VQF_WAVE_FENCE();
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    __bf16 h[4] = {(__bf16)v[4 * i], (__bf16)v[4 * i + 1], (__bf16)v[4 * i + 2], (__bf16)v[4 * i + 3]};
    *reinterpret_cast<uint2*>(wl + lane * 40 + 8 * i) = *reinterpret_cast<const uint2*>(h);
  }
  VQF_WAVE_FENCE();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int pc = 64 * k + lane, el = wave * WSPAN + pc * 8;
    if (pc < 160 && el < W5) *reinterpret_cast<uint4*>(rowp + el) = *reinterpret_cast<const uint4*>(wl + pc * 16);
  }
}

// scale[i] = keep ? 1/(1-p) : 0 for the 20 elements starting at flat index e0 (e0 % 4 == 0)
__device__ __forceinline__ void keep_scale20(const uint8_t* __restrict__ keep, uint64_t seed,
                                             uint32_t thr, float inv_keep, long long e0,
                                             float (&sc)[CPT]) {
  if (keep) {
    const uint32_t* k32 = reinterpret_cast<const uint32_t*>(keep + e0);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const uint32_t w = k32[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) sc[4 * i + j] = ((w >> (8 * j)) & 0xFFu) ? inv_keep : 0.f;
    }
  } else if (thr != 0u) {
    // elements e0 .. e0+19 are halfwords (e0 & 7) .. +19 of the 24-halfword stream of calls e0/8, e0/8 + 1, e0/8 + 2
    // (e0 % 4 == 0: the window starts at halfword 0 or 4, i.e. at word 0 or 2 of the 12-word stream)
    const uint64_t g0 = (uint64_t)(e0 >> 3);
    const uint4 r0 = philox4x32_10(g0, seed), r1 = philox4x32_10(g0 + 1, seed), r2 = philox4x32_10(g0 + 2, seed);
    const bool hi = (e0 & 4) != 0;
    const uint32_t t16 = thr >> 16;
    // The 12 words are NAMED VALUES picked by a switch that folds after unrolling, never an array: `hi ? w[k+2] : w[k]` on
    // an array became a dynamically indexed private array in the LDS-transposed kernels -- 48 bytes of scratch stores and
    // 40 of loads per thread and row, 1.2 GB of extra HBM writes per backward launch at the headline shape (rocprofv3
    // WRITE_SIZE 3.30 GB against 2.09 GB of dP + partials; profiles/r03_pmc_fuse.txt).
    auto word = [&](int i) -> uint32_t {
      switch (i) {
        case 0: return r0.x; case 1: return r0.y; case 2: return r0.z; case 3: return r0.w;
        case 4: return r1.x; case 5: return r1.y; case 6: return r1.z; case 7: return r1.w;
        case 8: return r2.x; case 9: return r2.y; case 10: return r2.z; default: return r2.w;
      }
    };
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const uint32_t lo_w = word(k), hi_w = word(k + 2);
      const uint32_t x = hi ? hi_w : lo_w;
      sc[2 * k] = (x & 0xFFFFu) >= t16 ? inv_keep : 0.f;
      sc[2 * k + 1] = (x >> 16) >= t16 ? inv_keep : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < CPT; ++i) sc[i] = 1.0f;
  }
}

// grid (N, LS); block 256.  Each block walks rows l = ls, ls+LS, ... of sample n with the sample's
// 20 q values and the 20 projection biases of its columns held in registers (one row per block re-read
// both for every row: 3x the load instructions); the next row's P chunk is fetched before the current
// row is reduced.  Dropout keys on the flat element index, so the masks do not depend on this mapping.
// COAL: 0 = direct (strided) P loads, 1 = LDS-transposed loads with the next row's pieces prefetched into registers,
// 2 = LDS-transposed loads issued at the top of the row (20 registers fewer: one more block per CU; A/B, option fuse_coal = 2)
template <typename PT, int COAL>
__global__ void __launch_bounds__(256)
mfb_fuse_fwd_kernel(const PT* __restrict__ P, const float* __restrict__ pbias,
                    const float* __restrict__ q,
                    const float* __restrict__ cascade, const uint8_t* __restrict__ keep,
                    uint64_t seed, uint32_t thr, float inv_keep, int L, int O, int LS,
                    float* __restrict__ R, float* __restrict__ rowssq, float* __restrict__ zdrop,
                    unsigned short* __restrict__ Rb, int ldrb) {
  // Rb != nullptr: a bf16 copy of R (round-to-nearest-even) with row pitch ldrb >= O, columns O .. ldrb-1 zero: the K-padded A
  // operand of the co-attention conv's bf16 GEMM (BASELINE config 3) straight from the registers that hold R -- no
  // vqf_cast_f32_bf16 pass over the 401 MB tensor.  ldrb / 4 <= 256.
#if VQF_FUSE_ROWBAR
  __shared__ float red[2][4];
#endif
  __shared__ __attribute__((aligned(16))) char tl[COAL ? 4 * WLDS : 16];
  const int n = blockIdx.x, ls = blockIdx.y;
  const int W5 = KP * O;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nt = O / TPT;                       // active threads (250 at O = 1000); O / TPT <= 256
  const bool act = tid < nt;
  float qq[CPT], pb[CPT], p[CPT], pn[CPT];
  Raw<PT> rn;                                   // COAL: the next row's span of this wave, lane-linear pieces
#pragma unroll
  for (int i = 0; i < CPT; ++i) { qq[i] = 0.f; pb[i] = 0.f; pn[i] = 0.f; p[i] = 0.f; }
  if (act) {
    load20(q + (long long)n * W5 + CPT * tid, qq);
    if (pbias) load20(pbias + CPT * tid, pb);
    if (!COAL && ls < L) load20(P + ((long long)n * L + ls) * W5 + CPT * tid, pn);
  }
  if (COAL == 1 && ls < L) raw_load(P + ((long long)n * L + ls) * W5, W5, wave, lane, rn);     // every lane of the wave
  int it = 0;
  for (int l = ls; l < L; l += LS, ++it) {
    const long long row = (long long)n * L + l;
    const long long e0 = row * W5 + (long long)CPT * tid;
    float ssq = 0.f;
    if (COAL == 1) {
      raw_to_own(rn, tl + wave * WLDS, lane, p);
      if (l + LS < L) raw_load(P + (row + LS) * W5, W5, wave, lane, rn);   // prefetch the next row of this block
    } else if (COAL == 2) {
      raw_load(P + row * W5, W5, wave, lane, rn);
      raw_to_own(rn, tl + wave * WLDS, lane, p);
    }
    if (act) {
      if (!COAL) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) p[i] = pn[i];
        if (l + LS < L) load20(P + (row + LS) * W5 + CPT * tid, pn);      // prefetch the next row of this block
      }
      float sc[CPT], cc[CPT];
      keep_scale20(keep, seed, thr, inv_keep, e0, sc);
#pragma unroll
      for (int i = 0; i < CPT; ++i) p[i] = (p[i] + pb[i]) * qq[i];
      if (cascade) {
        load20(cascade + e0, cc);
#pragma unroll
        for (int i = 0; i < CPT; ++i) p[i] *= cc[i];
      }
#pragma unroll
      for (int i = 0; i < CPT; ++i) p[i] *= sc[i];
      if (zdrop) store20(zdrop + e0, p);
      f32x4 r;
#pragma unroll
      for (int j = 0; j < TPT; ++j) {
        const float s = (((p[5 * j] + p[5 * j + 1]) + p[5 * j + 2]) + p[5 * j + 3]) + p[5 * j + 4];
        const float a = fabsf(s);
        ssq += a;                                  // (sign(s) sqrt|s|)^2 == |s|
        const float rt = sqrtf(a);
        r[j] = s < 0.f ? -rt : rt;                 // sqrt(relu(s)) - sqrt(relu(-s))
      }
      *reinterpret_cast<f32x4*>(R + row * O + TPT * tid) = r;
      if (Rb) {
        __bf16 rb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[j] = (__bf16)r[j];
        *reinterpret_cast<uint2*>(Rb + row * ldrb + TPT * tid) = *reinterpret_cast<const uint2*>(rb);
      }
    } else if (Rb && TPT * tid < ldrb) {
      *reinterpret_cast<uint2*>(Rb + row * ldrb + TPT * tid) = make_uint2(0u, 0u);
    }
    ssq = wave_sum(ssq);
#if VQF_FUSE_ROWBAR
    if ((tid & 63) == 0) red[it & 1][tid >> 6] = ssq;
    __syncthreads();                               // red[] is double-buffered: one barrier per row
    if (tid == 0) {
      rowssq[4 * row] = (red[it & 1][0] + red[it & 1][1]) + (red[it & 1][2] + red[it & 1][3]);
      rowssq[4 * row + 1] = 0.f; rowssq[4 * row + 2] = 0.f; rowssq[4 * row + 3] = 0.f;
    }
#else
    // every wave leaves the sum of ITS quarter of the row: no workgroup barrier in the row loop, the four waves of a block drift
    // apart and cover each other's load latency; vqf_l2_group_norm adds the 4 L partials of a sample in a fixed order
    if ((tid & 63) == 0) rowssq[4 * row + (tid >> 6)] = ssq;
#endif
  }
}

// grid (N, LS); block 256.  Each block walks rows l = ls, ls+LS, ... of sample n.
template <bool CASC, bool DBIAS, typename DPT, typename PT, bool COAL>
__global__ void __launch_bounds__(256)
mfb_fuse_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ dzdrop,
                    const float* __restrict__ Y,
                    const float* __restrict__ inv, const float* __restrict__ coefA,
                    const float* __restrict__ coefB, const PT* __restrict__ P,
                    const float* __restrict__ pbias,
                    const float* __restrict__ q, const float* __restrict__ cascade,
                    const uint8_t* __restrict__ keep, uint64_t seed, uint32_t thr, float inv_keep,
                    int L, int O, int LS, DPT* __restrict__ dP, float* __restrict__ dq_part,
                    float* __restrict__ dcascade, float* __restrict__ db_part) {
  __shared__ __attribute__((aligned(16))) char tl[COAL ? 4 * WLDS : 16];
  const int n = blockIdx.x, ls = blockIdx.y;
  const int W5 = KP * O;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float ca = coefA[n], cb = coefB[n], hi = 0.5f * inv[n];
  // COAL (O / 4 <= 256: one pass): every lane of a wave takes part in the coalesced P loads / dP stores, threads past
  // O / 4 only skip the arithmetic
  for (int t = threadIdx.x; COAL ? t < 256 : t < O / TPT; t += 256) {
    const bool act = t < O / TPT;
    float qq[CPT], dq[CPT], db[CPT], pb[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) { qq[i] = 0.f; pb[i] = 0.f; dq[i] = 0.f; db[i] = 0.f; }
    if (act) {
      load20(q + (long long)n * W5 + CPT * t, qq);
      if (pbias) load20(pbias + CPT * t, pb);
    }
    for (int l = ls; l < L; l += LS) {
      const long long row = (long long)n * L + l;
      const long long e0 = row * W5 + (long long)CPT * t;
      float p[CPT], sc[CPT], cc[CPT], dp[CPT], dc[CPT], dzx[CPT];
#pragma unroll
      for (int i = 0; i < CPT; ++i) { p[i] = 0.f; dp[i] = 0.f; }
      if (COAL) {
        Raw<PT> rw;
        raw_load(P + row * W5, W5, wave, lane, rw);
        raw_to_own(rw, tl + wave * WLDS, lane, p);
      }
      if (act) {
        const f32x4 dy = *reinterpret_cast<const f32x4*>(dY + row * O + TPT * t);
        const f32x4 y = *reinterpret_cast<const f32x4*>(Y + row * O + TPT * t);
        if (!COAL) load20(P + e0, p);
        if (pbias) {
#pragma unroll
          for (int i = 0; i < CPT; ++i) p[i] += pb[i];
        }
        if (CASC) load20(cascade + e0, cc);
        keep_scale20(keep, seed, thr, inv_keep, e0, sc);
        float ds[TPT];
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
          const float ay = fabsf(y[j]);
          // d sqrt(relu(s)) - sqrt(relu(-s)) = 0.5/|R| for R != 0, and 0 at 0 (relu'(0) = 0)
          ds[j] = ay > 0.f ? (ca * dy[j] - cb * y[j]) * (hi / ay) : 0.f;
        }
        if (dzdrop) load20(dzdrop + e0, dzx);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
          const float dz = (dzdrop ? ds[i / KP] + dzx[i] : ds[i / KP]) * sc[i];
          if (CASC) {
            dp[i] = dz * qq[i] * cc[i];
            dq[i] += dz * p[i] * cc[i];
            dc[i] = dz * p[i] * qq[i];
          } else {
            dp[i] = dz * qq[i];
            dq[i] += dz * p[i];
          }
          if (DBIAS) db[i] += dp[i];
        }
        if (!COAL) store20(dP + e0, dp);
        if (CASC) store20(dcascade + e0, dc);
      }
      if (COAL) own_store(dP + row * W5, W5, wave, lane, tl + wave * WLDS, dp);
    }
    if (act) {
      const long long po = ((long long)n * LS + ls) * W5 + CPT * t;
      store20(dq_part + po, dq);
      if (DBIAS) store20(db_part + po, db);
    }
  }
}

// Coalesced (LDS-transposed) P / dP access.  Measured at the headline shape (tools/fuse_bench.py, profiles/r03_fuse_ab.log,
// r03_fuse_ab2.log, dropout 0.1): fp32 forward 0.64 -> 0.50 ms, fp32 backward 1.08-1.16 -> 0.83-0.89 ms, bf16-P forward 0.43 ->
// 0.37 ms; bf16 backward 0.576 -> 0.594 on one box and 0.586 -> 0.554 on another (coalesced is the default there too).  The
// forward WITHOUT its register prefetch of the next row (COAL = 2: 126 VGPRs, four blocks per CU instead of three) is another
// 3 % faster: 0.496 -> 0.481 ms (5.0 TB/s), bf16-P 0.370 -> 0.358.  Round 2 had measured the forward as a tie and the bf16
// kernels as slower: those variants were spilling the Philox words to scratch (see keep_scale20).  Option fuse_coal = 0 forces
// the direct access everywhere, 1 the round-3 coalesced forward WITH prefetch.
bool fuse_coalesced(bool dflt) {
  const int v = g_vqf_opt[VQF_OPT_FUSE_COAL];
  return v < 0 ? dflt : v != 0;
}

int pick_ls_fwd(int N, int L) {
  // forward: 8 blocks per CU worth of (sample, row-subset) pairs, at least ~8 rows per block
  { const int v = g_vqf_opt[VQF_OPT_FUSE_LS]; if (v >= 1 && v <= L) return v; }   // tuning probe
  int ls = 1;
  while ((long long)N * ls < 4096 && ls * 16 <= L) ls *= 2;
  return ls;
}

int pick_ls(int N, int L) {
  // enough blocks to cover 256 CUs x 4, but never more splits than rows
  { const int v = g_vqf_opt[VQF_OPT_FUSE_LS_BWD]; if (v >= 1 && v <= L && v <= 16) return v; }   // tuning probe
  int ls = 1;
  while ((long long)N * ls < 2048 && ls * 2 <= L && ls < 16) ls *= 2;
  return ls;
}

}  // namespace

static int fuse_bwd_impl(const float* dY, const float* dzdrop, const float* Y, const float* inv,
                     const float* coefA,
                     const float* coefB, const void* P, int p_bf16, const float* pbias, const float* q,
                     const float* cascade,
                     const uint8_t* keep, uint64_t seed, float p_drop, int N, int L, int O,
                     void* dP, int dp_bf16, float* dq, float* dcascade, float* dbiasP, void* ws,
                     size_t ws_bytes, void* stream) {
  if (!dY || !Y || !inv || !coefA || !coefB || !P || !q || !dP || !dq || N <= 0 || L <= 0 || O <= 0)
    return VQF_E_BADARG;
  if (O % TPT) return VQF_E_UNSUPPORTED;
  if ((cascade != nullptr) != (dcascade != nullptr)) return VQF_E_BADARG;
  if (dp_bf16 && cascade) return VQF_E_UNSUPPORTED;
  if (p_bf16 && !dp_bf16) return VQF_E_UNSUPPORTED;          // bf16 P only in the all-bf16 image fusion
  if (p_drop < 0.f || p_drop >= 1.f) return VQF_E_BADARG;
  if (!aligned16(dY) || (dzdrop && !aligned16(dzdrop)) || (pbias && !aligned16(pbias)) || !aligned16(Y) || !aligned16(P) || !aligned16(q) || !aligned16(dP) ||
      !aligned16(dq) || (cascade && (!aligned16(cascade) || !aligned16(dcascade))) ||
      (keep && (((uintptr_t)keep) & 3)))
    return VQF_E_ALIGN;
  const int LS = pick_ls(N, L);
  const int W5 = KP * O;
  const bool direct = (LS == 1);      // dq partial == dq
  if (!direct || dbiasP) {
    if (!ws || ws_bytes < vqf_mfb_fuse_bwd_ws_bytes(N, L, O) || !aligned16(ws)) return VQF_E_WORKSPACE;
  }
  float* dq_part = direct ? dq : (float*)ws;
  float* db_part = (float*)ws + (size_t)N * LS * W5;
  const uint32_t thr = (keep || p_drop == 0.f) ? 0u : drop_threshold_host(p_drop);
  const float inv_keep = (keep || p_drop > 0.f) ? 1.0f / (1.0f - p_drop) : 1.0f;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(N, LS);
  // coalesced P / dP access through the LDS transpose: one pass of 256 threads over the row, whole 16-byte pieces
  const bool coal = fuse_coalesced(true) && O / TPT <= 256 && (W5 % ((p_bf16 || dp_bf16) ? 8 : 4) == 0);
#define VQF_BWD1(C_, D_, T_, PT_, CO_)                                                           \
  VQF_LAUNCH(KID_MFB_FUSE_BWD, (mfb_fuse_bwd_kernel<C_, D_, T_, PT_, CO_>), grid, dim3(256), 0, s, dY, dzdrop, Y, inv, \
             coefA, coefB, (const PT_*)P, pbias, q, cascade, keep, seed, thr, inv_keep, L, O, LS, (T_*)dP, dq_part, \
             dcascade, db_part)
#define VQF_BWD(C_, D_, T_, PT_) do { if (coal) VQF_BWD1(C_, D_, T_, PT_, true); else VQF_BWD1(C_, D_, T_, PT_, false); } while (0)
  if (p_bf16)       { if (dbiasP) VQF_BWD(false, true, __bf16, __bf16); else VQF_BWD(false, false, __bf16, __bf16); }
  else if (dp_bf16) { if (dbiasP) VQF_BWD(false, true, __bf16, float); else VQF_BWD(false, false, __bf16, float); }
  else if (cascade) { if (dbiasP) VQF_BWD(true, true, float, float); else VQF_BWD(true, false, float, float); }
  else              { if (dbiasP) VQF_BWD(false, true, float, float); else VQF_BWD(false, false, float, float); }
#undef VQF_BWD
#undef VQF_BWD1
  int rc = vqf_last_error();
  if (rc) return rc;
  if (!direct) {
    rc = vqf_group_reduce_f32(dq_part, N, LS, W5, dq, stream);
    if (rc) return rc;
  }
  if (dbiasP)
    rc = vqf_colreduce_2stage(db_part, N * LS, W5, dbiasP, (float*)ws + (size_t)2 * N * LS * W5, s);
  return rc;
}



static int fuse_fwd_impl(const void* P, int p_bf16, const float* pbias, const float* q, const float* cascade,
                         const uint8_t* keep, uint64_t seed, float p_drop, int N, int L, int O, float* R,
                         float* rowssq, float* zdrop, void* stream, void* R_bf16 = nullptr, int ldrb = 0) {
  if (!P || !q || !R || !rowssq || N <= 0 || L <= 0 || O <= 0) return VQF_E_BADARG;
  if (R_bf16 && (ldrb < O || (ldrb % 4) || ldrb / 4 > 256 || (((uintptr_t)R_bf16) & 7))) return VQF_E_BADARG;
  if ((O % TPT) || O / TPT > 256) return VQF_E_UNSUPPORTED;     // one thread per 4 pooled outputs: O <= 1024 (the reference's 1000)
  if (p_drop < 0.f || p_drop >= 1.f) return VQF_E_BADARG;
  if (!aligned16(P) || (pbias && !aligned16(pbias)) || !aligned16(q) || !aligned16(R) ||
      (cascade && !aligned16(cascade)) ||
      (zdrop && !aligned16(zdrop)) || (keep && (((uintptr_t)keep) & 3)))
    return VQF_E_ALIGN;
  const uint32_t thr = (keep || p_drop == 0.f) ? 0u : drop_threshold_host(p_drop);
  const float inv_keep = (keep || p_drop > 0.f) ? 1.0f / (1.0f - p_drop) : 1.0f;
  const int LS = pick_ls_fwd(N, L);
  const bool coal = fuse_coalesced(true) && ((KP * O) % (p_bf16 ? 8 : 4) == 0);
  const bool nopf = coal && g_vqf_opt[VQF_OPT_FUSE_COAL] != 1;   // default: no register prefetch (126 VGPRs, 4 blocks per CU)
#define VQF_FWD(PT_, CO_)                                                                                             \
  VQF_LAUNCH(KID_MFB_FUSE_FWD, (mfb_fuse_fwd_kernel<PT_, CO_>), dim3(N, LS), dim3(256), 0, (hipStream_t)stream,         \
             (const PT_*)P, pbias, q, cascade, keep, seed, thr, inv_keep, L, O, LS, R, rowssq, zdrop, (unsigned short*)R_bf16, ldrb)
  if (p_bf16) { if (nopf) VQF_FWD(__bf16, 2); else if (coal) VQF_FWD(__bf16, 1); else VQF_FWD(__bf16, 0); }
  else        { if (nopf) VQF_FWD(float, 2); else if (coal) VQF_FWD(float, 1); else VQF_FWD(float, 0); }
#undef VQF_FWD
  return vqf_last_error();
}

extern "C" {

int vqf_mfb_fuse_fwd(const float* P, const float* pbias, const float* q, const float* cascade,
                     const uint8_t* keep,
                     uint64_t seed, float p_drop, int N, int L, int O, float* R, float* rowssq,
                     float* zdrop, void* stream) {
  return fuse_fwd_impl(P, 0, pbias, q, cascade, keep, seed, p_drop, N, L, O, R, rowssq, zdrop, stream);
}

int vqf_mfb_fuse_fwd_pbf16(const void* P_bf16, const float* pbias, const float* q, const uint8_t* keep, uint64_t seed,
                           float p_drop, int N, int L, int O, float* R, float* rowssq, void* stream) {
  return fuse_fwd_impl(P_bf16, 1, pbias, q, nullptr, keep, seed, p_drop, N, L, O, R, rowssq, nullptr, stream);
}

int vqf_mfb_fuse_fwd_pbf16_rb(const void* P_bf16, const float* pbias, const float* q, const uint8_t* keep, uint64_t seed,
                              float p_drop, int N, int L, int O, float* R, void* R_bf16, int ldrb, float* rowssq, void* stream) {
  if (!R_bf16) return VQF_E_BADARG;
  return fuse_fwd_impl(P_bf16, 1, pbias, q, nullptr, keep, seed, p_drop, N, L, O, R, rowssq, nullptr, stream, R_bf16, ldrb);
}

size_t vqf_mfb_fuse_bwd_ws_bytes(int N, int L, int O) {
  if (N <= 0 || L <= 0 || O <= 0) return 0;
  return ((size_t)2 * N * pick_ls(N, L) + VQF_REDUCE_SPLITS) * KP * O * sizeof(float);
}

int vqf_mfb_fuse_bwd(const float* dY, const float* dzdrop, const float* Y, const float* inv,
                     const float* coefA, const float* coefB, const float* P, const float* pbias, const float* q,
                     const float* cascade, const uint8_t* keep, uint64_t seed, float p_drop, int N, int L, int O,
                     float* dP, float* dq, float* dcascade, float* dbiasP, void* ws, size_t ws_bytes, void* stream) {
  return fuse_bwd_impl(dY, dzdrop, Y, inv, coefA, coefB, P, 0, pbias, q, cascade, keep, seed, p_drop, N, L, O, dP, 0, dq,
                       dcascade, dbiasP, ws, ws_bytes, stream);
}

int vqf_mfb_fuse_bwd_pbf16(const float* dY, const float* Y, const float* inv, const float* coefA,
                           const float* coefB, const void* P_bf16, const float* pbias, const float* q,
                           const uint8_t* keep, uint64_t seed, float p_drop, int N, int L, int O, void* dP_bf16,
                           float* dq, float* dbiasP, void* ws, size_t ws_bytes, void* stream) {
  return fuse_bwd_impl(dY, nullptr, Y, inv, coefA, coefB, P_bf16, 1, pbias, q, nullptr, keep, seed, p_drop, N, L, O,
                       dP_bf16, 1, dq, nullptr, dbiasP, ws, ws_bytes, stream);
}

int vqf_mfb_fuse_bwd_bf16dp(const float* dY, const float* Y, const float* inv, const float* coefA,
                            const float* coefB, const float* P, const float* pbias, const float* q,
                            const uint8_t* keep, uint64_t seed, float p_drop, int N, int L, int O, void* dP_bf16,
                            float* dq, float* dbiasP, void* ws, size_t ws_bytes, void* stream) {
  return fuse_bwd_impl(dY, nullptr, Y, inv, coefA, coefB, P, 0, pbias, q, nullptr, keep, seed, p_drop, N, L, O, dP_bf16, 1,
                       dq, nullptr, dbiasP, ws, ws_bytes, stream);
}

}  // extern "C"
