// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), gfx950 only.
//
//   C[m,n] (+)= sum_k Aop[m,k] * Bop[n,k] (+ bias[n]) (relu)
//
// Replaces the nn.Linear / 1x1 nn.Conv2d forward, dgrad and wgrad GEMMs of the
// reference path (list in include/vqa_fusion.h).  The dominant instance is the
// image projection img_conv1d (mfb.py:96): M = N*196, K = 2048, N = 5000.
//
// Design (MI355X-first, not a warp-tiling port):
//   * 256-thread workgroup = 4 wave64, one per SIMD; 4 workgroups per CU (127 VGPRs, 40 KB LDS).
//   * 128x128x16 block tile, each wave owns a 64x64 sub-tile = 2x2 MFMA
//     32x32 accumulators (64 accumulator VGPRs); the fp32 MFMA is 64 cycles per
//     instruction per SIMD, so 4 independent accumulators per wave keep the
//     pipe back-to-back and LDS/global traffic is <15 % of the issue slots.
//   * operands are staged global -> registers -> LDS with the NEXT tile's global
//     loads issued before the current tile's MFMAs (register prefetch) and two
//     LDS buffers, so there is one barrier per K-tile.
//   * "row-major K-contiguous" operands (activations, weights in forward) live
//     in LDS as [row][BK+4] and are read with one ds_read_b128 per 8 k (the 16-B
//     row pad makes the 16-lane b128 groups conflict-free); "K-major" operands
//     (both operands of wgrad, the weight in dgrad) live as [k][128] and are read
//     with conflict-free ds_read_b32.  Both give lane (i, h) the k-indices
//     {8g+4h+j}, j=0..3, so A and B fragments always agree on k.
//   * workgroup ids are remapped so that each XCD (private 4 MiB L2) walks a
//     contiguous range of output tiles: the 40 column tiles that share one
//     128-row slab of the (N*196, 2048) image tensor hit that XCD's L2.
//   * split-K (grid.y) with a deterministic slab reduction for the wgrad
//     shapes (few output tiles, K = N*196).
#include "common.h"
#include <atomic>

namespace {

// Measured on MI355X (img_conv1d forward, M=100352 N=5000 K=2048): BK=32 / 2 workgroups per CU
// 128.9 TF; BK=16 / 3 per CU 132.3 TF; BK=16 / 4 per CU 133.7 TF.  The shallower K-tile halves the
// LDS footprint (40,960 B) so that four workgroups (4 waves per SIMD) share a CU and cover each
// other's barrier / LDS-refill bubbles.
#ifndef VQF_SPLITK_WT
#define VQF_SPLITK_WT 0        // 1: write-through slab stores instead of a release fence in the in-launch split-K combine (A/B)
#endif
#ifndef VQF_GEMM_BK
#define VQF_GEMM_BK 16
#endif
#ifndef VQF_GEMM_WAVES_PER_SIMD
#define VQF_GEMM_WAVES_PER_SIMD 4
#endif
// VQF_GEMM_GLDS = 1: tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no
// ds_write pass).  The LDS destination of that instruction is wave-uniform base + lane*16 B, so
// the K-contiguous image is UNPADDED [row][16] and bank conflicts are avoided by an XOR swizzle of
// the 16-byte chunk index, applied on the global SOURCE address and again on the fragment reads.
#ifndef VQF_GEMM_GLDS
#define VQF_GEMM_GLDS 0
#endif
constexpr int BM = 128, BN = 128, BK = VQF_GEMM_BK, NTHREADS = 256;
constexpr bool GLDS = (VQF_GEMM_GLDS != 0) && (BK == 16);
constexpr int LD_RK = GLDS ? BK : BK + 4;     // floats; [row][k] image (16-B pad unless swizzled)
__device__ __forceinline__ int rk_swz(int row) { return GLDS ? ((row >> 2) & 3) : 0; }   // chunk XOR
constexpr int OP_FLOATS = BM * LD_RK;         // >= BK*BM
constexpr int STAGE_FLOATS = 2 * OP_FLOATS;   // A + B
constexpr int SMEM_BYTES = 2 * STAGE_FLOATS * 4;   // 73,728 B at BK=32 (double buffered)
constexpr int K4 = BK / 4;                    // float4 per tile row (K-contiguous layout)
constexpr int NLD = BM * BK / 4 / NTHREADS;   // float4 loads per thread per operand per tile
static_assert(BK % 8 == 0 && NLD >= 1, "BK must be a multiple of 8");

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  float* slab;             // split-K partials [splits][M][N] or nullptr
  int M, N, K, lda, ldb, ldc;
  int flags;
  int kchunk;              // K range per grid.y slice (multiple of BK)
  int tiles_m, tiles_n;
  int vecA, vecB;          // 16-B vector loads allowed
  long long sA, sB, sC;    // batch strides (grid.z), 0 when not batched
  const float* rowscale;   // epilogue: C = rowscale[(row0 + row) / rps] * acc + bias (vqf_gemm_f32_rowscale), or nullptr
  int rps;
  int row0;                // rows of the product that another launch computed (the large-tile kernel's whole-rounds block)
  int* cnt;                // split-K combined IN the launch: one arrival counter per output tile (zero before and after), or nullptr
};

// Thread -> (row, k) of its i-th float4 in the K-contiguous ("RK") tile image.  A ds_write_b128
// is serviced in groups of 8 consecutive lanes; with 64-byte tile rows (BK = 16) padded to 80 B,
// two ADJACENT rows overlap on 4 banks (2-way conflict on the staging stores; the fragment reads
// are conflict-free either way).  Remapping each 8-lane group to rows (r, r+4) removes the
// conflict (SQ_LDS_BANK_CONFLICT 5.1e8 -> 0, LDS busy -33 %) but measured 3 % SLOWER in a
// same-process A/B (130.5 vs 134.6 TF, tools/gemm_ab.py): the kernel is not LDS-bound and the
// plain order keeps consecutive lanes on consecutive global rows.  Hence off by default.
#ifndef VQF_GEMM_RK_REMAP
#define VQF_GEMM_RK_REMAP 0
#endif
__device__ __forceinline__ int rk_row(int f) {
  if (VQF_GEMM_RK_REMAP && K4 == 4) return ((f >> 5) << 3) + (((f >> 2) & 1) << 2) + ((f >> 3) & 3);
  return f / K4;
}
__device__ __forceinline__ int rk_k(int f) { return (f % K4) << 2; }

// ---- global -> register staging ------------------------------------------
// T == false: operand(row r, k) = p[r*ld + k]   (tile rows x BK, 8 float4 per row)
// T == true : operand(row r, k) = p[k*ld + r]   (BK k-rows x 128, 32 float4 per k-row)
template <bool T>
__device__ __forceinline__ void load_tile(const float* __restrict__ p, int ld, int r0, int R,
                                          int k0, int kend, bool vec, int tid, f32x4 (&v)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = tid + NTHREADS * i;
    int r, k;
    if (!T) { r = r0 + rk_row(f); k = k0 + rk_k(f); }
    else    { k = k0 + (f >> 5); r = r0 + ((f & 31) << 2); }
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (!T) {
      if (r < R) {
        const float* q = p + (long long)r * ld + k;
        if (vec && k + 3 < kend) x = *reinterpret_cast<const f32x4*>(q);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k + j < kend) x[j] = q[j];
        }
      }
    } else {
      if (k < kend) {
        const float* q = p + (long long)k * ld + r;
        if (vec && r + 3 < R) x = *reinterpret_cast<const f32x4*>(q);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (r + j < R) x[j] = q[j];
        }
      }
    }
    v[i] = x;
  }
}

// ---- fast path: per-thread source pointers computed ONCE, rows clamped into range -------------
// A row beyond the matrix edge only feeds accumulators whose outputs are never stored, so the
// loader clamps its row index instead of guarding (no branches / exec masking in the K loop).
// Requires 16-B aligned base, ld % 4 == 0 and, for the K-major layout, R % 4 == 0.
// Global address space (1) is spelled out: pointers kept in arrays otherwise decay to generic
// and hipcc emits flat_load, which also counts on lgkmcnt -- every LDS wait would then wait for
// the prefetch of the NEXT tile as well.
typedef const float __attribute__((address_space(1))) gfloat;
typedef const f32x4 __attribute__((address_space(1))) gf32x4;

template <bool T>
__device__ __forceinline__ void init_ptrs(const float* __restrict__ p0, int ld, int r0, int R, int k0,
                                          int tid, gfloat* (&q)[NLD]) {
  gfloat* p = (gfloat*)p0;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = tid + NTHREADS * i;
    if (!T) {
      const int r = min(r0 + rk_row(f), R - 1);
      q[i] = p + (long long)r * ld + (k0 + rk_k(f));
    } else {
      const int r = min(r0 + ((f & 31) << 2), R - 4);
      q[i] = p + (long long)(k0 + (f >> 5)) * ld + r;
    }
  }
}
template <bool T>
__device__ __forceinline__ void load_fast(gfloat* (&q)[NLD], int ld, f32x4 (&v)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    v[i] = *(gf32x4*)q[i];
    q[i] += T ? (long long)BK * ld : BK;
  }
}

// ---- direct global -> LDS staging (GLDS): each wave issues 2 x 1 KiB per operand and tile -------
constexpr int NGL = (BM * BK) / (4 * 64 * 4);     // wave-instructions per wave per operand (2 at BK=16)
template <bool T>
__device__ __forceinline__ void init_glds_ptrs(const float* __restrict__ p0, int ld, int r0, int R, int k0,
                                               int wave, int lane, gfloat* (&q)[NGL]) {
  gfloat* p = (gfloat*)p0;
#pragma unroll
  for (int i = 0; i < NGL; ++i) {
    const int blk = wave * NGL + i;               // 1-KiB block of the 8-KiB operand image
    if (!T) {
      const int row = blk * 16 + (lane >> 2);
      const int c = (lane & 3) ^ rk_swz(row);     // source chunk that belongs at LDS chunk position lane&3
      q[i] = p + (long long)min(r0 + row, R - 1) * ld + (k0 + 4 * c);
    } else {
      const int k = blk * 2 + (lane >> 5);
      q[i] = p + (long long)(k0 + k) * ld + min(r0 + ((lane & 31) << 2), R - 4);
    }
  }
}
template <bool T>
__device__ __forceinline__ void glds_tile(gfloat* (&q)[NGL], int ld, float* s, int wave) {
  typedef __attribute__((address_space(3))) float lds_float;
#pragma unroll
  for (int i = 0; i < NGL; ++i) {
    lds_float* dst = (lds_float*)(s + (wave * NGL + i) * 256);
    __builtin_amdgcn_global_load_lds(q[i], dst, 16, 0, 0);
    q[i] += T ? (long long)BK * ld : BK;
  }
}

template <bool T>
__device__ __forceinline__ void store_tile(float* s, int tid, const f32x4 (&v)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = tid + NTHREADS * i;
    if (!T) *reinterpret_cast<f32x4*>(s + rk_row(f) * LD_RK + (((rk_k(f) >> 2) ^ rk_swz(rk_row(f))) << 2)) = v[i];
    else    *reinterpret_cast<f32x4*>(s + (f >> 5) * BM + ((f & 31) << 2)) = v[i];
  }
}

// fragment of one 32-row sub-tile for k-group g8 (8 consecutive k)
template <bool T>
__device__ __forceinline__ f32x4 read_frag(const float* s, int row, int g8, int h) {
  if (!T) return *reinterpret_cast<const f32x4*>(s + row * LD_RK + (((2 * g8 + h) ^ rk_swz(row)) << 2));
  f32x4 x;
  const float* q = s + (g8 * 8 + 4 * h) * BM + row;
  x[0] = q[0]; x[1] = q[BM]; x[2] = q[2 * BM]; x[3] = q[3 * BM];
  return x;
}

// K loop of one 128x128 output tile.  EDGE = false: all four 32x32 sub-tiles of this wave are
// inside the matrix (the common case, no predication anywhere in the loop).  EDGE = true: tile on
// the matrix border; sub-tiles that are entirely outside are skipped (N = 5000 = 39*128 + 8).
template <bool TA, bool TB, bool FAST, bool EDGE>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& g, float* smem, const float* gA,
                                              const float* gB, int m0, int n0, int kbeg, int kend,
                                              int tid, int wr, int wc, int l31, int h,
                                              f32x16 (&acc)[2][2]) {
  const int ntiles = (kend - kbeg + BK - 1) / BK;
  const int nfull = (kend - kbeg) / BK;            // tiles that need no K guard
  bool v10 = true, v01 = true;                     // sub-tile (1,0) / (0,1) inside?  (EDGE only)
  if (EDGE) {
    v10 = (m0 + wr * 64 + 32) < g.M;
    v01 = (n0 + wc * 64 + 32) < g.N;
  }
  f32x4 ra[NLD], rb[NLD];
  auto mma_tile = [&](const float* sA, const float* sB) {
#pragma unroll
    for (int g8 = 0; g8 < BK / 8; ++g8) {
      if (!EDGE) {
        f32x4 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = read_frag<TA>(sA, wr * 64 + i * 32 + l31, g8, h);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = read_frag<TB>(sB, wc * 64 + j * 32 + l31, g8, h);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
      } else {
        const f32x4 a0 = read_frag<TA>(sA, wr * 64 + l31, g8, h);
        const f32x4 b0 = read_frag<TB>(sB, wc * 64 + l31, g8, h);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b0[kk], acc[0][0], 0, 0, 0);
        if (v10) {
          const f32x4 a1 = read_frag<TA>(sA, wr * 64 + 32 + l31, g8, h);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b0[kk], acc[1][0], 0, 0, 0);
        }
        if (v01) {
          const f32x4 b1 = read_frag<TB>(sB, wc * 64 + 32 + l31, g8, h);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b1[kk], acc[0][1], 0, 0, 0);
        }
      }
    }
  };

  if (FAST && GLDS) {
    // direct-to-LDS pipeline: the DMA of tile t+1 is in flight while tile t is multiplied; the
    // __syncthreads() at the end of an iteration carries the s_waitcnt vmcnt(0) that retires it.
    const int lane = tid & 63, wave = 2 * wr + wc;
    gfloat* ga[NGL];
    gfloat* gb[NGL];
    init_glds_ptrs<TA>(gA, g.lda, m0, g.M, kbeg, wave, lane, ga);
    init_glds_ptrs<TB>(gB, g.ldb, n0, g.N, kbeg, wave, lane, gb);
    if (nfull > 0) {
      glds_tile<TA>(ga, g.lda, smem, wave);
      glds_tile<TB>(gb, g.ldb, smem + OP_FLOATS, wave);
    } else {
      load_tile<TA>(gA, g.lda, m0, g.M, kbeg, kend, g.vecA, tid, ra);
      load_tile<TB>(gB, g.ldb, n0, g.N, kbeg, kend, g.vecB, tid, rb);
      store_tile<TA>(smem, tid, ra);
      store_tile<TB>(smem + OP_FLOATS, tid, rb);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
      float* sA = smem + (t & 1) * STAGE_FLOATS;
      float* sB = sA + OP_FLOATS;
      float* dA = smem + ((t + 1) & 1) * STAGE_FLOATS;
      const bool more = (t + 1) < ntiles;
      const bool dma = (t + 1) < nfull;
      if (!dma && more) {                          // K tail: guarded register path (zero fill)
        const int k0 = kbeg + (t + 1) * BK;
        load_tile<TA>(gA, g.lda, m0, g.M, k0, kend, g.vecA, tid, ra);
        load_tile<TB>(gB, g.ldb, n0, g.N, k0, kend, g.vecB, tid, rb);
      }
      if (!EDGE) {
        // all fragment reads of tile t FIRST, then the DMA of tile t+1, then the MFMAs: hipcc puts
        // an s_waitcnt vmcnt(0) in front of any LDS read that follows an LDS-DMA (it cannot tell the
        // two buffers apart), so a DMA issued before the reads would be waited for immediately.
        f32x4 af[BK / 8][2], bf[BK / 8][2];
#pragma unroll
        for (int g8 = 0; g8 < BK / 8; ++g8) {
#pragma unroll
          for (int i = 0; i < 2; ++i) af[g8][i] = read_frag<TA>(sA, wr * 64 + i * 32 + l31, g8, h);
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[g8][j] = read_frag<TB>(sB, wc * 64 + j * 32 + l31, g8, h);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (dma) {
          glds_tile<TA>(ga, g.lda, dA, wave);
          glds_tile<TB>(gb, g.ldb, dA + OP_FLOATS, wave);
        }
#pragma unroll
        for (int g8 = 0; g8 < BK / 8; ++g8)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g8][i][kk], bf[g8][j][kk], acc[i][j], 0, 0, 0);
      } else {
        if (dma) {
          glds_tile<TA>(ga, g.lda, dA, wave);
          glds_tile<TB>(gb, g.ldb, dA + OP_FLOATS, wave);
        }
        mma_tile(sA, sB);
      }
      if (more && !dma) {
        store_tile<TA>(dA, tid, ra);
        store_tile<TB>(dA + OP_FLOATS, tid, rb);
      }
      __syncthreads();
    }
    return;
  }

  gfloat* pa[NLD];
  gfloat* pb[NLD];
  if (FAST) {
    init_ptrs<TA>(gA, g.lda, m0, g.M, kbeg, tid, pa);
    init_ptrs<TB>(gB, g.ldb, n0, g.N, kbeg, tid, pb);
  }
  if (FAST && nfull > 0) {
    load_fast<TA>(pa, g.lda, ra);
    load_fast<TB>(pb, g.ldb, rb);
  } else {
    load_tile<TA>(gA, g.lda, m0, g.M, kbeg, kend, g.vecA, tid, ra);
    load_tile<TB>(gB, g.ldb, n0, g.N, kbeg, kend, g.vecB, tid, rb);
  }
  store_tile<TA>(smem, tid, ra);
  store_tile<TB>(smem + OP_FLOATS, tid, rb);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const float* sA = smem + (t & 1) * STAGE_FLOATS;
    const float* sB = sA + OP_FLOATS;
    const bool more = (t + 1) < ntiles;
    if (more) {                                    // prefetch tile t+1 into registers
      if (FAST && (t + 1) < nfull) {
        load_fast<TA>(pa, g.lda, ra);
        load_fast<TB>(pb, g.ldb, rb);
      } else {
        const int k0 = kbeg + (t + 1) * BK;
        load_tile<TA>(gA, g.lda, m0, g.M, k0, kend, g.vecA, tid, ra);
        load_tile<TB>(gB, g.ldb, n0, g.N, k0, kend, g.vecB, tid, rb);
      }
    }
    mma_tile(sA, sB);
    if (more) {
      float* d = smem + ((t + 1) & 1) * STAGE_FLOATS;
      store_tile<TA>(d, tid, ra);
      store_tile<TB>(d + OP_FLOATS, tid, rb);
    }
    __syncthreads();
  }
}

template <bool TA, bool TB, bool FAST>
__global__ void __launch_bounds__(NTHREADS, VQF_GEMM_WAVES_PER_SIMD) gemm_f32_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (scalar branches)
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  // Tile order.  (1) XCD-aware bijective remap: blocks b and b+8 share an XCD (private 4 MiB L2),
  // so each XCD walks a contiguous range of the linear tile order.  (2) The linear order is grouped:
  // GROUP_M row-tiles x all column-tiles per group, walked row-tile-fastest, so the ~64 blocks
  // resident on one XCD cover ~8x8 tiles and share 8 A slabs + 8 B slabs in that L2.
  const int nwg = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * g.tiles_n;
  const int gid = wg / per_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(g.tiles_m - first_m, GROUP_M);
  const int in_g = wg - gid * per_group;
  const int tm = first_m + in_g % gsz, tn = in_g / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.y * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* gA = g.A + (long long)blockIdx.z * g.sA;
  const float* gB = g.B + (long long)blockIdx.z * g.sB;
  const bool all_valid = (m0 + wr * 64 + 32) < g.M && (n0 + wc * 64 + 32) < g.N;   // wave-uniform
  if (all_valid)
    gemm_mainloop<TA, TB, FAST, false>(g, smem, gA, gB, m0, n0, kbeg, kend, tid, wr, wc, l31, h, acc);
  else
    gemm_mainloop<TA, TB, FAST, true>(g, smem, gA, gB, m0, n0, kbeg, kend, tid, wr, wc, l31, h, acc);

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool to_slab = g.slab != nullptr;
  float* out = to_slab ? g.slab + (long long)blockIdx.y * g.M * g.N
                       : g.C + (long long)blockIdx.z * g.sC;
  const int ldo = to_slab ? g.N : g.ldc;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + l31;
    if (col >= g.N) continue;
    const float bv = (!to_slab && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < g.M) {
          float* pc = out + (long long)row * ldo + col;
          float v = (g.rowscale ? acc[i][j][r] * g.rowscale[(g.row0 + row) / g.rps] : acc[i][j][r]) + bv;
          if (!to_slab) {
            if (g.flags & VQF_GEMM_ACCUM) v += *pc;
            if (g.flags & VQF_GEMM_RELU) v = fmaxf(v, 0.f);
          }
#if VQF_SPLITK_WT
          // slabs of an in-launch combine are stored WRITE-THROUGH (an agent-scope relaxed atomic store = sc1): no release fence
          if (to_slab && g.cnt) __hip_atomic_store(pc, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else
#endif
          *pc = v;
        }
      }
    }
  }
  if (to_slab && g.cnt) {      // split-K combined in this launch by the tile's last arriver (common.h)
    const VqfSplitkTile st = {g.cnt, g.slab, g.C, g.bias, g.M, g.N, g.ldc, g.flags};
    vqf_splitk_combine<BM, BN, NTHREADS, VQF_SPLITK_WT != 0>(st, bid, (int)gridDim.y, m0, n0, tid, smem);
  }
}

// C = sum_z slab[z] + bias (+C) (relu);  float4 over flattened (M,N) when possible
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N,
                                     float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                     int flags) {
  const long long total = (long long)M * N;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int row = (int)(i / N), col = (int)(i - (long long)row * N);
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += slab[(long long)z * total + i];
    if (bias) v += bias[col];
    float* pc = C + (long long)row * ldc + col;
    if (flags & VQF_GEMM_ACCUM) v += *pc;
    if (flags & VQF_GEMM_RELU) v = fmaxf(v, 0.f);
    *pc = v;
  }
}

template <bool TA, bool TB, bool FAST>
int launch_gemm_impl(const GemmArgs& g, dim3 grid, hipStream_t s, int kid) {
  static VqfDynLdsFlags attr = {};   // 72 KB of dynamic LDS: the attribute is needed on EVERY device this process uses
  if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_kernel<TA, TB, FAST>), SMEM_BYTES, attr)) return e;
  VQF_LAUNCH(kid, (gemm_f32_kernel<TA, TB, FAST>), grid, dim3(NTHREADS), SMEM_BYTES, s, g);
  return vqf_last_error();
}

template <bool TA, bool TB>
int launch_gemm(const GemmArgs& g, dim3 grid, hipStream_t s, int kid) {
  // fast (clamped, unguarded) loaders need vector-aligned operands; K-major operands also need
  // the row extent to be a multiple of 4 and at least 4
  const bool okA = g.vecA && (!TA || (g.M % 4 == 0 && g.M >= 4));
  const bool okB = g.vecB && (!TB || (g.N % 4 == 0 && g.N >= 4));
  if (okA && okB) return launch_gemm_impl<TA, TB, true>(g, grid, s, kid);
  return launch_gemm_impl<TA, TB, false>(g, grid, s, kid);
}

}  // namespace

// Arrival counters of the in-launch split-K combine: a ring of zero words in device memory (one copy per device: a module
// global).  A launch takes `tiles` consecutive words; its last arrivers leave them at zero, so the ring needs no clearing
// between launches, and concurrent launches (two streams) hold different words as long as fewer than VQF_SPLITK_RING tiles are
// in flight.  nullptr: no counters (the caller runs the two-launch form).
// Bounds: a launch takes at most RING / 4 words, so four launches of the largest admitted size may be in flight on
// different streams before the ring wraps onto words still in use; the library's callers hold at most two streams (the compute
// stream and the image projection's side stream).  A launch whose status is not VQF_OK may not have run its last arrivers:
// the launcher clears that launch's words on its stream (vqf_splitk_counters_clear), so a failed launch cannot leave tickets
// behind for the launch that draws the same words 65536 tiles later.
constexpr int VQF_SPLITK_RING = 1 << 16;
__device__ int vqf_splitk_ring[VQF_SPLITK_RING];      // zero-initialised when the code object is loaded
int* vqf_splitk_counters(int tiles) {
  static int* base[64] = {};
  static std::atomic<unsigned> next{0};
  if (tiles <= 0 || tiles > VQF_SPLITK_RING / 4) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!base[dev]) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(vqf_splitk_ring)) != hipSuccess) return nullptr;
    base[dev] = (int*)q;
  }
  unsigned cur = next.load(std::memory_order_relaxed), start;
  do {
    start = (cur % VQF_SPLITK_RING) + (unsigned)tiles <= (unsigned)VQF_SPLITK_RING ? cur % VQF_SPLITK_RING : 0u;   // contiguous
  } while (!next.compare_exchange_weak(cur, start + (unsigned)tiles, std::memory_order_relaxed));
  return base[dev] + start;
}

void vqf_splitk_counters_clear(int* words, int tiles, hipStream_t s) {
  if (words && tiles > 0) (void)hipMemsetAsync(words, 0, (size_t)tiles * sizeof(int), s);
}

// shared with gemm_bf16.hip
int vqf_splitk_reduce(const float* slab, int splits, int M, int N, float* C, int ldc,
                      const float* bias, int flags, hipStream_t s) {
  const long long total = (long long)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  VQF_LAUNCH(KID_SPLITK_REDUCE, splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, slab, splits, M,
             N, C, ldc, bias, flags);
  return vqf_last_error();
}

extern "C" size_t vqf_gemm_f32_ws_bytes(int ta, int tb, int M, int N, int K) {
  return vqf_gemm_f32_big_ws_bytes(ta, tb, M, N, K);
}

static int gemm_tile128(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                        int ldc, const float* bias, int flags, const float* rowscale, int rps, int row0, void* ws,
                        size_t ws_bytes, hipStream_t s);

extern "C" int vqf_gemm_f32_big_rows(int ta, int tb, int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return vqf_gemm_f32_big_rows_impl(ta, tb, M, N, K, 0, (size_t)1 << 40);
}

extern "C" int vqf_gemm_f32(int ta, int tb, int M, int N, int K, const float* A, int lda,
                            const float* B, int ldb, float* C, int ldc, const float* bias,
                            int flags, void* ws, size_t ws_bytes, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc < N)
    return VQF_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0)) {
    int rc = VQF_OK, done = 0;
    // large shapes take the 256x256-tile LDS-DMA kernel (gemm_f32_big.hip) ...
    if (vqf_gemm_f32_big_try(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, nullptr, 1, ws, ws_bytes, s, &rc, &done)) {
      if (rc != VQF_OK || done >= M) return rc;
      // ... mid-size ones its whole-rounds row block only (ta == 0 there): the remaining rows follow below
      // (always the 128x128 kernel, without split-K: it adds a row's k in the same order, so the split product has the bits of the unsplit one)
      A += (size_t)done * lda; C += (size_t)done * ldc; M -= done;
      ws = nullptr; ws_bytes = 0;
    }
    // small-M products (the LSTM's recurrent GEMMs): one tile per wave, no split-K slabs / reduce launch (gemm_f32_wave.hip)
    else if (vqf_gemm_f32_wave_try(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, s, &rc)) return rc;
    // the M = 512 forward projections: one round of 128x80 tiles, no K slices (gemm_f32_n80.hip)
    else if (!ta && !tb && vqf_gemm_f32_n80_try(M, N, K, A, lda, B, ldb, C, ldc, bias, flags, s, &rc)) return rc;
  }
  return gemm_tile128(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, nullptr, 1, 0, ws, ws_bytes, s);
}

extern "C" int vqf_gemm_f32_rowscale(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B,
                                     int ldb, float* C, int ldc, const float* bias, int flags, const float* rowscale,
                                     int rows_per_scale, void* stream) {
  if (!A || !B || !C || !rowscale || rows_per_scale <= 0 || M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc < N)
    return VQF_E_BADARG;
  if (flags & VQF_GEMM_ACCUM) return VQF_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int row0 = 0;
  if (aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0)) {
    int rc = VQF_OK, done = 0;
    if (vqf_gemm_f32_big_try(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, rowscale, rows_per_scale, nullptr, 0, s, &rc,
                             &done)) {
      if (rc != VQF_OK || done >= M) return rc;
      A += (size_t)done * lda; C += (size_t)done * ldc; M -= done; row0 = done;
    }
  }
  return gemm_tile128(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, rowscale, rows_per_scale, row0, nullptr, 0, s);
}

// the 128x128-tile kernel of this file (ws == nullptr: no split-K)
static int gemm_tile128(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                        int ldc, const float* bias, int flags, const float* rowscale, int rps, int row0, void* ws,
                        size_t ws_bytes, hipStream_t s) {
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.slab = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.flags = flags;
  g.sA = g.sB = g.sC = 0;
  g.rowscale = rowscale; g.rps = rps; g.row0 = row0; g.cnt = nullptr;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + BN - 1) / BN;
  g.vecA = aligned16(A) && (lda % 4 == 0);
  g.vecB = aligned16(B) && (ldb % 4 == 0);

  // split-K: only when the output has too few tiles to fill the chip and K is deep.  Two
  // workgroups per CU already keep the matrix pipe busy (the others only cover bubbles), so the
  // wave-quantisation model uses 512 slots; measured: img_conv1d wgrad (640 tiles) 134 TF at 4
  // splits vs 124 TF at 8.
  const long long tiles = (long long)g.tiles_m * g.tiles_n;
  const int ktiles = (K + BK - 1) / BK;
  int splits = 1;
  const int slots = 512;
  // (a chip-filling output with a short K is not split either: 5000 x 2048 x 512, 640 tiles -- two slices + the 123 MB slab
  //  reduce 130 us, unsplit 110, tools/gemm_m512_probe.py)
  // (nor a product that fills most of the CUs once with K <= 512: 3584 x 1024 x 512, 224 tiles -- two slices + reduce 59-61 us,
  //  unsplit 52-53, tools/small_gemm_splitk_ab.py)
  if (ws && tiles < 1024 && ktiles >= 32 && !(tiles >= 512 && K <= 1024) && !(tiles >= 192 && tiles <= 256 && K <= 512)) {
    double best = 1e30;
    for (int sp = 1; sp <= 16; ++sp) {
      if (ktiles / sp < 16) break;
      if ((size_t)sp * M * N * sizeof(float) > ws_bytes) break;
      const double blocks = (double)tiles * sp;
      const double rounds = (double)((long long)((blocks + slots - 1) / slots));
      // time ~ rounds * (per-block work ~ 1/sp) ; small penalty per split for the reduce pass
      const double cost = rounds / sp * (1.0 + 0.01 * sp);
      if (cost < best - 1e-12) { best = cost; splits = sp; }
    }
  }
  int kt_per = (ktiles + splits - 1) / splits;
  g.kchunk = kt_per * BK;
  splits = (K + g.kchunk - 1) / g.kchunk;
  if (splits > 1) {
    g.slab = (float*)ws;
    // option gemm_splitk_fused = 1: combined inside this launch by each tile's last-arriving workgroup.  NOT the default for
    // this kernel: measured (tools/gemm_m512_probe.py, gpurun_out/m512_probe.log) the in-launch combine costs 3-20 us MORE per
    // product than the slab-reduce launch it replaces (512x5000x2048: 139 vs 116 us; write-through slab stores instead of the
    // release fence: 136) -- every slice's release + ticket and the last arrivers' serial slab reads (splits x 64 KB each) sit on
    // the tail of a 100-us launch, the chip-wide reduce kernel takes ~8 us.  The large-tile kernels keep it (a wash there).
    if (vqf_opt(VQF_OPT_GEMM_SPLITK_FUSED, 0) == 1) g.cnt = vqf_splitk_counters((int)tiles);
  }

  dim3 grid((unsigned)tiles, (unsigned)splits);
  const int kid = KID_GEMM_A0B0 + 2 * (ta ? 1 : 0) + (tb ? 1 : 0);
  if (g_vqf_prof_on) vqf_prof_dims(M, N, K);
  vqf_stat_bump(VQF_STAT_GEMM_F32_TILE128);
  int rc;
  if (!ta && !tb) rc = launch_gemm<false, false>(g, grid, s, kid);
  else if (!ta && tb) rc = launch_gemm<false, true>(g, grid, s, kid);
  else if (ta && !tb) rc = launch_gemm<true, false>(g, grid, s, kid);
  else rc = launch_gemm<true, true>(g, grid, s, kid);
  if (rc != VQF_OK) {
    vqf_splitk_counters_clear(g.cnt, (int)tiles, s);
    return rc;
  }
  if (splits > 1 && !g.cnt) rc = vqf_splitk_reduce((const float*)ws, splits, M, N, C, ldc, bias, flags, s);
  return rc;
}

// Batched form (grid.z = batch, no bias, no split-K): the per-sample products of
// hieCoAtten.py:32,38,41,45,48 (affinity, attention-weighted sums).
extern "C" int vqf_gemm_f32_batched(int ta, int tb, int batch, int M, int N, int K, const float* A,
                                    int lda, long long strideA, const float* B, int ldb,
                                    long long strideB, float* C, int ldc, long long strideC,
                                    int flags, void* stream) {
  if (!A || !B || !C || batch <= 0 || batch > 65535 || M <= 0 || N <= 0 || K <= 0 || lda <= 0 ||
      ldb <= 0 || ldc < N)
    return VQF_E_BADARG;
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = nullptr; g.slab = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.flags = flags;
  g.sA = strideA; g.sB = strideB; g.sC = strideC;
  g.rowscale = nullptr; g.rps = 1; g.row0 = 0; g.cnt = nullptr;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + BN - 1) / BN;
  g.vecA = aligned16(A) && (lda % 4 == 0) && (strideA % 4 == 0);
  g.vecB = aligned16(B) && (ldb % 4 == 0) && (strideB % 4 == 0);
  g.kchunk = ((K + BK - 1) / BK) * BK;
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)batch);
  hipStream_t s = (hipStream_t)stream;
  const int kid = KID_GEMM_A0B0 + 2 * (ta ? 1 : 0) + (tb ? 1 : 0);
  vqf_stat_bump(VQF_STAT_GEMM_F32_TILE128);
  if (!ta && !tb) return launch_gemm<false, false>(g, grid, s, kid);
  if (!ta && tb) return launch_gemm<false, true>(g, grid, s, kid);
  if (ta && !tb) return launch_gemm<true, false>(g, grid, s, kid);
  return launch_gemm<true, true>(g, grid, s, kid);
}
