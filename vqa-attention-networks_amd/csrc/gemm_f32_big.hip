// Large-tile fp32 GEMM (gfx950 only): the structure of gemm_bf16_big.hip with fp32 operands and
// v_mfma_f32_32x32x2_f32.
//
//   C[m,n] = sum_k Aop[m,k] * Bop[n,k] (+ bias[n]) (relu)      each operand K-contiguous ((rows, K)) or
//   K-major ((K, rows)), as in gemm_f32.hip.
//
// 256x256 workgroup tile, 8 waves of 64x128 (2x4 MFMA tiles, 128 accumulator registers, 2 waves per
// SIMD), operands staged by LDS-DMA (global_load_lds_dwordx4) in K slabs of 16 into five 32 KB slots
// with four slabs in flight (counted s_waitcnt vmcnt + raw s_barrier).  A slab costs a wave 64 MFMAs
// of 64 cycles, so the copy / barrier / fragment-read overheads that limit the bf16 kernel are a few
// per cent here, and there is no VGPR staging, no ds_write pass and one barrier per 16 k.
//   * K-contiguous operand: [row][16] floats (64-byte rows, byte-identical to the bf16 image), chunk
//     XOR (row >> 2) & 3 on the copy's source address and on the ds_read_b128; a lane's 4 floats feed
//     4 MFMAs: lane (r, h) holds k = 8h + 4ks + e of row r for MFMA (ks, e).
//   * K-major operand: [k][256] floats (1 KB k-rows, a wave copy = one k-row, whole lines).  The rows of the
//     wave's strip are INTERLEAVED over its MFMA tiles: A row 2r+i belongs to row-tile i (B column 4c+j to column-tile
//     j), so ONE ds_read_b64 (ds_read_b128) of a k-row hands lane r the operand value of all 2 (4) tiles: per k-pair
//     and wave 2 LDS instructions feed 8 MFMAs, conflict-free without a swizzle (32 lanes read 256 / 512 contiguous
//     bytes).  The accumulator -> output map follows: tile (i, j), MFMA row rm, MFMA column cm is output row
//     2 rm + i (K-major A) and column 4 cm + j.  (Round 1 read these fragments as 4 ds_read_b32 per tile and kept the
//     weight gradients on the 128x128 kernel.)
//   * K-contiguous B: the same column interleave, by a row permutation on the copy's per-lane source address (LDS row
//     32j + c of a wave strip holds column 4c + j; the loop is unchanged).  Either way a lane's four column tiles are 16
//     contiguous output bytes: the epilogue issues 32 global_store_dwordx4 per lane and tile instead of 128 dword stores.
//   * deterministic split-K for few-tile / long-K shapes (the weight gradient).
//   * three loop forms (gemm_f32_big_kernel's MODE, VQF_GEMM_F32_PP): lockstep (0), the ping-pong loop of
//     gemm_bf16_big.hip (1: 5 % slower with 64-cycle MFMAs) and, the default, lockstep with waves 4-7 half a slab behind
//     waves 0-3 (2).
// Preconditions (else the caller uses gemm_f32.hip): K % 4 == 0 (a last slab shorter than 16 k is zero-filled), M >= 256, N >= 128, >= 1024 workgroups (tiles x
// splits), no accumulate flag, 16-byte aligned bases, lda/ldb % 4 == 0, a K-major operand's row extent % 4 == 0.
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace {

typedef const float __attribute__((address_space(1))) gfloat;

#ifndef VQF_F32BIG_TK
#define VQF_F32BIG_TK 16
#endif
#ifndef VQF_F32BIG_NSLOT
#define VQF_F32BIG_NSLOT (VQF_F32BIG_TK == 16 ? 5 : 2)
#endif
constexpr int TM = 256, TN = 256, TK = VQF_F32BIG_TK, NT = 512;   // 8 waves: 2 (M) x 4 (N), 128 x 64 outputs each
constexpr int ROW_B = TK * 4, CH = ROW_B / 16;         // K-contiguous image: bytes per row, 16-byte chunks per row
constexpr int OP_BYTES = 256 * ROW_B;                  // per operand per slab (either layout)
constexpr int SLOT_BYTES = 2 * OP_BYTES;
constexpr int NSLOT = VQF_F32BIG_NSLOT;                // slab s lives in slot s % NSLOT
constexpr int SMEM_BIG = NSLOT * SLOT_BYTES;
constexpr int NG = OP_BYTES / (NT * 16);               // 2 LDS-DMA instructions per thread per operand per slab
#ifndef VQF_F32BIG_GROUP_M
#define VQF_F32BIG_GROUP_M 8
#endif
constexpr int GROUP_M = VQF_F32BIG_GROUP_M;
#ifndef VQF_F32BIG_DEFAULT_MODE
#define VQF_F32BIG_DEFAULT_MODE 2     // loop form when the option gemm_f32_loop is unset (see k_loop)
#endif
#ifndef VQF_F32BIG_ALTMAP
#define VQF_F32BIG_ALTMAP 1           // wave -> strip map with SIMD partners in different column strips (see gemm_f32_big_kernel)
#endif

struct BigArgs {
  const float* A;
  const float* B;
  float* C;              // output, or the split-K slabs (then ldc = N, no bias / relu)
  const float* bias;
  int M, N, K, lda, ldb, ldc, flags;
  int tiles_m, tiles_n, kchunk, splits;
  int group_m;           // row tiles per group of the tile order (pick_group_m)
  int joint;             // split-K launches: XCD remap over all (split, tile) items jointly (gemm_bf16_big.hip tile_coord has the argument)
  // two-phase work order (see gemm_f32_big_kernel): phase 1 walks the column tiles 0 .. tiles_n_full-1 (all of them, or all
  // but a SHORT last one), phase 2 deals the short edge tiles e = 0 .. tiles_m-1 (column tile tiles_n-1) to the workgroups
  // (edge_w0 + e % edge_wn) % gridDim.x
  int tiles_n_full, edge_w0, edge_wn;
  const float* rowscale;   // epilogue: C = rowscale[row / rps] * acc + bias (vqf_gemm_f32_rowscale), or nullptr
  int rps;
  const float* zeros;      // 16 bytes of zeros in device memory: source of the copies past K in the last slab when K % 16 != 0
  // split-K combined in the launch (common.h vqf_splitk_combine): arrival counters (one per output tile) and the real
  // destination -- C / ldc above then describe the slabs; cnt == nullptr: the caller runs vqf_splitk_reduce
  int* cnt;
  float* Cfinal;
  int ldc_final;
  // STREAM-K TAIL (splits == 1, no edge phase): the first sk_fw tiles (whole rounds of sk_V virtual workers) are whole work
  // items as ever; the K slabs of the remaining sk_tail tiles -- fewer than sk_V: the partial round that would leave most
  // CUs idle -- form ONE sequence of sk_tail x S slabs that the virtual workers share out evenly (sk_q consecutive slabs each,
  // at most two tiles touched).  A fragment's accumulators go to its tile's own slab image (sk_slab + (tile x 3 + part) x 256
  // x 256 floats) and the tile's LAST-arriving fragment sums the 2-3 parts in part order (vqf_splitk_combine, tile-local
  // form).  Workgroup b plays the virtual workers b, b + gridDim.x, ...: the result does not depend on the launch's CU limit.
  int sk_tail, sk_fw, sk_V, sk_q;
  float* sk_slab;
#ifdef VQF_F32BIG_CLOCK
  unsigned long long* dbg;   // diagnostic build only (tools/f32_clock.py): per workgroup and wave half, s_memtime / s_memrealtime stamps
#endif
};

// per-lane global source pointers of the NG copies of one operand slab.
//   K-contiguous: copy i, wave w, lane l -> row i*128 + 16w + (l >> 2), LDS chunk l & 3 (rows clamped to
//   R-1: edge tiles read duplicates of the last row, whose results are never stored).
//   K-major:      copy i, wave w, lane l -> k-row i*8 + w, floats 4l .. 4l+3 of that row (column chunks
//   past R are clamped to the last whole chunk: R % 4 == 0).
// chunk XOR of row r: the 16 lanes of a ds_read_b128 phase must hit 16 distinct 16-byte slots of the 256-byte bank row
__device__ __forceinline__ int swz(int r) { return CH == 4 ? ((r >> 2) & 3) : ((r >> 1) & 7); }

// A K-contiguous B is stored with its columns PERMUTED inside each wave strip of 128 (PERM): LDS row 32 j + c of a strip
// holds column 4 c + j, so that column tile j of a wave is the set of columns = j (mod 4) and a lane's four column tiles
// are 16 contiguous output bytes (a K-major B has this property without a permutation: see FragB).
// `blocked` (bit s = wave strip s of 128 LDS rows): that strip of a PERM operand keeps the IDENTITY order (LDS row 32 j + c =
// column 32 j + c) -- the short edge strips of the last column tile, whose waves then multiply only the column tiles that hold
// live columns (see the kernel).
template <bool T, bool PERM>
__device__ __forceinline__ void init_src(gfloat* (&q)[NG], const float* base, int ld, int r0, int R, int k0,
                                         int wave, int lane, int blocked = 0) {
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    if (!T) {
      constexpr int RPW = 1024 / ROW_B;                        // rows per wave copy (1 KB)
      const int rho = i * (8 * RPW) + wave * RPW + lane / CH;  // LDS row
      const int chunk = (lane % CH) ^ swz(rho);                // source chunk that lands on LDS slot lane % CH
      const bool perm = PERM && !((blocked >> (rho >> 7)) & 1);
      const int row = perm ? (rho & ~127) + 4 * (rho & 31) + ((rho >> 5) & 3) : rho;
      q[i] = (gfloat*)(base + (long long)min(r0 + row, R - 1) * ld + k0 + chunk * 4);
    } else {
      const int k = i * 8 + wave;
      q[i] = (gfloat*)(base + (long long)(k0 + k) * ld + min(r0 + lane * 4, R - 4));
    }
  }
}

template <bool T>
__device__ __forceinline__ void stage_operand(gfloat* (&q)[NG], int ld, char* s, int wave) {
  typedef __attribute__((address_space(3))) char lds_char;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    lds_char* dst = (lds_char*)(s + (i * 8 + wave) * 1024);    // wave-uniform; the DMA adds lane * 16
    __builtin_amdgcn_global_load_lds(q[i], dst, 16, 0, 0);
    q[i] += T ? (long long)TK * ld : TK;
  }
}

// The LAST slab of a K that is not a multiple of 16 (K % 4 == 0: co_att_conv1's K = 1000): copies whose 4 k lie past K take
// the 16 zero bytes instead (both operands, so nothing beyond either matrix is read and 0 * 0 is added).
template <bool T>
__device__ __forceinline__ void stage_operand_tail(gfloat* (&q)[NG], char* s, int wave, int lane, int kvalid, gfloat* zeros) {
  typedef __attribute__((address_space(3))) char lds_char;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    bool ok;
    if (!T) {
      constexpr int RPW = 1024 / ROW_B;
      const int rho = i * (8 * RPW) + wave * RPW + lane / CH;
      ok = 4 * ((lane % CH) ^ swz(rho)) < kvalid;
    } else {
      ok = i * 8 + wave < kvalid;
    }
    lds_char* dst = (lds_char*)(s + (i * 8 + wave) * 1024);
    __builtin_amdgcn_global_load_lds(ok ? q[i] : zeros, dst, 16, 0, 0);
  }
}

// one slab of both operands into a slot; kvalid = the slab's k inside K (TK except in a short last slab)
template <bool TA, bool TB>
__device__ __forceinline__ void stage_slab(const BigArgs& g, gfloat* (&qa)[NG], gfloat* (&qb)[NG], char* slot_base, int wave,
                                           int lane, int kvalid) {
  if (kvalid == TK) {                                  // (wave-uniform)
    stage_operand<TA>(qa, g.lda, slot_base, wave);
    stage_operand<TB>(qb, g.ldb, slot_base + OP_BYTES, wave);
  } else {
    stage_operand_tail<TA>(qa, slot_base, wave, lane, kvalid, (gfloat*)g.zeros);
    stage_operand_tail<TB>(qb, slot_base + OP_BYTES, wave, lane, kvalid, (gfloat*)g.zeros);
  }
}

// Operand values of one k-step (8 k) of a slab for the wave's strip (A: 64 rows = 2 tiles, B: 128 columns = 4 tiles).
// MFMA step (ks, e) of a slab multiplies k = 8ks + e (lanes 0-31) and k = 8ks + 4 + e (lanes 32-63): any fixed pairing
// of the slab's 16 k works as long as both operands use the same one (v_mfma_f32_32x32x2_f32 takes k = 0 from lanes
// 0-31 and k = 1 from lanes 32-63).
//   K-contiguous: one ds_read_b128 per tile: lane (r, h) gets k = 8ks + 4h + e, e = 0..3, of LDS row 32 tile + r.
//   K-major:      one ds_read_b64 (A) / ds_read_b128 (B) per e: lane (r, h) gets k-row 8ks + 4h + e, strip rows
//                 2r, 2r+1 (columns 4r .. 4r+3): element t belongs to tile t (interleaved strip, see the header).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool T>
struct FragA {                                           // v(i, e): value of row-tile i for step e of this k-step
  f32x4 fc[2];                                           // K-contiguous: [tile] (e in the vector)
  f32x2 ft[4];                                           // K-major: [e] (tile in the vector)
  template <int NJ = 4>
  __device__ __forceinline__ void load(const char* s, int strip0, int ks, int lane) {
    if (NJ == 0) return;                                 // a wave without live columns multiplies nothing
    const int r = lane & 31, h = lane >> 5;
    if (T) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        ft[e] = *reinterpret_cast<const f32x2*>(s + (8 * ks + 4 * h + e) * 1024 + (strip0 + 2 * r) * 4);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fc[i] = *reinterpret_cast<const f32x4*>(s + (strip0 + 32 * i + r) * ROW_B + (((2 * ks + h) ^ swz(r)) << 4));
    }
  }
  __device__ __forceinline__ float v(int i, int e) const { return T ? ft[e][i] : fc[i][e]; }
};
template <bool T>
struct FragB {
  f32x4 f[4];                                            // K-contiguous: [tile] (e in the vector); K-major: [e] (tile in the vector)
  template <int NJ = 4>                                  // K-contiguous: only the first NJ column tiles are read
  __device__ __forceinline__ void load(const char* s, int strip0, int ks, int lane) {
    if (NJ == 0) return;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int x = 0; x < 4; ++x)
      if (T || x < NJ)
        f[x] = T ? *reinterpret_cast<const f32x4*>(s + (8 * ks + 4 * h + x) * 1024 + (strip0 + 4 * r) * 4)
                 : *reinterpret_cast<const f32x4*>(s + (strip0 + 32 * x + r) * ROW_B + (((2 * ks + h) ^ swz(r)) << 4));
  }
  __device__ __forceinline__ float v(int j, int e) const { return T ? f[e][j] : f[j][e]; }
};

// accumulator tile (i, j), register e, lane (cm = lane & 31, h = lane >> 5): MFMA row rm = (e & 3) + 8 (e >> 2) + 4h,
// MFMA column cm.  Output row = strip row 2 rm + i (K-major A: interleaved strip) or 32 i + rm; output column = strip
// column 4 cm + j (both B layouts): a lane stores its four column tiles as ONE 16-byte store, 32 per lane and tile
// instead of 128 dword stores (the epilogue is store-issue bound: the 256 KB tile took ~12 slab times).
template <bool TA, bool GUARD_M, bool VEC>
__device__ __forceinline__ void store_tile(const BigArgs& g, float* C, const f32x16 (&acc)[2][4], int row0, int col0,
                                           int lane, bool relu, bool use_bias) {
  const int cm = lane & 31, h = lane >> 5;
  const int col = col0 + 4 * cm;
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = (use_bias && col + j < g.N) ? g.bias[col + j] : 0.f;
  // vmcnt(0) for the biases, once (no wait may sit between the stores) -- as a BUILTIN: hipcc does not see through an asm
  // wait, would carry "bias loads pending" around the persistent tile loop and re-wait with vmcnt(0) at every slab
  __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rm = (e & 3) + 8 * (e >> 2) + 4 * h;
      const int row = row0 + (TA ? 2 * rm + i : 32 * i + rm);
      if (GUARD_M && row >= g.M) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[i][j][e] + bv[j];
        if (relu) v[j] = fmaxf(v[j], 0.f);
      }
      float* cp = C + (long long)row * g.ldc + col;
      if (VEC) {
        if (col < g.N) *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col + j < g.N) cp[j] = v[j];
      }
    }
  }
}

// rowscale epilogue (vqf_gemm_f32_rowscale: F.normalize folded into co_att_conv1): acc *= rowscale[row / rps] ahead of the
// bias / relu / store pass, which stays as it is
template <bool TA>
__device__ __forceinline__ void scale_acc(const BigArgs& g, f32x16 (&acc)[2][4], int row0, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float rs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rm = (e & 3) + 8 * (e >> 2) + 4 * h;
      const int row = min(row0 + (TA ? 2 * rm + i : 32 * i + rm), g.M - 1);
      rs[e] = g.rowscale[row / g.rps];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j][e] *= rs[e];
  }
}

// blocked edge strips: accumulator tile (i, j) of a wave whose strip keeps the identity column order holds output column
// strip column 32 j + cm; only the first nj column tiles were multiplied.  Dword stores: these are the few short edge tiles.
template <bool TA>
__device__ __forceinline__ void store_tile_blocked(const BigArgs& g, float* C, const f32x16 (&acc)[2][4], int row0, int col0,
                                                   int lane, bool relu, bool use_bias, int nj) {
  const int cm = lane & 31, h = lane >> 5;
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = col0 + 32 * j + cm;
    bv[j] = (use_bias && j < nj && col < g.N) ? g.bias[col] : 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0) for the biases, as a builtin (see store_tile)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rm = (e & 3) + 8 * (e >> 2) + 4 * h;
      const int row = row0 + (TA ? 2 * rm + i : 32 * i + rm);
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = col0 + 32 * j + cm;
        if (j < nj && col < g.N) {
          float v = acc[i][j][e] + bv[j];
          if (relu) v = fmaxf(v, 0.f);
          C[(long long)row * g.ldc + col] = v;
        }
      }
    }
  }
}

__device__ __forceinline__ void wait_copies(int later) {     // all but the 4 * later youngest LDS-DMA copies of this wave have landed
  static_assert(2 * NG == 4 || NSLOT == 2, "the vmcnt immediates below assume 4 copies per thread per slab");
  if (later >= 3)      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the 8 k of one k-step for the wave's 2 x NJ MFMA tiles
template <bool TA, bool TB, int NJ>
__device__ __forceinline__ void mma_kstep(const FragA<TA>& fa, const FragB<TB>& fb, f32x16 (&acc)[2][4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.v(i, e), fb.v(j, e), acc[i][j], 0, 0, 0);
}

// K loop of one work item (S slabs), NJ = column tiles this wave multiplies (4; 1 or 2 in a short blocked edge strip; 0 in
// a strip without live columns: such a wave still copies, waits and synchronises with the others).
// MODE 0: lockstep loop (one barrier per slab, fragment reads one k-step ahead of the MFMAs; both waves of a SIMD
//         interleave their MFMAs on the matrix pipe).
// MODE 1: ping-pong loop of gemm_bf16_big.hip (for A/Bs, gemm_f32_loop = 1: with 64-cycle MFMAs it is 5 % slower here).
// MODE 2: the lockstep loop with waves 4-7 HALF A SLAB behind waves 0-3 (the default): both halves still share the
//         matrix pipe of their SIMD, but while one half sits in its per-slab barrier / first fragment reads the other is
//         in the middle of its MFMAs and takes the whole pipe.  One barrier per slab as before: waves 0-3 execute it at
//         the START of slab s (after their vmcnt for slab s), waves 4-7 in the MIDDLE of slab s-1 (after lgkmcnt(0) for
//         all their reads of slab s-1 and their vmcnt for slab s).  Hazards: every wave's copies of slab s are waited for
//         in front of that barrier and every read of slab s comes after it; the slot of slab s-1 is refilled (slab s+4) by
//         waves 0-3 right behind that barrier and by waves 4-7 at their own slab-s start, in both cases after every read
//         of slab s-1 (waves 0-3: consumed by MFMAs issued before the barrier; waves 4-7: the lgkmcnt(0) above).
template <bool TA, bool TB, int MODE, int NJ>
__device__ __forceinline__ void k_loop(const BigArgs& g, char* smem, int& slot, const int S, gfloat* (&qa)[NG],
                                       gfloat* (&qb)[NG], f32x16 (&acc)[2][4], const int wave, const int lane,
                                       const int wr, const int wc, const int late, const int kt) {
  constexpr bool PP = MODE == 1;
  if (MODE == 0 || (MODE == 2 && late == 0)) {
    for (int s = 0; s < S; ++s) {
      wait_copies(min(NSLOT - 2, S - 1 - s));          // my copies of slab s; later slabs stay in flight
      __builtin_amdgcn_s_barrier();
      const char* sA = smem + slot * SLOT_BYTES;
      const char* sB = sA + OP_BYTES;
      FragA<TA> fa[2];                                 // fragment double buffer: reads run one k-step ahead
      FragB<TB> fb[2];
      fa[0].template load<NJ>(sA, wr * 64, 0, lane);
      fb[0].template load<NJ>(sB, wc * 128, 0, lane);
      if (s + NSLOT - 1 < S) {                         // refill the slot of slab s-1 (its address math hides LDS latency)
        const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
        stage_slab<TA, TB>(g, qa, qb, smem + sl * SLOT_BYTES, wave, lane, (s + NSLOT == S) ? kt : TK);
      }
#pragma unroll
      for (int ks = 0; ks < TK / 8; ++ks) {
        if (ks + 1 < TK / 8) {
          fa[(ks + 1) & 1].template load<NJ>(sA, wr * 64, ks + 1, lane);
          fb[(ks + 1) & 1].template load<NJ>(sB, wc * 128, ks + 1, lane);
        }
        mma_kstep<TA, TB, NJ>(fa[ks & 1], fb[ks & 1], acc);
      }
      slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
    }
    if (MODE == 2) __builtin_amdgcn_s_barrier();       // pairs with the mid-slab barrier of waves 4-7 in their last slab
  } else if (MODE == 2) {                              // waves 4-7: half a slab behind
    static_assert(TK == 16, "the staggered loop splits a slab into its two k-steps");
    wait_copies(min(NSLOT - 1, S) - 1);                // my copies of slab 0
    __builtin_amdgcn_s_barrier();                      // pairs with the slab-0 barrier of waves 0-3
    for (int s = 0; s < S; ++s) {
      const char* sA = smem + slot * SLOT_BYTES;
      const char* sB = sA + OP_BYTES;
      FragA<TA> fa[2];
      FragB<TB> fb[2];
      fa[0].template load<NJ>(sA, wr * 64, 0, lane);
      fb[0].template load<NJ>(sB, wc * 128, 0, lane);
      if (s + NSLOT - 1 < S) {                         // slab s+4 into the slot of slab s-1
        const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
        stage_slab<TA, TB>(g, qa, qb, smem + sl * SLOT_BYTES, wave, lane, (s + NSLOT == S) ? kt : TK);
      }
      fa[1].template load<NJ>(sA, wr * 64, 1, lane);
      fb[1].template load<NJ>(sB, wc * 128, 1, lane);
      mma_kstep<TA, TB, NJ>(fa[0], fb[0], acc);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): every read of slab s by this wave has returned
      wait_copies(min(s + NSLOT - 1, S - 1) - (s + 1));   // my copies of slab s+1; later slabs stay in flight
      __builtin_amdgcn_s_barrier();                    // pairs with the slab-(s+1) barrier of waves 0-3 (their final one for s = S-1)
      __builtin_amdgcn_sched_barrier(0);
      mma_kstep<TA, TB, NJ>(fa[1], fb[1], acc);
      slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
    }
  } else {
    static_assert(!PP || (2 * NG == 4 && NSLOT >= 3 && TK == 16), "the ping-pong loop assumes 4 copies per thread per slab and at least 3 slots");
    wait_copies(min(NSLOT - 1, S) - 1);                // my copies of slab 0
    __builtin_amdgcn_s_barrier();                      // #0: every wave's copies of slab 0 have landed
    if (late) __builtin_amdgcn_s_barrier();            // waves 4-7 fall half a slab behind (wave-uniform branch)
    for (int s = 0; s < S; ++s) {
      // ---------------- L(s) ----------------
      __builtin_amdgcn_sched_barrier(0);
      const char* sA = smem + slot * SLOT_BYTES;
      const char* sB = sA + OP_BYTES;
      FragA<TA> fa[2];
      FragB<TB> fb[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        fb[ks].template load<NJ>(sB, wc * 128, ks, lane);
        fa[ks].template load<NJ>(sA, wr * 64, ks, lane);
      }
      if (s + NSLOT - 1 < S) {                         // slab s+4 into the slot of slab s-1
        const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
        stage_slab<TA, TB>(g, qa, qb, smem + sl * SLOT_BYTES, wave, lane, (s + NSLOT == S) ? kt : TK);
      }
      wait_copies(min(s + NSLOT - 1, S - 1) - (s + 1));   // my copies of slab s+1; later slabs stay in flight
      __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // ---------------- M(s) ----------------
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mma_kstep<TA, TB, NJ>(fa[0], fb[0], acc);
      mma_kstep<TA, TB, NJ>(fa[1], fb[1], acc);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
    }
    if (!late) __builtin_amdgcn_s_barrier();           // waves 0-3 absorb the barrier waves 4-7 spent on the stagger
  }
}

// Wave -> strip map: waves 0-3 own the row strips 0-3 of column strip 0, waves 4-7 those of column strip 1, so the two
// waves that share a SIMD (w and w + 4) sit in DIFFERENT column strips: in a short last column tile (N = 5000: 136 live
// columns = 128 + 8) every SIMD then hosts one wave with 4 live column tiles and one with 1, instead of two SIMDs doing
// all of the tile's work while the other two multiply clamped duplicates.
// SK: the stream-K tail (BigArgs::sk_*) is its own instantiation -- compiled into the regular one its work-item logic cost the
// image projection 5 % (15.1 vs 14.3 ms, profiles/r04_big_ab.log)
template <bool TA, bool TB, int MODE, bool SK = false>
__global__ void __launch_bounds__(NT, 2) gemm_f32_big_kernel(const BigArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if VQF_F32BIG_ALTMAP
  const int wr = wave & 3, wc = wave >> 2;             // strip rows wr*64 .. +63, strip columns wc*128 .. +127
#else
  const int wr = wave >> 1, wc = wave & 1;
#endif
  const int late = wave >> 2;                          // waves 4-7: the half that runs behind in the staggered / ping-pong loops

  // ---- PERSISTENT workgroups (one per CU: a tile needs the CU's whole LDS): workgroup b runs work items b, b + gridDim.x,
  // ... of the order below -- the items the hardware dispatcher would have handed the same XCD (gridDim.x % 8 == 0, or a
  // single round).  The slab ring runs on across tiles: once a tile's K loop has ended every wave's LDS reads have
  // returned (they precede the loop's last barrier in every loop form), so each wave issues the NEXT tile's first
  // NSLOT-1 slabs right behind its own output stores, and the next K loop starts on landed data instead of paying a
  // workgroup turn-around (7.8 us per 467-us tile of the image projection: tools/f32_clock.py) plus a cold prologue.
  // ---- work-item order, PHASE 1 (column tiles 0 .. tiles_n_full-1): split index slowest; bijective XCD remap, then groups
  // of GROUP_M row tiles, m fastest.  PHASE 2 (only when the last column tile is short and the host split it off): its
  // tiles_m edge tiles cost a fraction of a full tile (see the wave map above), so they are dealt at the end, starting at
  // the workgroups that drew one full tile fewer in phase 1 (edge_w0, edge_wn: the host picks the deal that minimises
  // the longest workgroup).
  const int G = gridDim.x;
  const int ntiles1 = g.tiles_m * g.tiles_n_full;
  const int F = (SK && g.sk_tail > 0) ? g.sk_fw : ntiles1 * g.splits;     // phase-1 items (stream-K: the whole rounds only)
  const int E = g.tiles_m * (g.tiles_n - g.tiles_n_full);   // phase-2 items
  int cur_w = blockIdx.x, cur_e;
  {
    int r = (int)blockIdx.x - g.edge_w0;
    if (r < 0) r += G;
    cur_e = r < g.edge_wn ? r : E;
  }
  int z = 0, m0 = 0, n0 = 0, kbeg = 0, S = 0, kt = TK, nj = 4;  // current work item (uniform over the workgroup); nj: this wave's live column tiles
  // stream-K tail state: next virtual worker of this workgroup, its remaining slab range, and the current item's fragment
  // descriptor (frag_parts == 0: a whole tile)
  int sk_v = blockIdx.x, sk_it = 0, sk_end = 0;
  int frag_parts = 0, frag_part = 0, frag_tile = 0, frag_k0 = 0, frag_S = 0;
  bool frag_last = false;
#ifdef VQF_F32BIG_CLOCK
  int w_dbg = 0;
#endif
  gfloat* qa[NG];
  gfloat* qb[NG];
  int slot = 0;                                        // ring slot of the current tile's slab 0, then of slab s
  auto next_item = [&]() -> bool {                     // locate the next work item, issue its first NSLOT-1 slabs
    int tm, tn;
    if (SK) frag_parts = 0;
    if (cur_w < F) {
#ifdef VQF_F32BIG_CLOCK
      w_dbg = cur_w;
#endif
      int id;
      if (g.joint) {                                   // split-K: an XCD's run walks whole (split, row group) units of 32 tiles
        const int q8 = F / 8, r8 = F % 8, xcd = cur_w % 8, k = cur_w / 8;
        id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
        z = id / ntiles1;
        id -= z * ntiles1;
      } else {
        z = cur_w / ntiles1;
        id = cur_w % ntiles1;
        const int q8 = ntiles1 / 8, r8 = ntiles1 % 8, xcd = id % 8, k = id / 8;
        id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
      }
      cur_w += G;
      const int per_group = g.group_m * g.tiles_n_full;
      const int grp = id / per_group, in = id % per_group;
      const int gm0 = grp * g.group_m;
      const int gsz = min(g.group_m, g.tiles_m - gm0);
      tm = gm0 + in % gsz; tn = in / gsz;
    } else if (cur_e < E) {
#ifdef VQF_F32BIG_CLOCK
      w_dbg = F + cur_e;
#endif
      z = 0; tm = cur_e; tn = g.tiles_n - 1;
      cur_e += g.edge_wn;
    } else if (SK && g.sk_tail > 0) {
      // stream-K tail: the next fragment of this workgroup's current virtual worker, or the next virtual worker's first one
      const int Stile = (g.K + TK - 1) / TK, T = g.sk_tail * Stile;
      while (sk_it >= sk_end) {
        if (sk_v >= g.sk_V) return false;
        sk_it = min(T, sk_v * g.sk_q);
        sk_end = min(T, sk_it + g.sk_q);
        sk_v += G;
      }
      const int tt = sk_it / Stile, s0 = sk_it - tt * Stile;
      const int len = min(Stile - s0, sk_end - sk_it);
      const int v_first = (tt * Stile) / g.sk_q;                     // the virtual worker that holds the tile's first slab
      frag_part = sk_it / g.sk_q - v_first;
      frag_parts = ((tt + 1) * Stile - 1) / g.sk_q - v_first + 1;
      frag_tile = tt;
      sk_it += len;
      int id = g.sk_fw + tt;
      {
        const int q8 = ntiles1 / 8, r8 = ntiles1 % 8, xcd = id % 8, k = id / 8;
        id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
      }
      const int per_group = g.group_m * g.tiles_n_full;
      const int grp = id / per_group, in = id % per_group;
      const int gm0 = grp * g.group_m;
      const int gsz = min(g.group_m, g.tiles_m - gm0);
      tm = gm0 + in % gsz; tn = in / gsz;
      z = 0;
      frag_k0 = s0 * TK;
      frag_S = len;
      frag_last = (s0 + len == Stile);
    } else {
      return false;
    }
    m0 = tm * TM; n0 = tn * TN;
    kbeg = z * g.kchunk;
    {
      const int klen = min(g.K, kbeg + g.kchunk) - kbeg;
      S = (klen + TK - 1) / TK;                        // slabs of this split
      kt = klen - (S - 1) * TK;                        // k of the last slab (TK, or K % TK: only the last split can be short)
    }
    if (SK && frag_parts > 0) {                           // a stream-K fragment: slabs frag_k0 / TK .. of the tile
      kbeg = frag_k0;
      kt = frag_last ? kt : TK;
      S = frag_S;
    }
    // live columns of the two column strips: a K-contiguous B strip with <= 64 of them is staged in identity order and its
    // waves multiply 1 or 2 column tiles; a strip without any multiplies nothing
    const int live0 = g.N - n0, live1 = g.N - n0 - 128;
    const int blocked = TB ? 0 : ((live0 <= 64 ? 1 : 0) | (live1 <= 64 ? 2 : 0));
    const int live = wc ? live1 : live0;
    nj = live <= 0 ? 0 : (TB || live > 64) ? 4 : (live > 32 ? 2 : 1);
    init_src<TA, false>(qa, g.A, g.lda, m0, g.M, kbeg, wave, lane);
    init_src<TB, true>(qb, g.B, g.ldb, n0, g.N, kbeg, wave, lane, blocked);
    int sl = slot;
#pragma unroll
    for (int p = 0; p < NSLOT - 1; ++p) {
      if (p < S) {
        stage_slab<TA, TB>(g, qa, qb, smem + sl * SLOT_BYTES, wave, lane, (p + 1 == S) ? kt : TK);
      }
      sl = (sl + 1 == NSLOT) ? 0 : sl + 1;
    }
    return true;
  };
  if (!next_item()) return;
  for (;;) {
#ifdef VQF_F32BIG_CLOCK
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#ifdef VQF_F32BIG_CLOCK
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
  // (wave-uniform branches; a K-major B cannot be re-ordered per strip, so it only knows full and empty strips)
  if (nj == 4)                k_loop<TA, TB, MODE, 4>(g, smem, slot, S, qa, qb, acc, wave, lane, wr, wc, late, kt);
  else if (!TB && nj == 2)    k_loop<TA, TB, MODE, 2>(g, smem, slot, S, qa, qb, acc, wave, lane, wr, wc, late, kt);
  else if (!TB && nj == 1)    k_loop<TA, TB, MODE, 1>(g, smem, slot, S, qa, qb, acc, wave, lane, wr, wc, late, kt);
  else                        k_loop<TA, TB, MODE, 0>(g, smem, slot, S, qa, qb, acc, wave, lane, wr, wc, late, kt);

#ifdef VQF_F32BIG_CLOCK
  const unsigned long long c2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
#endif
  if (SK && frag_parts > 1) {
    // a stream-K fragment: the accumulators go to this tile's part image (a full 256 x 256 tile, pitch 256: no edge guards),
    // the tile's last-arriving fragment adds the parts up in part order and applies bias / ReLU (vqf_splitk_combine)
    float* img = g.sk_slab + ((size_t)frag_tile * 3 + frag_part) * (TM * TN);
    BigArgs gl = g;
    gl.M = TM; gl.N = TN; gl.ldc = TN;
    store_tile<TA, false, true>(gl, img, acc, wr * 64, wc * 128, lane, false, false);
    const VqfSplitkTile st = {g.cnt, g.sk_slab + (size_t)frag_tile * 3 * (TM * TN), g.Cfinal, g.bias, g.M, g.N, g.ldc_final,
                              g.flags, (long long)TM * TN, TN};
    vqf_splitk_combine<TM, TN, NT>(st, frag_tile, frag_parts, m0, n0, tid, reinterpret_cast<float*>(smem));
    if (!next_item()) break;
    continue;
  }
  const bool split = g.splits > 1;
  const bool relu = !split && (g.flags & VQF_GEMM_RELU) != 0;
  float* C = split ? g.C + (size_t)z * g.M * g.N : g.C;
  const bool use_bias = !split && g.bias != nullptr;
  const bool vec = (g.N % 4 == 0) && (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                   (!split || (((size_t)g.M * g.N) % 4 == 0));
  const int row0 = m0 + wr * 64, col0 = n0 + wc * 128;
  if (g.rowscale && !split && nj > 0) scale_acc<TA>(g, acc, row0, lane);
  if (nj == 4) {
    if (vec) {
      if (m0 + TM <= g.M) store_tile<TA, false, true>(g, C, acc, row0, col0, lane, relu, use_bias);
      else                store_tile<TA, true, true>(g, C, acc, row0, col0, lane, relu, use_bias);
    } else {
      store_tile<TA, true, false>(g, C, acc, row0, col0, lane, relu, use_bias);
    }
  } else if (nj > 0) {
    store_tile_blocked<TA>(g, C, acc, row0, col0, lane, relu, use_bias, nj);
  }
  if (split && g.cnt) {
    // The K slices of a tile are combined in this launch by the tile's last arriver.  Every wave has left the K loop when the
    // combine's first barrier releases (each wave runs the same number of barriers per item in every loop form), the ring is
    // idle until next_item() below issues the next item's slabs: its first bytes serve as the ticket's broadcast word.
    const VqfSplitkTile st = {g.cnt, g.C, g.Cfinal, g.bias, g.M, g.N, g.ldc_final, g.flags};
    vqf_splitk_combine<TM, TN, NT>(st, (m0 / TM) * g.tiles_n + n0 / TN, g.splits, m0, n0, tid, reinterpret_cast<float*>(smem));
  }
#ifdef VQF_F32BIG_CLOCK
  if (g.dbg && lane == 0 && (wave & 3) == 0) {         // one record per wave half: [entry, loop start, loop end, stores issued] cycles,
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // [entry, loop end] 100 MHz ticks, slabs; stores landed at the last stamp
    const unsigned long long c3 = __builtin_amdgcn_s_memtime();
    unsigned long long* d = g.dbg + ((size_t)w_dbg * 2 + late) * 8;
    d[0] = c1 - c0; d[1] = c2 - c1; d[2] = c3 - c2; d[3] = r2 - r0; d[4] = c2 - c0; d[5] = (unsigned long long)S;
    d[6] = r0; d[7] = c0;
  }
#endif
  if (!next_item()) break;
  }
}

// split count that minimises (rounds of 256 one-per-CU workgroups) x (time of one workgroup) + slab traffic, in
// microseconds: a 256x256 tile costs a CU 2*256*256 / (157.3e12 / 256) = 0.213 us per unit of k; every split writes
// and the reduce reads an M x N fp32 slab (~4 TB/s).
int pick_splits(int tiles, int K, int M, int N, size_t ws_bytes) {
  if (tiles >= 768) return 1;
  int best = 1;
  double best_cost = 1e30;
  for (int sp = 1; sp <= 32; ++sp) {      // (up to 32: HieCoAtten's 1024 x 512 x 50176 weight gradient is 8 tiles -- 32 slices of 49 slabs fill a round)
    if (sp > 1 && (size_t)sp * M * N * sizeof(float) > ws_bytes) break;
    if (sp > 1 && K / sp < 16 * TK) break;
    const long long blocks = (long long)tiles * sp;
    const double rounds = (double)((blocks + 255) / 256);
    const int slabs = (K / TK + sp - 1) / sp;
    const double cost = rounds * slabs * TK * 0.213 + (sp > 1 ? sp * (double)M * N * 8.0 / 4.0e6 : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
  }
  return best;
}

// Row tiles per group of the tile order.  An XCD (private L2) works on a contiguous run of the order: the ~32 tiles
// resident at a time when one split covers the chip, or its tiles / 8 share of EACH split for the few-tile split-K
// launches (the image projection's weight gradient: 160 tiles -> 20 per XCD and split).  A run of `share` tiles
// walked row-tile-fastest in groups of gm covers gm A panels + share / gm B panels: pick the gm in 4..8 that minimises
// that, preferring one that divides the run (5 x 4 for the weight gradient: 12.3 -> 10.0 GB beyond L2, PMC).
int pick_group_m(int tiles_m, int tiles_n, int splits) {
  const int ntiles = tiles_m * tiles_n;
  const int share = (splits > 1 && ntiles / 8 < 32) ? (ntiles / 8 > 0 ? ntiles / 8 : 1) : 32;
  int best = GROUP_M, best_cost = 1 << 30;
  for (int gm = 8; gm >= 4; --gm) {
    const int cost = 4 * (gm + (share + gm - 1) / gm) + (share % gm ? 2 : 0) + (tiles_m % gm ? 1 : 0);
    if (cost < best_cost) { best_cost = cost; best = gm; }
  }
  return best;
}

// Phase 2 of the work order: which workgroups take the short edge tiles (kernel header).  Costs in eighths of a full tile:
// a full tile keeps every SIMD busy for 8 units, an edge tile for nj0 + nj1 (one wave of each column strip per SIMD).
// After phase 1 the first h = F % G workgroups hold one full tile more than the others ("light").  Two deals are priced
// and the one with the shorter longest workgroup is taken: the light workgroups only, or everybody starting at the first
// light one.  One workgroup per item (gridDim = F + E): workgroup F + e simply takes edge e.
void deal_edge_tiles(BigArgs& g, int G) {
  const int F = g.tiles_m * g.tiles_n_full * g.splits, E = g.tiles_m * (g.tiles_n - g.tiles_n_full);
  g.edge_w0 = 0; g.edge_wn = G > 0 ? G : 1;
  if (E == 0) return;
  if (G >= F + E) { g.edge_w0 = F; g.edge_wn = E; return; }
  const int rem = g.N - (g.tiles_n - 1) * TN;          // live columns of the edge tile (1 .. 192 here)
  auto njf = [](int live) { return live <= 0 ? 0 : live > 64 ? 4 : live > 32 ? 2 : 1; };
  const long long c8 = njf(rem) + njf(rem - 128);
  const int h = F % G, light = G - h;
  const long long a = F / G;
  const long long maxA = std::max(h > 0 ? (a + 1) * 8 : 0LL, a * 8 + ((E + light - 1) / light) * c8);
  auto cnt = [&](int r) { return E > r ? (long long)((E - r - 1) / G + 1) : 0LL; };
  const long long maxB = std::max(a * 8 + cnt(0) * c8, h > 0 ? (a + 1) * 8 + cnt(light) * c8 : 0LL);
  g.edge_w0 = h;
  g.edge_wn = (maxA <= maxB) ? light : G;
}

template <bool TA, bool TB>
int launch(const BigArgs& g_in, hipStream_t s) {
  // > 64 KB of dynamic LDS needs the attribute, once per device and instantiation (common.h)
  static VqfDynLdsFlags attr = {}, attr_pp = {};
  const int lo = vqf_opt(VQF_OPT_GEMM_F32_LOOP, VQF_F32BIG_DEFAULT_MODE);   // A/B switch: 0 lockstep, 1 ping-pong, 2 staggered lockstep
  const int mode = (lo >= 0 && lo <= 2) ? lo : VQF_F32BIG_DEFAULT_MODE;
  const int kid = KID_GEMM_A0B0 + 2 * (TA ? 1 : 0) + (TB ? 1 : 0);
  // persistent workgroups: one per CU, each walking its share of the work items (VQF_OPT_GEMM_F32_PERSIST = 0: one
  // workgroup per item, as in round 1); VQF_OPT_GEMM_CU_LIMIT leaves part of the chip to other streams
  const int total = g_in.tiles_m * g_in.tiles_n * g_in.splits;
  int nwg = total;
  if (vqf_opt(VQF_OPT_GEMM_F32_PERSIST, 1) != 0) {
    int cus = vqf_cu_count() & ~7;                     // a multiple of 8 keeps every workgroup's items on its own XCD
    const int lim = vqf_opt(VQF_OPT_GEMM_CU_LIMIT, 0) & ~7;
    if (lim >= 8 && lim < cus) cus = lim;
    if (cus >= 8 && total > cus) nwg = cus;
  }
  const dim3 grid(nwg);
  BigArgs g = g_in;
  deal_edge_tiles(g, nwg);
  if (g.sk_tail > 0) {                                 // stream-K tail: one loop form (the default one)
    static VqfDynLdsFlags attr_sk = {};
    if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_big_kernel<TA, TB, VQF_F32BIG_DEFAULT_MODE, true>), SMEM_BIG, attr_sk))
      return e;
    VQF_LAUNCH(kid, (gemm_f32_big_kernel<TA, TB, VQF_F32BIG_DEFAULT_MODE, true>), grid, dim3(NT), SMEM_BIG, s, g);
    return vqf_last_error();
  }
  if (mode == 1) {
    if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_big_kernel<TA, TB, 1>), SMEM_BIG, attr_pp)) return e;
    VQF_LAUNCH(kid, (gemm_f32_big_kernel<TA, TB, 1>), grid, dim3(NT), SMEM_BIG, s, g);
    return vqf_last_error();
  }
  if (mode == 2) {
    static VqfDynLdsFlags attr_st = {};
    if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_big_kernel<TA, TB, 2>), SMEM_BIG, attr_st)) return e;
    VQF_LAUNCH(kid, (gemm_f32_big_kernel<TA, TB, 2>), grid, dim3(NT), SMEM_BIG, s, g);
    return vqf_last_error();
  }
  if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_big_kernel<TA, TB, 0>), SMEM_BIG, attr)) return e;
  VQF_LAUNCH(kid, (gemm_f32_big_kernel<TA, TB, 0>), grid, dim3(NT), SMEM_BIG, s, g);
  return vqf_last_error();
}

#ifndef F32BIG_MIN_K
#define F32BIG_MIN_K 1536
#endif
#ifndef F32BIG_WGRAD_MIN_BLOCKS
#define F32BIG_WGRAD_MIN_BLOCKS 256   // co_att_conv1's wgrad: 16 tiles x 16 splits = one full round, 1.71 -> 1.52 ms
#endif
bool big_applies(int ta, int tb, int M, int N, int K, int flags, size_t ws_bytes) {
  const int sw = vqf_opt(VQF_OPT_GEMM_F32_BIG, 1);   // A/B switch: 0 selects the 128x128 kernel everywhere, 2 this one wherever it CAN run
  if (sw == 0 || (K % 4) || K < 4 * TK || M < TM || N < 128 || (flags & VQF_GEMM_ACCUM)) return false;
  if (ta && (M % 4)) return false;
  if (tb && (N % 4)) return false;
  if (sw == 2) return true;
  // Only the large projections: a workgroup that needs a whole CU's LDS starts when the CU has drained, which
  // costs mid-size launches more than the kernel gains (HieCoAtten, 392-tile GEMMs: step 5.57 -> 5.77 ms) ...
  const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  // (and a K long enough to amortise the tile's prologue / 256 KB epilogue: at K = 1024 the 128x128 kernel is 5 % faster)
  if (tiles >= 1024) return K >= F32BIG_MIN_K;
  // ... and the deep-K weight gradients (both operands K-major, few tiles, K = N*L): img_conv1d's 20 x 8 tiles x 8
  // splits of 784 slabs each, co_att_conv1's 4 x 4 tiles x 16 splits (one workgroup per CU, 392 slabs each).
  if (ta && tb && K >= 16384 && tiles * pick_splits(tiles, K, M, N, ws_bytes) >= F32BIG_WGRAD_MIN_BLOCKS) return true;   // needs its slabs
  // ... and the LSTM's recurrent weight gradient (4096 x 1024 x 7168: 64 tiles x 4 splits; 447 us against 483-522 on the 128x128
  // kernel, tools/gemm_m512_probe.py) -- K-major products with 4096 <= K < 16384 whose splits fill a round and whose last column
  // tile is not mostly dead (N = 300, the input-weight gradient, stays: 225 vs 204 us)
  if (ta && tb && K >= 4096 && ((N + TN - 1) / TN) * TN * 100LL <= (long long)N * 107 &&
      tiles * pick_splits(tiles, K, M, N, ws_bytes) >= F32BIG_WGRAD_MIN_BLOCKS)
    return true;
  return false;
}

// Mid-size shapes (co_att_conv1 forward / dgrad: 1568 tiles = 6.125 rounds of 256 CUs; HieCoAtten's 392-tile products: 1.53):
// on whole rounds this kernel beats the 128x128 one by 8-9 % also at K = 512 .. 1024 (142 vs 131 TF, tools/gemm_mid_probe.py),
// but a last partial round costs it a whole one.  So such a product is SPLIT BY ROWS: the first r row tiles -- the largest r
// whose r x tiles_n tiles fill whole rounds to within 4 % -- run here, the remaining rows go back to the caller (128x128 /
// per-wave kernels).  Row blocks are independent, every output element keeps one fixed summation order.  Returns the rows
// taken (a multiple of 256), 0 = no split.  Rounds are counted on ALL CUs whatever gemm_cu_limit says, so the split -- and
// with it every bit of the result -- does not depend on that option.
int whole_round_rows(int ta, int tb, int M, int N, int K, int flags) {
  if (vqf_opt(VQF_OPT_GEMM_F32_BIG, 1) == 0 || vqf_opt(VQF_OPT_GEMM_F32_ROUNDS, 1) == 0) return 0;
  if (ta || (flags & VQF_GEMM_ACCUM) || (K % 4) || K < 256 || N < 128 || (tb && (N % 4))) return 0;
  const int tn = (N + TN - 1) / TN;
  if ((long long)tn * TN * 100 > (long long)N * 107) return 0;       // a ragged last column tile: > 7 % of the MFMAs on dead columns
  const int cus = vqf_cu_count() & ~7;
  if (cus < 8) return 0;
  for (int r = M / TM; r >= 1 && r * tn >= cus; --r) {
    const int tiles = r * tn, rounds = (tiles + cus - 1) / cus;
    if ((rounds * cus - tiles) * 25 <= rounds * cus) return r * TM;
  }
  return 0;
}

// STREAM-K TAIL.  A mid-size product whose tile count is a whole number of rounds plus a substantial partial round (HieCoAtten's
// 50176 x 512 products: 392 tiles = 1.53 rounds of 256 CUs) either runs that partial round with most CUs idle or hands its
// rows to the 128x128 kernel (whole_round_rows above: 544 tiles on 256 CUs, 92 TF).  Instead the K slabs of the partial round's
// tiles are shared out evenly over the CUs (kernel header, BigArgs::sk_*): 1.53 rounds take 1.53 tile times.  Returns the
// number of tail tiles (0 = not this way).  Conditions: persistent launch form, no rowscale epilogue, whole column tiles
// (N % 256 == 0), K >= 512, one to four rounds, a tail between 1/8 and 24/25 of a round, scratch for 3 part images per tail tile.
int streamk_tail(int ta, int tb, int M, int N, int K, int flags, size_t ws_bytes, bool rowscale) {
  // OPT-IN (option gemm_f32_streamk = 1), measured and not the default: HieCoAtten's img_emb product 0.94 -> 0.92 ms, the input
  // gradient of its concatenated layers 0.478 -> 0.490 (gpurun_out/r04: tools/gemm_census.py).  A fragment of ~68 slabs pays the
  // tile epilogue (~9 slab times of stores, now into its part image) and its share of the combine (release + ticket; the last
  // arriver reads 2-3 x 256 KB) on top: ~0.32 ms for the 0.25 ms of work the even shares would take -- what the whole-rounds row
  // split (whole_round_rows) gets from the 128x128 kernel for the same rows.  Kept: correct, deterministic, tested.
  if (vqf_opt(VQF_OPT_GEMM_F32_BIG, 1) == 0 || vqf_opt(VQF_OPT_GEMM_F32_STREAMK, 0) != 1) return 0;
  if (vqf_opt(VQF_OPT_GEMM_F32_PERSIST, 1) == 0) return 0;
  if (rowscale || (flags & VQF_GEMM_ACCUM) || (K % 4) || K < 512 || M < TM || (N % TN) || (ta && (M % 4))) return 0;
  const int V = vqf_cu_count() & ~7;
  if (V < 8) return 0;
  const int tiles = ((M + TM - 1) / TM) * (N / TN);
  if (tiles < V || tiles >= 4 * V) return 0;
  const int tail = tiles % V;
  if (tail * 8 < V || tail * 25 > V * 24) return 0;
  if ((size_t)tail * 3 * TM * TN * sizeof(float) > ws_bytes) return 0;
  return tail;
}

// 16 zero bytes in device memory (BigArgs::zeros), per device
__device__ __attribute__((aligned(16))) float vqf_f32big_zeros[4] = {0.f, 0.f, 0.f, 0.f};
const float* zeros16() {
  static const float* ptr[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!ptr[dev]) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(vqf_f32big_zeros)) != hipSuccess) return nullptr;
    ptr[dev] = (const float*)q;
  }
  return ptr[dev];
}

}  // namespace

// scratch the big kernel would like for this shape (split-K slabs); 0 when it does not apply
size_t vqf_gemm_f32_big_ws_bytes(int ta, int tb, int M, int N, int K) {
  if (!big_applies(ta, tb, M, N, K, 0, (size_t)1 << 40)) return 0;
  const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  const int sp = pick_splits(tiles, K, M, N, (size_t)1 << 40);
  return sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
}

// rows of a (ta, tb, M, N, K) product that the large-tile kernel takes: M (all of it), 0 (none), or the whole-rounds row
// block of a mid-size shape (the rest runs on the 128x128 / per-wave kernels)
int vqf_gemm_f32_big_rows_impl(int ta, int tb, int M, int N, int K, int flags, size_t ws_bytes) {
  if (big_applies(ta, tb, M, N, K, flags, ws_bytes)) return M;
  if (streamk_tail(ta, tb, M, N, K, flags, ws_bytes, false) > 0) return M;
  return whole_round_rows(ta, tb, M, N, K, flags);
}

// 0 = this kernel does not apply (caller falls back to gemm_f32.hip), 1 = launched (rc holds the status) on the first
// *rows_done rows (M, or the whole-rounds block of a mid-size shape: the caller computes the remaining rows)
int vqf_gemm_f32_big_try(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                          int ldc, const float* bias, int flags, const float* rowscale, int rps, void* ws, size_t ws_bytes,
                          hipStream_t s, int* rc, int* rows_done) {
  const size_t ws_ok = (ws && aligned16(ws)) ? ws_bytes : 0;
  const bool whole = big_applies(ta, tb, M, N, K, flags, ws_ok);
  const int sk_tail = whole ? 0 : streamk_tail(ta, tb, M, N, K, flags, ws_ok, rowscale != nullptr);
  const int Mb = (whole || sk_tail > 0) ? M : whole_round_rows(ta, tb, M, N, K, flags);
  if (Mb <= 0) return 0;
  BigArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = Mb; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.flags = flags;
  g.rowscale = rowscale; g.rps = rps > 0 ? rps : 1;
  g.zeros = nullptr;
  if (K % TK) {
    g.zeros = zeros16();
    if (!g.zeros) return 0;
  }
  g.tiles_m = (Mb + TM - 1) / TM;
  g.tiles_n = (N + TN - 1) / TN;
  const int tiles = g.tiles_m * g.tiles_n;
  int splits = whole ? pick_splits(tiles, K, Mb, N, ws_ok) : 1;
  if (splits > 1 && (rowscale || (K % TK))) return 0;        // (neither occurs: split-K shapes are the deep-K weight gradients)
  const int slabs = (K + TK - 1) / TK;
  const int per = (slabs + splits - 1) / splits;
  g.kchunk = per * TK;
  splits = (K + g.kchunk - 1) / g.kchunk;
  g.splits = splits;
  g.cnt = nullptr; g.Cfinal = C; g.ldc_final = ldc;
  g.sk_tail = 0; g.sk_fw = 0; g.sk_V = 0; g.sk_q = 0; g.sk_slab = nullptr;
  if (sk_tail > 0) {
    // (the stream-K tail only exists for products that are NOT split over K: `splits` is 1 on this path by construction -- see
    //  its initialisation above --, so the split-K branch below never draws a second set of counters over g.cnt)
    if (splits != 1) return 0;
    const int V = vqf_cu_count() & ~7;
    g.cnt = vqf_splitk_counters(sk_tail);
    if (g.cnt) {
      g.sk_tail = sk_tail; g.sk_V = V; g.sk_fw = tiles - sk_tail;
      g.sk_q = (int)(((long long)sk_tail * slabs + V - 1) / V);
      g.sk_slab = (float*)ws;
    }
  }
  if (splits > 1) {
    g.C = (float*)ws; g.ldc = N;
    // combined in the launch when there are enough tiles for the last arrivers to read their slabs side by side (each reads
    // splits x 256 KB at one CU's rate: 16 tiles x 16 splits would take longer than the chip-wide reduce launch it replaces)
    if (tiles >= 64 && vqf_opt(VQF_OPT_GEMM_SPLITK_FUSED, 1) != 0) g.cnt = vqf_splitk_counters(tiles);
  }
  // a SHORT last column tile of a K-contiguous B (N = 5000: 136 of 256 columns) is split off into phase 2 of the work order:
  // its waves multiply only the column tiles that hold live columns (kernel header)
  const int rem = N % TN;
  const bool edge_opt = !tb && splits == 1 && g.tiles_n > 1 && rem > 0 && rem <= 192 && vqf_opt(VQF_OPT_GEMM_F32_EDGE, 1) != 0;
  g.tiles_n_full = edge_opt ? g.tiles_n - 1 : g.tiles_n;
  g.edge_w0 = 0; g.edge_wn = 1;
  g.group_m = pick_group_m(g.tiles_m, g.tiles_n_full, splits);
  g.joint = 0;
  if (splits > 1 && g.sk_tail == 0 && g.tiles_n <= 32 && 32 % g.tiles_n == 0 && vqf_opt(VQF_OPT_GEMM_SPLITK_ORDER, 0) == 1) {
    g.joint = 1;                                        // an XCD's 32 CUs take group_m row tiles x all column tiles of one split
    g.group_m = std::min(g.tiles_m, 32 / g.tiles_n);
  }
#ifdef VQF_F32BIG_CLOCK
  g.dbg = (splits == 1 && ws && ws_bytes >= (size_t)tiles * 2 * 8 * 8) ? (unsigned long long*)ws : nullptr;
#endif
  vqf_prof_dims(Mb, N, K);
  vqf_stat_bump(VQF_STAT_GEMM_F32_BIG);
  if (ta) *rc = tb ? launch<true, true>(g, s) : launch<true, false>(g, s);
  else    *rc = tb ? launch<false, true>(g, s) : launch<false, false>(g, s);
  if (*rc != VQF_OK) vqf_splitk_counters_clear(g.cnt, g.sk_tail > 0 ? g.sk_tail : tiles, s);
  if (*rc == VQF_OK && splits > 1 && !g.cnt) *rc = vqf_splitk_reduce((const float*)ws, splits, Mb, N, C, ldc, bias, flags, s);
  *rows_done = Mb;
  return 1;
}
