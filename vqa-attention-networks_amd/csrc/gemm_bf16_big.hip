// Large-tile bf16 GEMM for the projection shapes that dominate the bf16 mode (gfx950 only):
//
//   C[m,n] (fp32) = sum_k Aop[m,k] * Bop[n,k] (+ bias[n]) (relu)      A, B stored as bf16, each either
//   K-contiguous ((rows, K), "ta/tb = 0") or K-major ((K, rows), "ta/tb = 1"), as in gemm_bf16.hip.
//
// Why a second kernel: gemm_bf16.hip keeps the fp32 kernel's decomposition (128x128 block, four
// 64x64 wave tiles).  With v_mfma_f32_32x32x16_bf16 a 64x64 wave tile needs 4 KB of LDS reads per
// 128 MFMA cycles = 32 B/clk per wave, i.e. the CU's whole 128 B/clk of LDS bandwidth for its four
// SIMDs: that kernel is LDS-bound at ~25 % of the bf16 MFMA peak (675 TFLOP/s on the image
// projection).  Here a workgroup owns a 256x256 tile, 8 waves (2 x 4) own 128x64 each (4x2 MFMA
// tiles, 128 accumulator registers, 2 waves per SIMD): 6 KB of fragment reads per 256 MFMA cycles.
//   * staging: global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass) in K slabs of 32;
//     five 32 KB LDS slots (all 160 KB), four slabs in flight.  A wave waits for ITS copies of slab s
//     with a counted s_waitcnt vmcnt (later slabs stay in flight), a raw s_barrier then covers the
//     other waves' copies and everybody's reads of slab s-1, whose slot is re-filled right behind it
//     (a plain __syncthreads() would drain the copies with vmcnt(0) at every slab).
//   * K-contiguous operand: LDS image [row][32] bf16 (64-byte rows, unpadded because the LDS-DMA
//     destination is wave-uniform base + lane*16); bank conflicts are removed by XOR-ing the 16-byte
//     chunk index with (row >> 2) & 3 on the per-lane global SOURCE address and again on the fragment
//     read (ds_read_b128): the 16 lanes of a read phase hit the 16 distinct slots of the bank row.
//   * K-major operand (both operands of the weight gradient): LDS image [k][256] bf16 (512-byte k-rows,
//     every copy instruction moves whole 128-byte lines) read with ds_read_b64_tr_b16; the 64-byte
//     granule index is XOR-ed with k & 3 so that the 4 k-rows of a transposing read land on
//     different bank quarters.
//   * deterministic split-K (fixed-order slab reduction, vqf_splitk_reduce) when the tile count alone
//     cannot fill the chip (the weight gradient: 20 x 8 tiles, K = 100352).
// History of this file's loops on the image projection (M=100352, N=5000, K=2048, random operands, fp32 output):
//   round 1, gemm_bf16_big_kernel (all 8 waves in lockstep: barrier -> 12 fragment reads -> 16 MFMAs): 675 -> 876 TFLOP/s over the
//     128x128 kernel; the matrix pipes were 44 % busy (the two waves of a SIMD read together, then multiplied together).
//   round 2, gemm_bf16_pp_kernel / gemm_bf16_pp16_kernel (the default): ping-pong halves -- waves 4-7 half a slab behind waves 0-3,
//     one multiplies while its SIMD partner loads -- and v_mfma_f32_16x16x32_bf16 with one ds_read_b128 per fragment for
//     K-contiguous operands: 1081 TFLOP/s with fp32 output, 1197-1230 with bf16 output (0.48-0.49 of the 2.5 PFLOP/s peak), weight
//     gradient 1113-1145; PMC (profiles/r03_pmc_gemm.txt): MFMA pipe 67 % / 58 % busy at the 1.5-1.6 GHz the chip holds under
//     this load; the loop is bound by the issue cost of the LDS-DMA copies (s_memtime stamps), not by idle pipes.
//   round 3: nothing changed in the loops; the persistent launch takes a CU limit (library option gemm_cu_limit) so that config 3
//     can run these GEMMs beside the LSTM recursion.
// Preconditions (else the caller falls back to gemm_bf16.hip): K % 32 == 0, M >= 256, N >= 128, no
// accumulate flag, a K-major operand's row extent % 8 == 0; 16-byte aligned bases, lda/ldb % 8 == 0.
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;
typedef const bf16_t __attribute__((address_space(1))) gbf16;

constexpr int TM = 256, TN = 256, TK = 32, NT = 512;   // 8 waves: 2 (M) x 4 (N), 128 x 64 outputs each
constexpr int OP_BYTES = 256 * TK * 2;                 // 16 KB per operand per slab (either layout)
constexpr int SLOT_BYTES = 2 * OP_BYTES;               // 32 KB
constexpr int NSLOT = 5;                               // slab s lives in slot s % 5: all 160 KB of LDS
constexpr int SMEM_BIG = NSLOT * SLOT_BYTES;
constexpr int NG = OP_BYTES / (NT * 16);               // 2 LDS-DMA instructions per thread per operand per slab
constexpr int GROUP_M = 8;
// ping-pong loop knobs (compile-time; tools/build_variant.sh for A/Bs)
#ifndef VQF_PP_TAIL
#define VQF_PP_TAIL 0          // MFMAs of a slab issued AFTER the barrier that ends M: the partner's MFMAs start behind them
#endif
#ifndef VQF_PP_GLDS_M
#define VQF_PP_GLDS_M 0        // 1: the refill copies are issued inside M (after the first MFMAs) instead of in L
#endif

struct BigArgs {
  const bf16_t* A;
  const bf16_t* B;
  float* C;              // output, or the split-K slabs (then ldc = N, no bias / relu)
  const float* bias;
  const float* rowscale; // (0,0) layout, 16x16x32 kernel only: C = rowscale[row / rps] * acc + bias (vqf_gemm_bf16_rowscale), or nullptr
  int rps;
  int M, N, K, lda, ldb, ldc, flags;
  int tiles_m, tiles_n, kchunk, splits;
  int group_m;           // row tiles per group of the tile order (pick_group_m)
  int joint;             // split-K launches: the XCD remap runs over ALL (split, tile) items jointly (tile_coord)
  // split-K combined in the launch (common.h vqf_splitk_combine): arrival counters (one per output tile) and the real
  // destination -- C / ldc above then describe the slabs; cnt == nullptr: the caller runs vqf_splitk_reduce
  int* cnt;
  float* Cfinal;
  int ldc_final;
#ifdef VQF_PP_STAMPS
  unsigned long long* dbg;   // diagnostic build only (tools/pp_stamps.py): per workgroup and wave, 8 cycle sums
#endif
};

#ifdef VQF_PP_STAMPS
// s_memtime stamp, fenced so that the segments hold what they are named for (cdna_hip_programming.md section 7)
#define VQF_STAMP(t)                                                                         \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
#else
#define VQF_STAMP(t) do { } while (0)
#endif

// per-lane global source pointers of the NG copies of one operand slab.
//   K-contiguous: copy i, wave w, lane l -> row i*128 + 16w + (l >> 2), LDS chunk l & 3 (rows clamped to
//   R-1: edge tiles read duplicates of the last row, whose results are never stored).
//   K-major:      copy i, wave w, lane l -> k-row i*16 + 2w + (l >> 5), LDS chunk l & 31 of that row
//   (column chunks past R are clamped to the last whole chunk: R % 8 == 0).
template <bool T>
__device__ __forceinline__ void init_src(gbf16* (&q)[NG], const bf16_t* base, int ld, int r0, int R, int k0,
                                         int wave, int lane) {
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    if (!T) {
      const int row = i * 128 + wave * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);         // source chunk that lands on LDS slot lane & 3
      q[i] = (gbf16*)(base + (long long)min(r0 + row, R - 1) * ld + k0 + chunk * 8);
    } else {
      const int k = i * 16 + wave * 2 + (lane >> 5);
      const int c = lane & 31;
      const int chunk = (((c >> 2) ^ (k & 3)) << 2) | (c & 3); // 64-byte granule XOR k & 3
      q[i] = (gbf16*)(base + (long long)(k0 + k) * ld + min(r0 + chunk * 8, R - 8));
    }
  }
}

template <bool T>
__device__ __forceinline__ void stage_operand(gbf16* (&q)[NG], int ld, char* s, int wave) {
  typedef __attribute__((address_space(3))) char lds_char;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    lds_char* dst = (lds_char*)(s + (i * 8 + wave) * 1024);    // wave-uniform; the DMA adds lane * 16
    __builtin_amdgcn_global_load_lds(q[i], dst, 16, 0, 0);
    q[i] += T ? (long long)TK * ld : TK;
  }
}

// operand fragment of v_mfma_f32_32x32x16_bf16 for rows row0 .. row0+31, k = 16 ks .. +15 of a slab:
// lane (r, h) holds k = 8h .. 8h+7 of row r.
template <bool T>
__device__ __forceinline__ bf16x8 read_frag(const char* s, int row0, int ks, int lane) {
  if (!T) {
    const int r = lane & 31, h = lane >> 5;
    return *reinterpret_cast<const bf16x8*>(s + (row0 + r) * 64 + (((2 * ks + h) ^ ((r >> 2) & 3)) << 4));
  }
  // K-major image: 16-lane group g covers rows row0 + 16 (g & 1) .. +15 and k-block 16 ks + 8 (g >> 1);
  // lane 4q+p of the group points at k-row q, rows 4p .. 4p+3 of the block and RECEIVES its own
  // row's 4 k-values (ds_read_b64_tr_b16); two reads (k-rows q and q + 4) make one operand.
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int k = 16 * ks + 8 * (g >> 1) + q;
  const int mo = row0 + 16 * (g & 1) + 4 * p;
  const char* a0 = s + k * 512 + ((((mo >> 5) ^ (k & 3)) << 6) | ((2 * mo) & 63));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * 512));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- tile order: split index slowest; bijective XCD remap (blocks b and b + 8 share an XCD's L2: each XCD walks a
// contiguous run of the linear order), then groups of GROUP_M row tiles x all column tiles, row tile fastest
struct TileCoord { int z, m0, n0, kbeg; };
// JOINT form (split-K launches, round 5): the remap runs over all splits x tiles items, so an XCD's contiguous run walks whole
// groups of group_m = 32 / tiles_n row tiles x ALL column tiles of ONE split -- 32 tiles, one per CU of the XCD, that share
// group_m A panels and every B panel of that K range (the per-split remap handed an XCD 20 of a split's 160 tiles and 12 of the
// next split's: a 5 x 4 rectangle, each A panel fetched by two XCDs and each B panel by four: 5.8 GB beyond L2 for 1.46 GB of
// operands in the image projection's bf16 weight gradient, profiles/r04_pmc_gemm.txt).
__device__ __forceinline__ TileCoord tile_coord(const BigArgs& g, int w) {    // w: work item (tile x split)
  const int ntiles = g.tiles_m * g.tiles_n;
  int z, id;
  if (g.joint) {
    const int ntot = ntiles * g.splits;
    const int q8 = ntot / 8, r8 = ntot % 8, xcd = w % 8, k = w / 8;
    id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    z = id / ntiles;
    id -= z * ntiles;
  } else {
    z = w / ntiles;
    id = w % ntiles;
    const int q8 = ntiles / 8, r8 = ntiles % 8, xcd = id % 8, k = id / 8;
    id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  const int per_group = g.group_m * g.tiles_n;
  const int grp = id / per_group, in = id % per_group;
  const int gm0 = grp * g.group_m;
  const int gsz = min(g.group_m, g.tiles_m - gm0);
  return TileCoord{z, (gm0 + in % gsz) * TM, (in / gsz) * TN, z * g.kchunk};
}

// all but the 4 * later youngest LDS-DMA copies of this wave have landed (4 copies per thread and slab)
__device__ __forceinline__ void wait_copies(int later) {
  if (later >= 3)      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool GUARD_M, typename OT>
__device__ __forceinline__ void store_tile(const BigArgs& g, OT* C, const f32x16 (&acc)[4][2], const float (&bv)[2],
                                           int row_base, int col0, bool relu) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + j * 32;
    if (col >= g.N) continue;
    OT* cp = C + (long long)row_base * g.ldc + col;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int dr = i * 32 + (e & 3) + 8 * (e >> 2);       // compile-time row offset
        float v = acc[i][j][e] + bv[j];
        if (relu) v = fmaxf(v, 0.f);
        if (!GUARD_M || row_base + dr < g.M) cp[(long long)dr * g.ldc] = (OT)v;
      }
    }
  }
}

template <bool TA, bool TB>
__global__ void __launch_bounds__(NT, 2) gemm_bf16_big_kernel(const BigArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;             // rows wr*128 .. +127, columns wc*64 .. +63

  // ---- tile order: split index slowest; bijective XCD remap, then groups of GROUP_M row tiles, m fastest
  const TileCoord tc = tile_coord(g, blockIdx.x);
  const int z = tc.z, m0 = tc.m0, n0 = tc.n0, kbeg = tc.kbeg;
  const int S = (min(g.K, kbeg + g.kchunk) - kbeg) / TK;       // slabs of this split

  gbf16* qa[NG];
  gbf16* qb[NG];
  init_src<TA>(qa, g.A, g.lda, m0, g.M, kbeg, wave, lane);
  init_src<TB>(qb, g.B, g.ldb, n0, g.N, kbeg, wave, lane);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
  for (int p = 0; p < NSLOT - 1; ++p)
    if (p < S) {
      stage_operand<TA>(qa, g.lda, smem + p * SLOT_BYTES, wave);
      stage_operand<TB>(qb, g.ldb, smem + p * SLOT_BYTES + OP_BYTES, wave);
    }
  int slot = 0;                                        // slot of slab s
  for (int s = 0; s < S; ++s) {
    const int later = min(NSLOT - 2, S - 1 - s);       // slabs issued after slab s that may stay in flight
    wait_copies(later);
    __builtin_amdgcn_s_barrier();
    const char* sA = smem + slot * SLOT_BYTES;
    const char* sB = sA + OP_BYTES;
    bf16x8 a0[4], b0[2], a1[4], b1[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = read_frag<TA>(sA, wr * 128 + i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) b0[j] = read_frag<TB>(sB, wc * 64 + j * 32, 0, lane);
    if (s + NSLOT - 1 < S) {                           // refill the slot of slab s-1 (its address math hides LDS latency)
      const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
      stage_operand<TA>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
      stage_operand<TB>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = read_frag<TA>(sA, wr * 128 + i * 32, 1, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) b1[j] = read_frag<TB>(sB, wc * 64 + j * 32, 1, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
    slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
  }

  // ---- epilogue: D[row = (e & 3) + 8 (e >> 2) + 4 h][col = lane & 31]; a half-wave stores 128 contiguous bytes
  const int r = lane & 31, h = lane >> 5;
  const bool split = g.splits > 1;
  const bool relu = !split && (g.flags & VQF_GEMM_RELU) != 0;
  float* C = split ? g.C + (size_t)z * g.M * g.N : g.C;
  float bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + r;
    bv[j] = (!split && g.bias && col < g.N) ? g.bias[col] : 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the biases, once: no wait may sit between the stores
  const int row_base = m0 + wr * 128 + 4 * h;
  if (g.flags & VQF_GEMM_OUT_BF16) {                 // bf16 storage of the result (round-to-nearest-even), never with split-K
    __bf16* Cb = reinterpret_cast<__bf16*>(g.C);
    if (m0 + TM <= g.M) store_tile<false>(g, Cb, acc, bv, row_base, n0 + wc * 64 + r, relu);
    else                store_tile<true>(g, Cb, acc, bv, row_base, n0 + wc * 64 + r, relu);
  } else {
    if (m0 + TM <= g.M) store_tile<false>(g, C, acc, bv, row_base, n0 + wc * 64 + r, relu);
    else                store_tile<true>(g, C, acc, bv, row_base, n0 + wc * 64 + r, relu);
  }
  if (split && g.cnt) {      // the tile's K slices are combined in this launch by its last arriver (ring idle: see gemm_f32_big.hip)
    const VqfSplitkTile st = {g.cnt, g.C, g.Cfinal, g.bias, g.M, g.N, g.ldc_final, g.flags};
    vqf_splitk_combine<TM, TN, NT>(st, (m0 / TM) * g.tiles_n + n0 / TN, g.splits, m0, n0, threadIdx.x, reinterpret_cast<float*>(smem));
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Ping-pong variant (round 2, the default): same tile, staging, LDS images and epilogue as the kernel above, but the
// two waves that share a SIMD (wave w and w + 4: row halves wr = 0 / 1 of the tile) run HALF A SLAB APART.  A slab
// is two barrier-separated segments per wave,
//     L(s): 12 fragment reads of slab s (both k-steps) + the 4 LDS-DMA copies that refill the slot of slab s-1
//           + counted vmcnt (my copies of slab s+1 have landed) + lgkmcnt(0)
//     M(s): the slab's 16 MFMAs back to back under s_setprio 1 (no LDS, no VMEM: nothing in it can stall),
// and waves 4-7 execute one extra s_barrier before the loop (waves 0-3 one after it), so in every barrier interval one
// wave of each SIMD multiplies while its partner loads: the matrix pipe sees one uninterrupted MFMA stream instead
// of two waves that read together and then multiply together (the lockstep kernel above: 44 % MFMA-busy).
// sched_barrier(0) pins both segments between their barriers (hipcc otherwise moves MFMAs across raw s_barriers).
// Hazards, with global barrier numbers (#0 = pre-loop; waves 0-3: L(s) in (#2s, #2s+1), M(s) in (#2s+1, #2s+2);
// waves 4-7 one interval later):
//   RAW  slab s+1 is read by waves 0-3 after #2s+2 and by waves 4-7 after #2s+3; every wave's counted vmcnt for ITS
//        copies of slab s+1 sits at the end of its L(s), i.e. before #2s+1 (waves 0-3) / #2s+2 (waves 4-7).
//   WAR  the copies of slab s+4 overwrite the slot of slab s-1; they are issued after #2s at the earliest, the
//        reads of slab s-1 were retired (lgkmcnt(0) in front of the barrier that ends L(s-1)) before #2s-1 / #2s.
template <bool TA, bool TB>
__global__ void __launch_bounds__(NT, 2) gemm_bf16_pp_kernel(const BigArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;             // rows wr*128 .. +127, columns wc*64 .. +63

  // PERSISTENT workgroups (gemm_f32_big.hip has the argument): workgroup b runs work items b, b + gridDim.x, ...; the
  // slab ring runs on across tiles, each wave issues the next tile's first NSLOT-1 slabs behind its own output stores
  // (every LDS read of a tile is retired before the barrier its last L segment ends with, and both wave halves have
  // passed that barrier when either leaves the loop).  The barrier pattern below is balanced per tile.
  const int total = g.tiles_m * g.tiles_n * g.splits;
  int w = blockIdx.x;
  int z, m0, n0, kbeg, S;                              // current work item (uniform over the workgroup)
  gbf16* qa[NG];
  gbf16* qb[NG];
  int slot = 0;                                        // ring slot of the tile's slab 0, then of slab s
  auto begin_tile = [&]() {
    const TileCoord tc = tile_coord(g, w);
    z = tc.z; m0 = tc.m0; n0 = tc.n0; kbeg = tc.kbeg;
    S = (min(g.K, kbeg + g.kchunk) - kbeg) / TK;       // slabs of this split
    init_src<TA>(qa, g.A, g.lda, m0, g.M, kbeg, wave, lane);
    init_src<TB>(qb, g.B, g.ldb, n0, g.N, kbeg, wave, lane);
    int sl = slot;
#pragma unroll
    for (int p = 0; p < NSLOT - 1; ++p) {
      if (p < S) {
        stage_operand<TA>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
        stage_operand<TB>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
      }
      sl = (sl + 1 == NSLOT) ? 0 : sl + 1;
    }
  };
  begin_tile();
  for (;;) {
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  {                                                    // my copies of slab 0 (4 per slab and thread)
    const int later = min(NSLOT - 1, S) - 1;
    wait_copies(later);
  }
  __builtin_amdgcn_s_barrier();                        // #0: every wave's copies of slab 0 have landed
  if (wr) __builtin_amdgcn_s_barrier();                // waves 4-7 fall half a slab behind (wave-uniform branch)

#ifdef VQF_PP_STAMPS
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, acc_t[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
  for (int s = 0; s < S; ++s) {
    // ---------------- L(s): the partner wave of this SIMD is multiplying ----------------
    __builtin_amdgcn_sched_barrier(0);
    VQF_STAMP(t0);
    const char* sA = smem + slot * SLOT_BYTES;
    const char* sB = sA + OP_BYTES;
    bf16x8 a0[4], b0[2], a1[4], b1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b0[j] = read_frag<TB>(sB, wc * 64 + j * 32, 0, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = read_frag<TA>(sA, wr * 128 + i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) b1[j] = read_frag<TB>(sB, wc * 64 + j * 32, 1, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = read_frag<TA>(sA, wr * 128 + i * 32, 1, lane);
    VQF_STAMP(t1);                                     // (stamp builds: the 12 reads have RETURNED here)
    if (!VQF_PP_GLDS_M && s + NSLOT - 1 < S) {         // slab s+4 into the slot of slab s-1
      const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
      stage_operand<TA>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
      stage_operand<TB>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
    }
    VQF_STAMP(t2);
    {                                                  // my copies of slab s+1; later slabs stay in flight
      const int later = min(s + NSLOT - 1 - VQF_PP_GLDS_M, S - 1) - (s + 1);
      wait_copies(later);
    }
    VQF_STAMP(t3);
    __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): fragments in registers, slot s no longer read by me
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    VQF_STAMP(t4);
    // ---------------- M(s): the partner loads ----------------
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int n = 0; n < 16; ++n) {                     // MFMA n: k-step n >> 3, row tile (n >> 1) & 3, column tile n & 1
      const int i = (n >> 1) & 3, j = n & 1;
      if (n == 16 - VQF_PP_TAIL) {                     // early hand-over: the partner's MFMAs queue up behind my last ones
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (VQF_PP_GLDS_M && n == 2) {                   // refill copies behind the first MFMAs (their issue overlaps the matrix pipe)
        __builtin_amdgcn_sched_barrier(0);
        if (s + NSLOT - 1 < S) {
          const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
          stage_operand<TA>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
          stage_operand<TB>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[i][j] = (n < 8) ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    VQF_STAMP(t5);
    __builtin_amdgcn_sched_barrier(0);
    if (VQF_PP_TAIL == 0) __builtin_amdgcn_s_barrier();
    VQF_STAMP(t6);
#ifdef VQF_PP_STAMPS
    acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += t4 - t3; acc_t[4] += t5 - t4;
    acc_t[5] += t6 - t5; acc_t[6] += (s ? t0 - t7 : 0);
    t7 = t6;
#endif
    slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
  }
  if (!wr) __builtin_amdgcn_s_barrier();               // waves 0-3 absorb the barrier waves 4-7 spent on the stagger
#ifdef VQF_PP_STAMPS
  if (g.dbg && lane == 0) {
    unsigned long long* d = g.dbg + ((size_t)w * 8 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 7; ++i) d[i] = acc_t[i];
    d[7] = (unsigned long long)S;
  }
#endif

  // ---- epilogue (as above)
  const int r = lane & 31, h = lane >> 5;
  const bool split = g.splits > 1;
  const bool relu = !split && (g.flags & VQF_GEMM_RELU) != 0;
  float* C = split ? g.C + (size_t)z * g.M * g.N : g.C;
  float bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + r;
    bv[j] = (!split && g.bias && col < g.N) ? g.bias[col] : 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const int row_base = m0 + wr * 128 + 4 * h;
  if (g.flags & VQF_GEMM_OUT_BF16) {
    __bf16* Cb = reinterpret_cast<__bf16*>(g.C);
    if (m0 + TM <= g.M) store_tile<false>(g, Cb, acc, bv, row_base, n0 + wc * 64 + r, relu);
    else                store_tile<true>(g, Cb, acc, bv, row_base, n0 + wc * 64 + r, relu);
  } else {
    if (m0 + TM <= g.M) store_tile<false>(g, C, acc, bv, row_base, n0 + wc * 64 + r, relu);
    else                store_tile<true>(g, C, acc, bv, row_base, n0 + wc * 64 + r, relu);
  }
  if (split && g.cnt) {      // the tile's K slices are combined in this launch by its last arriver (ring idle: see gemm_f32_big.hip)
    const VqfSplitkTile st = {g.cnt, g.C, g.Cfinal, g.bias, g.M, g.N, g.ldc_final, g.flags};
    vqf_splitk_combine<TM, TN, NT>(st, (m0 / TM) * g.tiles_n + n0 / TN, g.splits, m0, n0, threadIdx.x, reinterpret_cast<float*>(smem));
  }
  w += gridDim.x;
  if (w >= total) break;
  begin_tile();
  }
}


// ---------------------------------------------------------------------------------------------------------------
// pp16: the ping-pong loop on v_mfma_f32_16x16x32_bf16, both operands K-contiguous (the projections' forward GEMMs).
//   * one MFMA covers the slab's whole 32-deep k: a fragment = one ds_read_b128 (lane l: row l & 15, 16-byte chunk
//     l >> 4 of the 64-byte row); 8 A + 4 B reads and 32 MFMAs of 16 cycles per slab and wave.  The chip holds a higher
//     clock on this shape than on 32x32x16 (MI355X_MICROARCH.md, DVFS give-back item 7).
//   * chunk swizzle g(row) = (-(row >> 2)) & 3 on the copy's source address and on the read: the four 16-lane phases of
//     a ds_read_b128 ({0-3,12-15,20-27}, ...) then hit 16 distinct 16-byte slots for this lane -> (row, chunk) map.
//   * the wave's 64 columns are INTERLEAVED over its 4 column tiles (column 4c + j belongs to tile j, MFMA column c):
//     the copy of the B slab permutes rows on its per-lane SOURCE address (LDS row 16j + c holds B row 4c + j), the
//     loop is unchanged, and a lane's four column tiles are 16 contiguous output bytes: 32 global_store_dwordx4
//     (bf16 output: dwordx2) per lane instead of 128 dword (short) stores -- the epilogue is store-issue bound.
__device__ __forceinline__ int swz16(int row) { return (0 - (row >> 2)) & 3; }

template <bool IS_B>
__device__ __forceinline__ void init_src16(gbf16* (&q)[NG], const bf16_t* base, int ld, int r0, int R, int k0,
                                           int wave, int lane) {
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int rho = i * 128 + wave * 16 + (lane >> 2);           // LDS row this lane's 16 bytes land in
    const int chunk = (lane & 3) ^ swz16(rho);
    int row = rho;
    if (IS_B) { const int sig = rho & 63; row = (rho & ~63) + 4 * (sig & 15) + (sig >> 4); }   // strip row 16j + c <- column 4c + j
    q[i] = (gbf16*)(base + (long long)min(r0 + row, R - 1) * ld + k0 + chunk * 8);
  }
}

__device__ __forceinline__ bf16x8 read_frag16(const char* s, int row0, int lane) {
  const int row = row0 + (lane & 15);
  const char* p = s + row * 64 + (((lane >> 4) ^ swz16(row)) << 4);
  return *reinterpret_cast<const bf16x8*>(p);
}

template <bool GUARD_M, bool VEC, typename OT>
__device__ __forceinline__ void store_tile16(const BigArgs& g, OT* C, const f32x4 (&acc)[8][4], int row0, int col0,
                                             int lane, bool relu, bool use_bias) {
  const int cm = lane & 15, rq = lane >> 4;
  const int col = col0 + 4 * cm;                                  // this lane's 4 consecutive output columns
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = (use_bias && col + j < g.N) ? g.bias[col + j] : 0.f;
  __builtin_amdgcn_s_waitcnt(0x0F70);               // the biases, once: no wait may sit between the stores
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = row0 + 16 * i + 4 * rq + e;
      if (GUARD_M && row >= g.M) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[i][j][e] + bv[j];
        if (relu) v[j] = fmaxf(v[j], 0.f);
      }
      OT* cp = C + (long long)row * g.ldc + col;
      if (VEC) {
        if (col < g.N) {
          if (sizeof(OT) == 4) {
            *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
          } else {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<bf16x4*>(cp) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col + j < g.N) cp[j] = (OT)v[j];
      }
    }
  }
}

__global__ void __launch_bounds__(NT, 2) gemm_bf16_pp16_kernel(const BigArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;             // rows wr*128 .. +127, column strip wc*64 .. +63

  // persistent workgroups, the slab ring running on across tiles (see gemm_bf16_pp_kernel)
  const int total = g.tiles_m * g.tiles_n * g.splits;
  int w = blockIdx.x;
  int z, m0, n0, kbeg, S;                              // current work item (uniform over the workgroup)
  gbf16* qa[NG];
  gbf16* qb[NG];
  int slot = 0;                                        // ring slot of the tile's slab 0, then of slab s
  auto begin_tile = [&]() {
    const TileCoord tc = tile_coord(g, w);
    z = tc.z; m0 = tc.m0; n0 = tc.n0; kbeg = tc.kbeg;
    S = (min(g.K, kbeg + g.kchunk) - kbeg) / TK;       // slabs of this split
    init_src16<false>(qa, g.A, g.lda, m0, g.M, kbeg, wave, lane);
    init_src16<true>(qb, g.B, g.ldb, n0, g.N, kbeg, wave, lane);
    int sl = slot;
#pragma unroll
    for (int p = 0; p < NSLOT - 1; ++p) {
      if (p < S) {
        stage_operand<false>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
        stage_operand<false>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
      }
      sl = (sl + 1 == NSLOT) ? 0 : sl + 1;
    }
  };
  begin_tile();
  for (;;) {
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  {                                                    // my copies of slab 0 (4 per slab and thread)
    const int later = min(NSLOT - 1, S) - 1;
    wait_copies(later);
  }
  __builtin_amdgcn_s_barrier();                        // #0: every wave's copies of slab 0 have landed
  if (wr) __builtin_amdgcn_s_barrier();                // waves 4-7 fall half a slab behind (wave-uniform branch)

  for (int s = 0; s < S; ++s) {
    // ---------------- L(s) ----------------
    __builtin_amdgcn_sched_barrier(0);
    const char* sA = smem + slot * SLOT_BYTES;
    const char* sB = sA + OP_BYTES;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = read_frag16(sB, wc * 64 + 16 * j, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = read_frag16(sA, wr * 128 + 16 * i, lane);
    if (!VQF_PP_GLDS_M && s + NSLOT - 1 < S) {         // slab s+4 into the slot of slab s-1
      const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
      stage_operand<false>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
      stage_operand<false>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
    }
    {                                                  // my copies of slab s+1; later slabs stay in flight
      const int later = min(s + NSLOT - 1 - VQF_PP_GLDS_M, S - 1) - (s + 1);
      wait_copies(later);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): fragments in registers, slot s no longer read by me
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- M(s) ----------------
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int n = 0; n < 32; ++n) {                     // MFMA n: row tile n >> 2, column tile n & 3
      const int i = n >> 2, j = n & 3;
      if (VQF_PP_GLDS_M && n == 4) {                   // refill copies behind the first MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (s + NSLOT - 1 < S) {
          const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
          stage_operand<false>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
          stage_operand<false>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
  }
  if (!wr) __builtin_amdgcn_s_barrier();               // waves 0-3 absorb the barrier waves 4-7 spent on the stagger

  const bool split = g.splits > 1;
  const bool relu = !split && (g.flags & VQF_GEMM_RELU) != 0;
  const bool use_bias = !split && g.bias != nullptr;
  const int row0 = m0 + wr * 128, col0 = n0 + wc * 64;
  const bool full = m0 + TM <= g.M;
  if (g.rowscale && !split) {                          // per-sample scale ahead of bias / relu (F.normalize folded into co_att_conv1)
    const int rq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float rs[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) rs[e] = g.rowscale[min(row0 + 16 * i + 4 * rq + e, g.M - 1) / g.rps];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j][e] *= rs[e];
    }
  }
  if (g.flags & VQF_GEMM_OUT_BF16) {                   // bf16 storage of the result (round-to-nearest-even), never with split-K
    __bf16* Cb = reinterpret_cast<__bf16*>(g.C);
    const bool vec = (g.N % 4 == 0) && (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(Cb) & 7) == 0);
    if (vec) { if (full) store_tile16<false, true>(g, Cb, acc, row0, col0, lane, relu, use_bias);
               else      store_tile16<true, true>(g, Cb, acc, row0, col0, lane, relu, use_bias); }
    else     { store_tile16<true, false>(g, Cb, acc, row0, col0, lane, relu, use_bias); }
  } else {
    float* C = split ? g.C + (size_t)z * g.M * g.N : g.C;
    const bool vec = (g.N % 4 == 0) && (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                     (!split || (((size_t)g.M * g.N) % 4 == 0));
    if (vec) { if (full) store_tile16<false, true>(g, C, acc, row0, col0, lane, relu, use_bias);
               else      store_tile16<true, true>(g, C, acc, row0, col0, lane, relu, use_bias); }
    else     { store_tile16<true, false>(g, C, acc, row0, col0, lane, relu, use_bias); }
  }
  if (split && g.cnt) {      // the tile's K slices are combined in this launch by its last arriver (ring idle: see gemm_f32_big.hip)
    const VqfSplitkTile st = {g.cnt, g.C, g.Cfinal, g.bias, g.M, g.N, g.ldc_final, g.flags};
    vqf_splitk_combine<TM, TN, NT>(st, (m0 / TM) * g.tiles_n + n0 / TN, g.splits, m0, n0, threadIdx.x, reinterpret_cast<float*>(smem));
  }
  w += gridDim.x;
  if (w >= total) break;
  begin_tile();
  }
}

int pick_splits(int tiles, int K, int M, int N, size_t ws_bytes) {
  if (tiles >= 768) return 1;
  int best = 1;
  double best_cost = 1e30;
  for (int sp = 1; sp <= 16; ++sp) {
    if (sp > 1 && (size_t)sp * M * N * sizeof(float) > ws_bytes) break;
    if (sp > 1 && K / sp < 16 * TK) break;
    const long long blocks = (long long)tiles * sp;
    const double rounds = (double)((blocks + 255) / 256);            // one workgroup per CU
    const double cost = rounds / sp + (sp > 1 ? 0.02 * sp : 0.0);    // + slab write / reduce traffic
    if (cost < best_cost - 1e-12) { best_cost = cost; best = sp; }
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

template <bool TA, bool TB>
int launch(const BigArgs& g, hipStream_t s) {
  // > 64 KB of dynamic LDS needs the attribute, once per device and instantiation (common.h)
  static VqfDynLdsFlags attr = {}, attr_pp = {};
  const int loopf = vqf_opt(VQF_OPT_GEMM_BF16_LOOP, 1);  // A/B switch (tools/gemm_bf16_ab.py flips it in one process):
  const bool pingpong = loopf != 0;                    // 0 selects the lockstep kernel of round 1
  // the ping-pong kernels run as persistent workgroups, one per CU (VQF_GEMM_BF16_PERSIST=0: one workgroup per item)
  const int total = g.tiles_m * g.tiles_n * g.splits;
  int nwg = total;
  if (vqf_opt(VQF_OPT_GEMM_BF16_PERSIST, 1) != 0) {
    int cus = vqf_cu_count() & ~7;                     // a multiple of 8 keeps every workgroup's items on its own XCD
    const int lim = vqf_opt(VQF_OPT_GEMM_CU_LIMIT, 0) & ~7;
    if (lim >= 8 && lim < cus) cus = lim;
    if (cus >= 8 && total > cus) nwg = cus;
  }
  if (pingpong && !TA && !TB && loopf != 3) {          // VQF_OPT_GEMM_BF16_LOOP = 3: the 32x32x16 ping-pong loop also for (0,0)
    static VqfDynLdsFlags attr16 = {};
    if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_bf16_pp16_kernel), SMEM_BIG, attr16)) return e;
    VQF_LAUNCH(KID_GEMM_BF16, gemm_bf16_pp16_kernel, dim3(nwg), dim3(NT), SMEM_BIG, s, g);
    return vqf_last_error();
  }
  if (pingpong) {
    if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_bf16_pp_kernel<TA, TB>), SMEM_BIG, attr_pp)) return e;
    VQF_LAUNCH(KID_GEMM_BF16, (gemm_bf16_pp_kernel<TA, TB>), dim3(nwg), dim3(NT), SMEM_BIG, s, g);
    return vqf_last_error();
  }
  if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_bf16_big_kernel<TA, TB>), SMEM_BIG, attr)) return e;
  VQF_LAUNCH(KID_GEMM_BF16, (gemm_bf16_big_kernel<TA, TB>), dim3(g.tiles_m * g.tiles_n * g.splits), dim3(NT), SMEM_BIG,
             s, g);
  return vqf_last_error();
}

bool big_applies(int ta, int tb, int M, int N, int K, int flags) {
  const bool enabled = vqf_opt(VQF_OPT_GEMM_BF16_BIG, 1) != 0;   // A/B switch: 0 selects the 128x128 kernel everywhere
  if (!enabled || (K % TK) || M < TM || N < 128 || (flags & VQF_GEMM_ACCUM)) return false;
  if (ta && (M % 8)) return false;
  if (tb && (N % 8)) return false;
  // few output tiles and a long K (co_att_conv1's weight gradient: 4 x 4 tiles): the 128x128 kernel's
  // 64 tiles x 4 splits measured faster (0.36 vs 0.47 ms) than 16 tiles x 16 splits here
  if (((M + TM - 1) / TM) * ((N + TN - 1) / TN) < 64) return false;
  return true;
}

}  // namespace

// scratch the big kernel would like for this shape (split-K slabs); 0 when it does not apply
size_t vqf_gemm_bf16_big_ws_bytes(int ta, int tb, int M, int N, int K) {
  if (!big_applies(ta, tb, M, N, K, 0)) return 0;
  const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  const int sp = pick_splits(tiles, K, M, N, (size_t)1 << 40);
  return sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
}

// 0 = this kernel does not apply (caller falls back to gemm_bf16.hip), 1 = launched (rc holds the status)
int vqf_gemm_bf16_big_try(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C,
                          int ldc, const float* bias, int flags, const float* rowscale, int rps, void* ws, size_t ws_bytes,
                          hipStream_t s, int* rc) {
  if (!big_applies(ta, tb, M, N, K, flags)) return 0;
  if (rowscale) {      // the epilogue scale lives in the 16x16x32 (0,0) kernel only; anything else falls back to the 128x128 kernel
    const int loopf = vqf_opt(VQF_OPT_GEMM_BF16_LOOP, 1);
    if (ta || tb || loopf == 0 || loopf == 3) return 0;
    ws = nullptr; ws_bytes = 0;                        // no split-K
  }
  BigArgs g;
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.bias = bias;
  g.rowscale = rowscale; g.rps = rps > 0 ? rps : 1;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.flags = flags;
  g.tiles_m = (M + TM - 1) / TM;
  g.tiles_n = (N + TN - 1) / TN;
  const int tiles = g.tiles_m * g.tiles_n;
  int splits = (flags & VQF_GEMM_OUT_BF16) ? 1 : pick_splits(tiles, K, M, N, (ws && aligned16(ws)) ? ws_bytes : 0);
  const int slabs = K / TK;
  const int per = (slabs + splits - 1) / splits;
  g.kchunk = per * TK;
  splits = (K + g.kchunk - 1) / g.kchunk;
  g.splits = splits;
  g.cnt = nullptr; g.Cfinal = C; g.ldc_final = ldc;
  if (splits > 1) {
    g.C = (float*)ws; g.ldc = N;
    if (tiles >= 64 && vqf_opt(VQF_OPT_GEMM_SPLITK_FUSED, 1) != 0) g.cnt = vqf_splitk_counters(tiles);   // (as gemm_f32_big.hip)
  }
  g.group_m = pick_group_m(g.tiles_m, g.tiles_n, splits);
  g.joint = 0;
  if (splits > 1 && g.tiles_n <= 32 && 32 % g.tiles_n == 0 && vqf_opt(VQF_OPT_GEMM_SPLITK_ORDER, 0) == 1) {
    g.joint = 1;                                        // an XCD's 32 CUs take group_m row tiles x all column tiles of one split
    g.group_m = std::min(g.tiles_m, 32 / g.tiles_n);
  }
#ifdef VQF_PP_STAMPS
  g.dbg = (splits == 1 && ws && ws_bytes >= (size_t)tiles * 8 * 8 * 8) ? (unsigned long long*)ws : nullptr;
#endif
  vqf_prof_dims(M, N, K);
  vqf_stat_bump(VQF_STAT_GEMM_BF16_BIG);
  if (ta) *rc = tb ? launch<true, true>(g, s) : launch<true, false>(g, s);
  else    *rc = tb ? launch<false, true>(g, s) : launch<false, false>(g, s);
  if (*rc != VQF_OK) vqf_splitk_counters_clear(g.cnt, tiles, s);
  if (*rc == VQF_OK && splits > 1 && !g.cnt) *rc = vqf_splitk_reduce((const float*)ws, splits, M, N, C, ldc, bias, flags, s);
  return 1;
}
