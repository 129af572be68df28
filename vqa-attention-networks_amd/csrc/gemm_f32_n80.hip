// One-round fp32 GEMM for the M = 512 projections (gfx950): C[m, n] = sum_k A[m, k] * B[n, k] (+ bias[n]) (relu), A (M, K) and
// B (N, K) K-contiguous -- the 2048 -> 5000 question / image projections of the MFB blocks (mfb.py:76,92,126,127;
// mhb_coAtt.py:60-66 ...) at batch 512.
//
// Why: 512 x 5000 outputs are 160 tiles of 128 x 128 -- 0.31 of a round of the 512 workgroup slots -- so vqf_gemm_f32 cut K
// in 3-4 slices and added a slab-reduce launch: 116 us for 67 us of MFMA time (K = 2048).  A 128 x 80 tile gives 4 x 63 = 252
// tiles: ONE workgroup per CU, the whole K each, no slabs, no second launch.
//
// Structure: 8 waves, wave w owns rows 16 w .. 16 w + 15 of the tile and all 80 columns = five v_mfma_f32_16x16x4_f32 tiles
// (lane = (c, kq) = (lane % 16, lane / 16); one MFMA multiplies the four k values its four lane groups hold).  Operands are
// staged by LDS-DMA (global_load_lds_dwordx4) in 32-wide k slabs -- A 128 rows = 16 KB, every wave copying its OWN 16 rows; B
// 80 rows = 10 KB, shared -- into a ring of four 26 KB slots, three to four slabs in flight, the 16-byte chunks of a row XOR-swizzled
// (swz below) so that each 16-lane group of a fragment read covers the 16 slots of a bank row.  One s_barrier per slab; a wave
// waits for its own copies with a counted vmcnt in front of it (no compiler-visible vector loads in the loop: a first form
// with A loaded straight into registers got a compiler-inserted vmcnt(0) in front of every slab's MFMAs).  Every output
// element is one accumulation chain in k order (slab, chunk, j; the four k of an MFMA summed inside it): deterministic, not
// the bit pattern of the 32x32x2 kernels.
// Preconditions: K % 128 == 0, K >= 512, 16-byte aligned A / B with lda, ldb % 4 == 0; any M, N (edge tiles are clamped on
// load and masked on store).
// The same kernel with FOUR column tiles = the four gates of 16 hidden units is the fused LSTM forward step
// (gemm_f32_n80_kernel<4, true>, vqf_lstm_step16_try below; mfb.py:69).
#include "common.h"

namespace {

typedef const float __attribute__((address_space(1))) gfloat;

constexpr int TM = 128, TN = 80, TK = 32, NT = 512, NSLOT = 4;
constexpr int A_BYTES = TM * TK * 4;                       // 16 KB per slab: wave w copies A chunks 2w, 2w+1
// B per slab: NTC column tiles x 16 rows x 128 bytes = 2 NTC copies -- NTC = 5 (the GEMM): wave w copies chunk w (and 8 + w for
// w < 2); NTC = 4 (the LSTM step below): one chunk per wave
constexpr int b_bytes(int ntc) { return 16 * ntc * TK * 4; }
constexpr int slot_bytes(int ntc) { return A_BYTES + b_bytes(ntc); }

struct N80Args {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K, lda, ldb, ldc, flags, tiles_m, tiles;
  const float* c_prev; float* c_out; float* h_out; int H;      // CELL (the fused LSTM step)
};

__device__ __forceinline__ float sigm_f(float x) { return 1.0f / (1.0f + expf(-x)); }     // lstm_cell.hip's sigm

// chunk XOR of LDS row rho.  ds_read_b128 serves a wave in four NON-contiguous groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11,
// 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS table), i.e. with lane = (c, kq) a group holds all 16 rows c, rows
// 4-11 with the OTHER kq of the pair; a group is conflict-free when its 16 lanes hit the 16 slots of a 256-byte bank row.
// i = (rho >> 1) & 7 alone ((rho & 1) picks the 128-byte half) left rows 4-11 on the slots of rows 0-3 / 12-15: 2-way, half of
// the LDS-active cycles (SQ_LDS_BANK_CONFLICT); flipping bit 1 for i in 2..5 separates them (checked exhaustively).
__device__ __forceinline__ int swz(int rho) {
  const int i = (rho >> 1) & 7;
  return i ^ (((i + 2) & 4) >> 1);
}

// all but the copies of the `later` youngest slabs of this wave have landed
__device__ __forceinline__ void wait_own(int later, bool two) {
  if (two) {                                               // waves 0, 1: 2 + 2 copies per slab
    if (later >= 3)      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {                                                 // waves 2-7: 2 + 1 copies per slab
    if (later >= 3)      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// NTC = 5, CELL = false: the GEMM above.
// NTC = 4, CELL = true: ONE STEP of the question encoder's LSTM (mfb.py:69; vqf_lstm_step_fwd) -- A = h_{t-1} (B, H), B = W_hh
// (4H, H), the tile's 64 columns are the FOUR GATES of 16 hidden units: LDS row 16 t + c of a B slab holds W_hh row t H + u0 + c
// (the copy's source address is per lane, nothing else moves), so column tile t of a lane is gate t (i, f, g, o) of unit u0 + c
// for the lane's four rows and THE CELL IS A PER-LANE EPILOGUE: pre = acc + gates_t (x_t W_ih^T + b, fetched at kernel entry),
// i, f, o = sigmoid, g = tanh, c_t = f c_{t-1} + i g, h_t = o tanh(c_t); gates_t leaves activated (kept for the backward).
// The arithmetic of vqf_lstm_cell_fwd in its order; the product's k order is this kernel's (not the bits of the per-wave form).
template <int NTC, bool CELL>
__global__ void __launch_bounds__(NT, 1) gemm_f32_n80_kernel(const N80Args g) {
  constexpr int SLOT_BYTES = slot_bytes(NTC), TNC = 16 * NTC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  // XCD-aware order: consecutive workgroups go to the 8 XCDs round-robin; an XCD's workgroups take CONSECUTIVE tiles, row tile
  // fastest, so the tiles_m workgroups that share a B panel share an L2
  const int per = (gridDim.x + 7) >> 3;
  const int lin = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (lin >= g.tiles) return;                              // (whole workgroup, before any barrier)
  const int tn = lin / g.tiles_m, tm = lin - tn * g.tiles_m;
  const int m0 = tm * TM, n0 = tn * TNC;                  // (CELL: n0 / 4 = the tile's first hidden unit)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, kq = lane >> 4;
  const bool two = NTC == 5 && wave < 2;
  const int S = g.K / TK;

  // CELL: what the epilogue adds to / multiplies with the accumulators is requested first (older than every copy: any counted
  // wait for a slab covers it); lane (c, kq), register j: row m0 + 16 wave + 4 kq + j, unit u0 + c
  float oldg[4][4], cpv[4];
  if (CELL) {
    const int u = (n0 >> 2) + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long row = m0 + 16 * wave + 4 * kq + j;
#pragma unroll
      for (int t = 0; t < 4; ++t) oldg[t][j] = g.C[row * g.ldc + t * g.H + u];
      cpv[j] = g.c_prev ? g.c_prev[row * g.H + u] : 0.f;
    }
  }

  // copy sources (rows past M / N are clamped: computed on a valid row, not stored).  A copy of 1 KB = 8 LDS rows of 128 bytes:
  // lane l writes position l & 7 of row (l >> 3), which holds source chunk (l & 7) ^ swz(row)
  gfloat* qa[2];
  gfloat* qb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = 16 * wave + 8 * i + (lane >> 3);        // my own 16 rows of A
    qa[i] = (gfloat*)(g.A + (long long)min(m0 + ra, g.M - 1) * g.lda + 4 * ((lane & 7) ^ swz(ra)));
    const int rho = 8 * (wave + 8 * i) + (lane >> 3);
    const int brow = CELL ? (min(rho, TNC - 1) >> 4) * g.H + (n0 >> 2) + (rho & 15)       // gate (rho >> 4), unit u0 + (rho & 15)
                          : min(n0 + min(rho, TNC - 1), g.N - 1);
    qb[i] = (gfloat*)(g.B + (long long)brow * g.ldb + 4 * ((lane & 7) ^ swz(rho)));
  }
  // piece p of slab s: 0, 1 = my two A chunks, 2 = B chunk `wave`, 3 = B chunk 8 + wave (waves 0, 1 only)
  auto issue_piece = [&](int s, int slot, int p) {
    lds_char* da = (lds_char*)(smem + slot * SLOT_BYTES + wave * 2048);        // wave-uniform; the DMA adds lane * 16
    lds_char* db = (lds_char*)(smem + slot * SLOT_BYTES + A_BYTES + wave * 1024);
    if (p == 0)      __builtin_amdgcn_global_load_lds(qa[0] + (long long)s * TK, da, 16, 0, 0);
    else if (p == 1) __builtin_amdgcn_global_load_lds(qa[1] + (long long)s * TK, da + 1024, 16, 0, 0);
    else if (p == 2) __builtin_amdgcn_global_load_lds(qb[0] + (long long)s * TK, db, 16, 0, 0);
    else if (two)    __builtin_amdgcn_global_load_lds(qb[1] + (long long)s * TK, db + 8 * 1024, 16, 0, 0);
  };
  auto issue = [&](int s, int slot) {
#pragma unroll
    for (int p = 0; p < 4; ++p) issue_piece(s, slot, p);
  };

  f32x4 acc[NTC];
#pragma unroll
  for (int t = 0; t < NTC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragments of one slab: A 2 x 16 bytes, B NTC tiles x 2 x 16 bytes per lane = 2 + 2 NTC reads, numbered 0 (A, u = 0), 1 (A, u = 1), 2 + 2 t + u
  constexpr int NFR = 2 + 2 * NTC;
  struct Frag { f32x4 a[2], b[NTC][2]; };
  auto read_frag = [&](int slot, Frag& f, int i) {
    const char* sa = smem + slot * SLOT_BYTES;
    if (i < 2) {
      const int ra = 16 * wave + c;
      f.a[i] = *reinterpret_cast<const f32x4*>(sa + ra * 128 + (((2 * kq + i) ^ swz(ra)) << 4));
    } else {
      const int t = (i - 2) >> 1, u = (i - 2) & 1, rho = 16 * t + c;
      f.b[t][u] = *reinterpret_cast<const f32x4*>(sa + A_BYTES + rho * 128 + (((2 * kq + u) ^ swz(rho)) << 4));
    }
  };

  // Software pipeline over the barrier: the fragments of slab s+1 are read (into the other register set) BEHIND the barrier that
  // makes slab s+1 visible, BETWEEN the MFMAs of slab s; slab s+4 is copied into the slot of slab s behind the same barrier
  // (every wave's reads of slab s were waited for -- lgkmcnt(0) -- in front of it).  The copy pieces and the fragment reads are
  // spread over the slab's eight groups of five MFMAs: an LDS-DMA piece costs its wave 60-180 issue cycles
  // (MI355X_MICROARCH.md), and with all four issued back to back behind the barrier BOTH waves of a SIMD stood in them at the same
  // time -- 3370 clocks per slab for 2560 of MFMA (97 us at K = 2048).
  auto step = [&](int s, int q, const Frag& cur, Frag& nxt) {
    const bool more = s + 1 < S, copy = s + NSLOT < S;
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
      const int u = grp >> 2, j = grp & 3;
#pragma unroll
      for (int t = 0; t < NTC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[u][j], cur.b[t][u][j], acc[t], 0, 0, 0);
      // (giving the two waves of a SIMD opposite halves of the slab for their copies and reads -- so that they do not stand in a
      //  piece's issue cycles together -- measured SLOWER: 108 vs 90 us)
      // (-DVQF_N80_NOCOPY / -DVQF_N80_NOREAD: diagnostic builds that leave the copies / the fragment reads out of the loop -- wrong
      //  results, the timing says what each costs; tools/build_variant.sh)
#ifndef VQF_N80_NOCOPY
      if (grp < 4) { if (copy) issue_piece(s + NSLOT, q, grp); }
#endif
#ifndef VQF_N80_NOREAD
      if (more && 2 * grp < NFR) { read_frag((q + 1) % NSLOT, nxt, 2 * grp); read_frag((q + 1) % NSLOT, nxt, 2 * grp + 1); }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#pragma unroll
  for (int p = 0; p < NSLOT; ++p) issue(p, p);             // S >= 16 > NSLOT
  Frag fr[2];
  wait_own(NSLOT - 1, two);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < NFR; ++i) read_frag(0, fr[0], i);
  // unrolled by NSLOT so that slots and register sets are compile-time: slab s lives in slot s % 4, set s % 2
  for (int s0 = 0; s0 < S; s0 += NSLOT) {
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) {
      const int s = s0 + q;
      if (s + 1 < S) wait_own(min(NSLOT - 2, S - 2 - s), two);     // my copies of slab s+1 have landed
      __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): my fragment reads of slab s have returned
#ifndef VQF_N80_NOBARRIER
      __builtin_amdgcn_s_barrier();                        // slab s+1 is visible; nobody reads slab s from LDS any more
#endif
      step(s, q, fr[q & 1], fr[(q + 1) & 1]);
    }
  }

  if (CELL) {
    // the cell: acc[t][j] = gate t of unit u0 + c, row m0 + 16 wave + 4 kq + j
    const int u = (n0 >> 2) + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long row = m0 + 16 * wave + 4 * kq + j;
      const float ig = sigm_f(acc[0][j] + oldg[0][j]);
      const float fg = sigm_f(acc[1][j] + oldg[1][j]);
      const float gg = tanhf(acc[2 % NTC][j] + oldg[2][j]);
      const float og = sigm_f(acc[3 % NTC][j] + oldg[3][j]);
      float* pg = g.C + row * g.ldc + u;
      pg[0] = ig; pg[g.H] = fg; pg[2 * g.H] = gg; pg[3 * g.H] = og;
      const float cn = fg * cpv[j] + ig * gg;
      g.c_out[row * g.H + u] = cn;
      g.h_out[row * g.H + u] = og * tanhf(cn);
    }
    return;
  }
  // epilogue: acc[t][j] = row m0 + 16 wave + 4 kq + j, column n0 + 16 t + c
  const bool relu = g.flags & VQF_GEMM_RELU;
#pragma unroll
  for (int t = 0; t < NTC; ++t) {
    const int col = n0 + 16 * t + c;
    if (col < g.N) {
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = m0 + 16 * wave + 4 * kq + j;
        if (row < g.M) {
          float v = acc[t][j] + bv;
          if (relu) v = fmaxf(v, 0.f);
          g.C[(long long)row * g.ldc + col] = v;
        }
      }
    }
  }
}

}  // namespace

// 1 (and *rc set) when the kernel took the product; 0: not its shape.  Used by vqf_gemm_f32 for ta == tb == 0.
int vqf_gemm_f32_n80_try(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         const float* bias, int flags, hipStream_t s, int* rc) {
  if (vqf_opt(VQF_OPT_GEMM_F32_N80, 1) == 0) return 0;
  if ((flags & ~VQF_GEMM_RELU) || (K % 128) || K < 512 || !aligned16(A) || !aligned16(B) || (lda % 4) || (ldb % 4)) return 0;
  const int tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
  const long long tiles = (long long)tiles_m * tiles_n;
  const int cus = vqf_cu_count() > 0 ? vqf_cu_count() : 256;
  // one round that fills most of the chip; with fewer tiles the 128x128 kernel's K slices win, with more its whole tiles do
  if (tiles > cus || tiles * 10 < (long long)cus * 8 || M > 1024) return 0;
  if (vqf_opt(VQF_OPT_GEMM_CU_LIMIT, 0) >= 8) return 0;    // (a CU-limited side stream keeps its persistent kernels)
  N80Args g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.flags = flags;
  g.tiles_m = tiles_m; g.tiles = (int)tiles;
  const int nwg = ((int)tiles + 7) & ~7;
  vqf_prof_dims(M, N, K);
  vqf_stat_bump(VQF_STAT_GEMM_F32_N80);
  static VqfDynLdsFlags attr = {};
  if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_n80_kernel<5, false>), NSLOT * slot_bytes(5), attr)) { *rc = e; return 1; }
  VQF_LAUNCH(KID_GEMM_A0B0, (gemm_f32_n80_kernel<5, false>), dim3(nwg), dim3(NT), NSLOT * slot_bytes(5), s, g);
  *rc = vqf_last_error();
  return 1;
}

// The fused LSTM step on this kernel (lstm_step_fwd in gemm_f32_wave.hip asks first): 1 = launched (*rc), 0 = not its shape.
int vqf_lstm_step16_try(const float* h_prev, const float* w_hh, float* gates, const float* c_prev, int B, int H, float* c_out,
                        float* h_out, hipStream_t s, int* rc) {
  if (vqf_opt(VQF_OPT_GEMM_F32_N80, 1) == 0 || vqf_opt(VQF_OPT_GEMM_CU_LIMIT, 0) >= 8) return 0;
  if ((B % TM) || (H % 128) || H < 512) return 0;          // whole row tiles (the cell has no edge guards), K = H in 128-k units
  const int tiles_m = B / TM, tiles_n = H / 16;
  const long long tiles = (long long)tiles_m * tiles_n;
  const int cus = vqf_cu_count() > 0 ? vqf_cu_count() : 256;
  if (tiles > cus || tiles * 10 < (long long)cus * 8) return 0;
  N80Args g = {};
  g.A = h_prev; g.B = w_hh; g.C = gates; g.M = B; g.N = 4 * H; g.K = H; g.lda = H; g.ldb = H; g.ldc = 4 * H;
  g.tiles_m = tiles_m; g.tiles = (int)tiles;
  g.c_prev = c_prev; g.c_out = c_out; g.h_out = h_out; g.H = H;
  const int nwg = ((int)tiles + 7) & ~7;
  vqf_prof_dims(B, 4 * H, H);
  vqf_stat_bump(VQF_STAT_GEMM_F32_N80);
  static VqfDynLdsFlags attr = {};
  if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_n80_kernel<4, true>), NSLOT * slot_bytes(4), attr)) { *rc = e; return 1; }
  VQF_LAUNCH(KID_LSTM_CELL_FWD, (gemm_f32_n80_kernel<4, true>), dim3(nwg), dim3(NT), NSLOT * slot_bytes(4), s, g);
  *rc = vqf_last_error();
  return 1;
}
