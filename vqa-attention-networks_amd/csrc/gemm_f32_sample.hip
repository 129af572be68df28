// Per-sample-tile fp32 GEMM (gfx950): C[m, n] = sum_k A[m, k] * Bop[n, k] (+ bias[n]) (relu) for row-major A (M = NS * L rows,
// K contiguous) whose rows come in SAMPLES of L = 192 + 4 e rows (e = 0 .. 7; HieCoAtten's 196 image regions,
// hieCoAtten.py:25,30,35 and the input gradient of :30,35).
//
// Why: BASELINE config 4 (B = 256) multiplies 50176 x 512 outputs.  On 256 x 256 tiles that is 392 tiles = 1.53 rounds of the
// 256 CUs: a second round at half occupancy, or a row split whose remainder runs on the 128 x 128 kernel at 92 TF (round 3/4:
// 0.87-0.92 ms for the img_emb product against 0.67 at the MFMA peak).  256 samples x 196 rows is an exact fit instead: ONE
// workgroup per CU owns a sample's 196 rows x 256 columns -- NS x N / 256 work items, a whole number of rounds for NS = 256 --
// with no partial round and no hand-over between kernels.  196 = 6 x 32 + 4: six row tiles of v_mfma_f32_32x32x2_f32 (3 per
// wave half) and the last four rows on v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4 x 4: four rows x 64 columns per instruction,
// the same 64 FLOP / clk / SIMD), so nothing is padded: 2 % of the MFMA time for the ragged rows instead of 14 % for a seventh
// 32-row tile.
//
// Structure: gemm_f32_big.hip's -- LDS-DMA staging (global_load_lds_dwordx4) of 16-k slabs into five 32 KB slots, four slabs in
// flight, counted vmcnt + one raw s_barrier per slab, XOR-swizzled K-contiguous images, fragment reads one k-step ahead.  8 waves =
// 2 row halves (96 rows = 3 MFMA row tiles) x 4 column strips (64 columns = 2 MFMA column tiles, columns interleaved 2 c + j so
// that a lane's two column tiles are 8 contiguous output bytes); waves 0-3 additionally own the four ragged rows of column group
// `wave`.  Every output element is one k-ordered fmaf chain in the slab order of the other fp32 kernels (k = 8 ks + e, 8 ks + 4
// + e), the ragged rows included: the same bits as gemm_f32.hip / gemm_f32_big.hip on the same operands.
// B: K-contiguous (N, K) (forward products) or K-major (K, N) (input gradients).  Preconditions: N % 256 == 0, K % 16 == 0,
// 16-byte aligned bases, lda / ldb / ldc % 4 == 0.
#include "common.h"

namespace {

typedef const float __attribute__((address_space(1))) gfloat;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TK = 16, NT = 512, TN = 256, RA = 256;     // RA: A rows staged per slab (a sample's L <= 220 rows + the next sample's first ones, unused)
constexpr int ROW_B = TK * 4, CH = ROW_B / 16;           // K-contiguous image: 64-byte rows, 4 chunks of 16 bytes
constexpr int OP_BYTES = 256 * ROW_B;                    // 16 KB per operand per slab (either layout)
constexpr int SLOT_BYTES = 2 * OP_BYTES, NSLOT = 5, SMEM = NSLOT * SLOT_BYTES;
constexpr int NG = OP_BYTES / (NT * 16);                 // 2 LDS-DMA instructions per thread per operand per slab

struct SampleArgs {
  const float* A; const float* B; float* C; const float* bias;
  int NS, L, M, N, K, lda, ldb, ldc, flags, tiles_n;
};

__device__ __forceinline__ int swz(int r) { return (r >> 2) & 3; }

// per-lane global source pointers of the NG copies of one operand slab (gemm_f32_big.hip init_src):
//   A (K-contiguous): copy i, wave w, lane l -> LDS row rho = i*128 + 16w + (l >> 2), source chunk (l & 3) ^ swz(rho), source row
//     m0 + rho clamped to M - 1 (rows >= L belong to the next sample and are never multiplied).
//   B K-contiguous: the same with the columns of each 64-column wave strip INTERLEAVED: LDS row 32 j + c of a strip holds column
//     2 c + j (MFMA column tile j of the strip = the columns = j mod 2).
//   B K-major: copy i, wave w, lane l -> k-row i*8 + w, floats 4l .. 4l+3 of that row.
template <bool KMAJOR, bool PERM>
__device__ __forceinline__ void init_src(gfloat* (&q)[NG], const float* base, int ld, int r0, int R, int k0, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    if (!KMAJOR) {
      const int rho = i * 128 + wave * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ swz(rho);
      const int row = PERM ? (rho & ~63) + 2 * (rho & 31) + ((rho >> 5) & 1) : rho;
      q[i] = (gfloat*)(base + (long long)min(r0 + row, R - 1) * ld + k0 + chunk * 4);
    } else {
      q[i] = (gfloat*)(base + (long long)(k0 + i * 8 + wave) * ld + r0 + lane * 4);
    }
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void stage_operand(gfloat* (&q)[NG], int ld, char* s, int wave) {
  typedef __attribute__((address_space(3))) char lds_char;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    lds_char* dst = (lds_char*)(s + (i * 8 + wave) * 1024);    // wave-uniform; the DMA adds lane * 16
    __builtin_amdgcn_global_load_lds(q[i], dst, 16, 0, 0);
    q[i] += KMAJOR ? (long long)TK * ld : TK;
  }
}

__device__ __forceinline__ void wait_copies(int later) {     // all but the 4 * later youngest LDS-DMA copies of this wave have landed
  if (later >= 3)      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// operand values of one k-step (8 k) for the wave's 3 row tiles / 2 column tiles; MFMA step (ks, e) multiplies k = 8 ks + e (lanes
// 0-31) and k = 8 ks + 4 + e (lanes 32-63), both operands alike
struct FragA {
  f32x4 f[3];                                              // [row tile] (e in the vector)
  __device__ __forceinline__ void load(const char* s, int row0, int ks, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      f[i] = *reinterpret_cast<const f32x4*>(s + (row0 + 32 * i + r) * ROW_B + (((2 * ks + h) ^ swz(r)) << 4));
  }
};
template <bool KMAJOR>
struct FragB {
  f32x4 fc[2];                                             // K-contiguous: [column tile] (e in the vector)
  f32x2 ft[4];                                             // K-major: [e] (column tile in the vector)
  __device__ __forceinline__ void load(const char* s, int col0, int ks, int lane) {
    const int r = lane & 31, h = lane >> 5;
    if (KMAJOR) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ft[e] = *reinterpret_cast<const f32x2*>(s + (8 * ks + 4 * h + e) * 1024 + (col0 + 2 * r) * 4);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fc[j] = *reinterpret_cast<const f32x4*>(s + (col0 + 32 * j + r) * ROW_B + (((2 * ks + h) ^ swz(r)) << 4));
    }
  }
  __device__ __forceinline__ float v(int j, int e) const { return KMAJOR ? ft[e][j] : fc[j][e]; }
};

template <bool TB>
__global__ void __launch_bounds__(NT, 2) gemm_f32_sample_kernel(const SampleArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 1, wc = wave >> 1;                 // rows wr*96 .. +95 of the sample, columns wc*64 .. +63 of the tile
  const int nrag = (g.L - 192) >> 2;                       // ragged 4-row groups (196 rows: one), all owned by waves 0-3 (column group = wave)
  const int S = g.K / TK;
  const int items = g.NS * g.tiles_n;

  for (int w = blockIdx.x; w < items; w += gridDim.x) {
    // column tile slowest: with a whole number of rounds a workgroup meets the same sample again for its next column tile
    const int tn = w / g.NS, n = w - tn * g.NS;
    const int m0 = n * g.L, n0 = tn * TN;
    gfloat* qa[NG];
    gfloat* qb[NG];
    init_src<false, false>(qa, g.A, g.lda, m0, g.M, 0, wave, lane);
    init_src<TB, true>(qb, g.B, g.ldb, n0, g.N, 0, wave, lane);
    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 rag[8];                                          // ragged rows: [4-row group] x (rows in the vector), column = 64 wave + lane
#pragma unroll
    for (int u = 0; u < 8; ++u) rag[u] = f32x4{0.f, 0.f, 0.f, 0.f};

    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): the previous item's output stores are out of the counted copy waits below
    __builtin_amdgcn_s_barrier();                          // every wave has left the previous item's K loop: the ring is free
#pragma unroll
    for (int p = 0; p < NSLOT - 1; ++p)
      if (p < S) {
        stage_operand<false>(qa, g.lda, smem + p * SLOT_BYTES, wave);
        stage_operand<TB>(qb, g.ldb, smem + p * SLOT_BYTES + OP_BYTES, wave);
      }
    int slot = 0;
    for (int s = 0; s < S; ++s) {
      wait_copies(min(NSLOT - 2, S - 1 - s));              // my copies of slab s; later slabs stay in flight
      __builtin_amdgcn_s_barrier();
      const char* sA = smem + slot * SLOT_BYTES;
      const char* sB = sA + OP_BYTES;
      FragA fa[2];
      FragB<TB> fb[2];
      fa[0].load(sA, wr * 96, 0, lane);
      fb[0].load(sB, wc * 64, 0, lane);
      if (s + NSLOT - 1 < S) {                             // refill the slot of slab s-1 (every read of it precedes this barrier)
        const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
        stage_operand<false>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
        stage_operand<TB>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 0) {
          fa[1].load(sA, wr * 96, 1, lane);
          fb[1].load(sB, wc * 64, 1, lane);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks].f[i][e], fb[ks].v(j, e), acc[i][j], 0, 0, 0);
      }
      // the ragged rows 192 + 4u .. +3 of the sample x the 64 columns of group `wave` on v_mfma_f32_4x4x1_16B_f32: lane l = block
      // l / 4, A value = row l % 4 (the same four rows in every block), B value = column l of the group, D = 4 rows x column l.
      // k in the order of the 32x32x2 chain: 8 ks + e, 8 ks + 4 + e.
      if (wave < 4 && nrag > 0) {
        f32x4 bq[4];                                       // B[column][k = 4 c .. 4 c + 3], c = 0..3
        if (TB) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) bq[c][e] = *reinterpret_cast<const float*>(sB + (4 * c + e) * 1024 + (wave * 64 + lane) * 4);
        } else {
          const int rl = wave * 64 + 32 * (lane & 1) + (lane >> 1);          // LDS row of column `lane` of the group (interleaved strip)
#pragma unroll
          for (int c = 0; c < 4; ++c) bq[c] = *reinterpret_cast<const f32x4*>(sB + rl * ROW_B + ((c ^ swz(rl)) << 4));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (u < nrag) {
            const int ra = 192 + 4 * u + (lane & 3);
            f32x4 aq[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) aq[c] = *reinterpret_cast<const f32x4*>(sA + ra * ROW_B + ((c ^ swz(ra)) << 4));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                rag[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[2 * ks][e], bq[2 * ks][e], rag[u], 0, 0, 0);
                rag[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[2 * ks + 1][e], bq[2 * ks + 1][e], rag[u], 0, 0, 0);
              }
          }
      }
      slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
    }

    // ---- epilogue: accumulator tile (i, j), register e, lane (cm, h): row wr*96 + 32 i + (e & 3) + 8 (e >> 2) + 4 h, columns
    // wc*64 + 2 cm + j -> one 8-byte store per (i, e)
    const bool relu = g.flags & VQF_GEMM_RELU;
    {
      const int cm = lane & 31, h = lane >> 5;
      const int col = n0 + wc * 64 + 2 * cm;
      float bv[2] = {0.f, 0.f};
      if (g.bias) { bv[0] = g.bias[col]; bv[1] = g.bias[col + 1]; }
      __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0) for the biases, once, as a builtin (gemm_f32_big.hip store_tile)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wr * 96 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
          float v0 = acc[i][0][e] + bv[0], v1 = acc[i][1][e] + bv[1];
          if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          *reinterpret_cast<f32x2*>(g.C + (long long)row * g.ldc + col) = f32x2{v0, v1};
        }
    }
    if (wave < 4 && nrag > 0) {
      const int col = n0 + wave * 64 + lane;
      const float bvr = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < nrag) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = rag[u][r] + bvr;
            if (relu) v = fmaxf(v, 0.f);
            g.C[(long long)(m0 + 192 + 4 * u + r) * g.ldc + col] = v;
          }
        }
    }
  }
}

}  // namespace

extern "C" {

// 1 when vqf_gemm_f32_sample takes this shape
int vqf_gemm_f32_sample_supported(int NS, int L, int N, int K) {
  const int sw = vqf_opt(VQF_OPT_GEMM_F32_SAMPLE, 1);      // 0 = never, 2 = wherever the kernel CAN run (tests), default: where it pays
  if (!(NS > 0 && L >= 192 && L <= 220 && (L % 4) == 0 && N >= TN && (N % TN) == 0 && K >= 4 * TK && (K % TK) == 0 &&
        (long long)NS * L < (1LL << 31) && sw != 0))
    return 0;
  // one workgroup per (sample, 256 columns): worth it once the items fill at least half of the CUs (a small batch has more
  // parallelism in the 128x128 tiles of vqf_gemm_f32)
  return sw == 2 || (long long)NS * (N / TN) * 2 >= vqf_cu_count();
}

int vqf_gemm_f32_sample(int tb, int NS, int L, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                        const float* bias, int flags, void* stream) {
  if (!A || !B || !C || lda < K || ldc < N || (tb ? ldb < N : ldb < K)) return VQF_E_BADARG;
  if (!vqf_gemm_f32_sample_supported(NS, L, N, K) || (flags & ~VQF_GEMM_RELU)) return VQF_E_UNSUPPORTED;
  if (!aligned16(A) || !aligned16(B) || (((uintptr_t)C) & 7) || (lda % 4) || (ldb % 4) || (ldc % 2)) return VQF_E_ALIGN;
  SampleArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.NS = NS; g.L = L; g.M = NS * L; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.flags = flags;
  g.tiles_n = N / TN;
  const int items = NS * g.tiles_n;
  int cus = vqf_cu_count();
  const int lim = vqf_opt(VQF_OPT_GEMM_CU_LIMIT, 0) & ~7;
  if (lim >= 8 && lim < cus) cus = lim;
  const int nwg = items < cus ? items : cus;
  hipStream_t s = (hipStream_t)stream;
  vqf_prof_dims(g.M, N, K);
  vqf_stat_bump(VQF_STAT_GEMM_F32_SAMPLE);
  static VqfDynLdsFlags attr0 = {}, attr1 = {};
  if (tb) {
    if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_sample_kernel<true>), SMEM, attr1)) return e;
    VQF_LAUNCH(KID_GEMM_A0B1, gemm_f32_sample_kernel<true>, dim3(nwg), dim3(NT), SMEM, s, g);
  } else {
    if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_sample_kernel<false>), SMEM, attr0)) return e;
    VQF_LAUNCH(KID_GEMM_A0B0, gemm_f32_sample_kernel<false>, dim3(nwg), dim3(NT), SMEM, s, g);
  }
  return vqf_last_error();
}

}  // extern "C"
