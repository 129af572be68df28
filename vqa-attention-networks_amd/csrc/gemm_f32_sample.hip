// Per-sample-tile fp32 GEMM (gfx950): C[m, n] = sum_k A[m, k] * Bop[n, k] (+ bias[n]) (relu) for row-major A (M = NS * L rows,
// K contiguous) whose rows come in SAMPLES of L = 192 + 4 e rows (e = 0 or 1; HieCoAtten's 196 image regions,
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

constexpr int TK = 16, NT = 512, TN = 256;               // (256 A rows are staged per slab: a sample's L <= 200 + the next sample's first ones, unused)
constexpr int ROW_B = TK * 4;                            // K-contiguous image: 64-byte rows, 4 chunks of 16 bytes
constexpr int OP_BYTES = 256 * ROW_B;                    // 16 KB per operand per slab (either layout)
constexpr int SLOT_BYTES = 2 * OP_BYTES, NSLOT = 5, SMEM = NSLOT * SLOT_BYTES;
constexpr int NG = OP_BYTES / (NT * 16);                 // 2 LDS-DMA instructions per thread per operand per slab
constexpr int MAXRAG = 1;                                // ragged 4-row groups per sample: L = 192 or 196 (its operands stay in registers for a whole slab)

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

// The same for slab `sidx` of an item whose first four slabs were issued AHEAD of the previous item's 48 main output stores
// (`behind_stores`): vector-memory operations complete in issue order, so while sidx < 4 the operations younger than slab sidx's
// copies are the rest of those four slabs, the 48 stores and the refills of this loop so far -- 60 in every case (see the kernel).
__device__ __forceinline__ void wait_slab(int sidx, int later, bool behind_stores) {
  if (behind_stores && sidx < NSLOT - 1) asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
  else wait_copies(later);
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
  const int late = wave >> 2;                              // waves 4-7: the half that runs half a slab behind (they share SIMDs with waves 0-3)
  const int nrag = (g.L - 192) >> 2;                       // ragged 4-row groups (196 rows: one) ...
  // ... x four 64-column groups, one per SIMD and two per wave half: group rg on wave 0, 1 (early half), 6, 7 (late half)
  const int rg = wave < 2 ? wave : wave - 4;
  const bool ragged = nrag > 0 && (wave < 2 || wave >= 6);
  const int S = g.K / TK;
  const int items = g.NS * g.tiles_n;

  // The slab ring runs on across a workgroup's items: when a wave leaves an item's K loop every wave has passed that loop's last
  // barrier, i.e. has finished reading every slab but the last one, so the four slots that do not hold the last slab are free and
  // the NEXT item's first four slabs are issued from this item's epilogue, AHEAD of its main output stores (see there); its fifth
  // slab follows behind the next loop's first barrier, which every wave reaches with its K loop behind it.
  int slot = 0;                                            // ring slot of the current item's slab 0, then of slab s
  bool behind = false;                                     // this item's slabs 0-3 were issued ahead of the previous item's main stores
  gfloat* qa[NG];
  gfloat* qb[NG];
  auto begin_item = [&](int w) {
    const int tn = w / g.NS, n = w - tn * g.NS;
    init_src<false, false>(qa, g.A, g.lda, n * g.L, g.M, 0, wave, lane);
    init_src<TB, true>(qb, g.B, g.ldb, tn * TN, g.N, 0, wave, lane);
    int sl = slot;
#pragma unroll
    for (int p = 0; p < NSLOT - 1; ++p) {
      if (p < S) {
        stage_operand<false>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
        stage_operand<TB>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
      }
      sl = (sl + 1 == NSLOT) ? 0 : sl + 1;
    }
  };
  if ((int)blockIdx.x < items) begin_item(blockIdx.x);
  for (int w = blockIdx.x; w < items; w += gridDim.x) {
    // column tile slowest: with a whole number of rounds a workgroup meets the same sample again for its next column tile
    const int tn = w / g.NS, n = w - tn * g.NS;
    const int m0 = n * g.L, n0 = tn * TN;
    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 rag[MAXRAG];                                          // ragged rows: [4-row group] x (rows in the vector), column = 64 rg + lane
#pragma unroll
    for (int u = 0; u < MAXRAG; ++u) rag[u] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The ragged rows 192 + 4u .. +3 of the sample x the 64 columns of group rg on v_mfma_f32_4x4x1_16B_f32: lane l = block l / 4, A
    // value = row l % 4 (the same four rows in every block), B value = column l of the group, D = 4 rows x column l; k in the order of
    // the 32x32x2 chain (8 ks + e, 8 ks + 4 + e).  rag_load reads the slab's operands up front.
    f32x4 bq[4], aq[MAXRAG][4];
    auto rag_load = [&](const char* sA, const char* sB) {
      if (TB) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) bq[c][e] = *reinterpret_cast<const float*>(sB + (4 * c + e) * 1024 + (rg * 64 + lane) * 4);
      } else {
        const int rl = rg * 64 + 32 * (lane & 1) + (lane >> 1);              // LDS row of column `lane` of the group (interleaved strip)
#pragma unroll
        for (int c = 0; c < 4; ++c) bq[c] = *reinterpret_cast<const f32x4*>(sB + rl * ROW_B + ((c ^ swz(rl)) << 4));
      }
#pragma unroll
      for (int u = 0; u < MAXRAG; ++u)
        if (u < nrag) {
          const int ra = 192 + 4 * u + (lane & 3);
#pragma unroll
          for (int c = 0; c < 4; ++c) aq[u][c] = *reinterpret_cast<const f32x4*>(sA + ra * ROW_B + ((c ^ swz(ra)) << 4));
        }
    };

    // one k-step (8 k) of the wave's 3 x 2 MFMA tiles; RAG: the ragged rows' two 4x4x1 MFMAs of (ks, e) ride behind the six 32x32x2
    // of the same (ks, e) -- each accumulates into the SAME four registers as the one before it, and issued back to back at the
    // end of the slab the 16-deep dependent chain left the matrix pipe idle while the SIMD's other wave sat in the barrier
    auto mma_kstep = [&](const FragA& fa, const FragB<TB>& fb, int ks, bool rag_on) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.f[i][e], fb.v(j, e), acc[i][j], 0, 0, 0);
        if (rag_on) {
#pragma unroll
          for (int u = 0; u < MAXRAG; ++u)
            if (u < nrag) {
              rag[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[u][2 * ks][e], bq[2 * ks][e], rag[u], 0, 0, 0);
              rag[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[u][2 * ks + 1][e], bq[2 * ks + 1][e], rag[u], 0, 0, 0);
            }
        }
      }
    };
    // K loop, gemm_f32_big.hip's staggered form: waves 4-7 run HALF A SLAB behind waves 0-3, so that while one half of a SIMD's two
    // waves sits in the per-slab barrier / its first fragment reads the other is in the middle of its MFMAs.  One barrier per slab:
    // waves 0-3 execute it at the START of slab s (behind their vmcnt for slab s), waves 4-7 in the MIDDLE of slab s-1 (behind
    // lgkmcnt(0) for all their reads of slab s-1 and their vmcnt for slab s).  Every wave's copies of slab s are waited for in front
    // of that barrier and every read of slab s comes after it; the slot of slab s-1 is refilled (slab s+4) behind it, after every
    // read of slab s-1 (waves 0-3: consumed by MFMAs issued before the barrier; waves 4-7: the lgkmcnt(0)).
    if (!late) {
      for (int s = 0; s < S; ++s) {
        wait_slab(s, min(NSLOT - 2, S - 1 - s), behind);   // my copies of slab s; later slabs (and the last item's stores) stay in flight
        __builtin_amdgcn_s_barrier();
        const char* sA = smem + slot * SLOT_BYTES;
        const char* sB = sA + OP_BYTES;
        FragA fa[2];
        FragB<TB> fb[2];
        fa[0].load(sA, wr * 96, 0, lane);
        fb[0].load(sB, wc * 64, 0, lane);
        if (s + NSLOT - 1 < S) {                           // slab s+4 into the slot of slab s-1
          const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
          stage_operand<false>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
          stage_operand<TB>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
        }
        fa[1].load(sA, wr * 96, 1, lane);
        fb[1].load(sB, wc * 64, 1, lane);
        if (ragged) rag_load(sA, sB);
        mma_kstep(fa[0], fb[0], 0, ragged);
        mma_kstep(fa[1], fb[1], 1, ragged);
        slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
      }
      __builtin_amdgcn_s_barrier();                        // pairs with the mid-slab barrier of waves 4-7 in their last slab
    } else {
      wait_slab(0, min(NSLOT - 1, S) - 1, behind);         // my copies of slab 0
      __builtin_amdgcn_s_barrier();                        // pairs with the slab-0 barrier of waves 0-3
      for (int s = 0; s < S; ++s) {
        const char* sA = smem + slot * SLOT_BYTES;
        const char* sB = sA + OP_BYTES;
        FragA fa[2];
        FragB<TB> fb[2];
        fa[0].load(sA, wr * 96, 0, lane);
        fb[0].load(sB, wc * 64, 0, lane);
        if (s + NSLOT - 1 < S) {                           // slab s+4 into the slot of slab s-1
          const int sl = (slot == 0) ? NSLOT - 1 : slot - 1;
          stage_operand<false>(qa, g.lda, smem + sl * SLOT_BYTES, wave);
          stage_operand<TB>(qb, g.ldb, smem + sl * SLOT_BYTES + OP_BYTES, wave);
        }
        fa[1].load(sA, wr * 96, 1, lane);
        fb[1].load(sB, wc * 64, 1, lane);
        if (ragged) rag_load(sA, sB);
        mma_kstep(fa[0], fb[0], 0, ragged);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): every read of slab s by this wave has returned
        wait_slab(s + 1, min(s + NSLOT - 1, S - 1) - (s + 1), behind);      // my copies of slab s+1; later slabs stay in flight
        __builtin_amdgcn_s_barrier();                      // pairs with the slab-(s+1) barrier of waves 0-3 (their final one for s = S-1)
        __builtin_amdgcn_sched_barrier(0);
        mma_kstep(fa[1], fb[1], 1, ragged);                // (ragged operands in registers: read before the lgkmcnt(0) above)
        slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
      }
    }

    // ---- epilogue.  Order matters: vector-memory operations retire in issue order and the copy waits count ALL of them, so the next
    // item's first four slabs are issued AHEAD of this item's 48 main output stores (behind the biases and the few ragged-row
    // stores): its K loop then starts as soon as those copies have landed, while the stores drain behind them -- issued behind the
    // stores (round-5 first form) the first slab waited for every store of the previous item, ~8 us per item.  The four slots
    // written are free: every wave has passed the K loop's last barrier, i.e. finished reading every slab but the last one.
    // accumulator tile (i, j), register e, lane (cm, h): row wr*96 + 32 i + (e & 3) + 8 (e >> 2) + 4 h, columns wc*64 + 2 cm + j ->
    // one 8-byte store per (i, e)
    const bool relu = g.flags & VQF_GEMM_RELU;
    const int cm = lane & 31, h = lane >> 5;
    const int col = n0 + wc * 64 + 2 * cm;
    float bv[2] = {0.f, 0.f};
    float bvr = 0.f;
    if (g.bias) {
      bv[0] = g.bias[col]; bv[1] = g.bias[col + 1];
      if (ragged) bvr = g.bias[n0 + rg * 64 + lane];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) for the biases, once, as a builtin (gemm_f32_big.hip store_tile)
    if (ragged) {
      const int colr = n0 + rg * 64 + lane;
#pragma unroll
      for (int u = 0; u < MAXRAG; ++u)
        if (u < nrag) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = rag[u][r] + bvr;
            if (relu) v = fmaxf(v, 0.f);
            g.C[(long long)(m0 + 192 + 4 * u + r) * g.ldc + colr] = v;
          }
        }
    }
    behind = w + (int)gridDim.x < items;
    if (behind) begin_item(w + gridDim.x);                 // the next item's slabs 0-3, AHEAD of the 48 stores below
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
}

}  // namespace

extern "C" {

// 1 when vqf_gemm_f32_sample takes this shape
int vqf_gemm_f32_sample_supported(int NS, int L, int N, int K) {
  const int sw = vqf_opt(VQF_OPT_GEMM_F32_SAMPLE, 1);      // 0 = never, 2 = wherever the kernel CAN run (tests), default: where it pays
  if (!(NS > 0 && L >= 192 && L <= 192 + 4 * MAXRAG && (L % 4) == 0 && N >= TN && (N % TN) == 0 && K >= 4 * TK && (K % TK) == 0 &&
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
