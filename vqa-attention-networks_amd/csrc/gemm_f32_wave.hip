// Small-M fp32 GEMM (gfx950 only): the products whose output has too few 128x128 tiles to fill 256 CUs
// -- above all the 26 recurrent products of the question encoder's LSTM per train step,
//    gates_t += h_{t-1} W_hh^T   (512 x 4096 x 1024, accumulate)        mfb.py:69 via host/functions.py::LstmBatchFn
//    dh_{t-1} = dG_t W_hh        (512 x 1024 x 4096, B K-major)
// which the 128x128 kernel ran as 128 tiles x 4 K-splits + a slab-reduce LAUNCH each (62 us per product, 43
// splitk_reduce launches per step).
//
//   C[m,n] (+)= sum_k A[m,k] * Bop[n,k] (+ bias[n]) (relu)     A (M,K) K-contiguous; B (N,K) "tb = 0" or (K,N) "tb = 1".
//
// Every WAVE owns a 32 x 64 output tile (1 x 2 tiles of v_mfma_f32_32x32x2_f32, 32 accumulator registers) over a K
// range and runs its OWN six-slot LDS-DMA pipeline (K slabs of 16, five in flight, counted vmcnt): there is no
// workgroup barrier in the K loop, the four waves of a workgroup (one per SIMD) drift apart freely, and a 64-cycle
// MFMA stream with fragment reads one k-step ahead keeps each SIMD's matrix pipe fed from a single wave.
//   WK = 1: the workgroup's 4 waves own 4 different tiles (same 64 columns, 4 consecutive row tiles), whole K each.
//   WK = 4: the 4 waves own ONE tile and a quarter of K each; their accumulators are summed through LDS in a fixed
//           order (wave 0 + 1 + 2 + 3): deterministic, no slab round trip through HBM, no second launch.
// The host picks the form that gives ~256 workgroups (one per CU).  LDS images: K-contiguous operands [row][16] floats
// with the chunk XOR (row >> 2) & 3 of gemm_f32_big.hip on the copy's source address and on the ds_read_b128; a
// K-major B as [k][64] floats read with ds_read_b64.  B's 64 columns are interleaved over the two column tiles
// (column 2c + j belongs to tile j), by a row permutation on the copy's source address for a K-contiguous B, so a lane's
// two accumulator tiles are 8 contiguous output bytes.
// Preconditions (else the caller uses gemm_f32.hip): A K-contiguous, K % 16 == 0 (K % 64 for WK = 4), 16-byte
// aligned bases, lda/ldb % 4 == 0, N % 4 == 0 for a K-major B.
#include "common.h"
#include <stdlib.h>

namespace {

typedef const float __attribute__((address_space(1))) gfloat;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WTM = 32, WTN = 64, TK = 16, NSLOT = 6, NT = 256;
constexpr int A_BYTES = WTM * TK * 4, B_BYTES = WTN * TK * 4, SLOT_BYTES = A_BYTES + B_BYTES;   // 2 + 4 KB
constexpr int WAVE_LDS = NSLOT * SLOT_BYTES;           // 36 KB per wave
constexpr int SMEM_WAVE = 4 * WAVE_LDS;                // 144 KB per workgroup: one workgroup per CU
constexpr int NGA = A_BYTES / 1024, NGB = B_BYTES / 1024, NGL = NGA + NGB;   // 2 + 4 copies per lane and slab

struct WaveArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K, lda, ldb, ldc, flags;
  int tiles_m, tiles_n, kpart;       // kpart: K range per wave (K for WK = 1, K / 4 for WK = 4)
  // fused LSTM step (gemm_f32_lstm_fwd_kernel): C = the step's gate buffer (B, 4H), N = 4H
  const float* c_prev;               // (B, H) or nullptr at the first step
  float* c_out;
  float* h_out;
  int H;
};

__device__ __forceinline__ int swz(int r) { return (r >> 2) & 3; }

template <bool TB>
__device__ __forceinline__ void init_src(gfloat* (&q)[NGL], const WaveArgs& g, int m0, int n0, int k0, int lane) {
#pragma unroll
  for (int i = 0; i < NGA; ++i) {                      // A: copy i -> rows 16 i + (l >> 2), LDS chunk l & 3
    const int row = 16 * i + (lane >> 2);
    const int chunk = (lane & 3) ^ swz(row);
    q[i] = (gfloat*)(g.A + (long long)min(m0 + row, g.M - 1) * g.lda + k0 + chunk * 4);
  }
#pragma unroll
  for (int i = 0; i < NGB; ++i) {
    if (!TB) {                                         // B (N,K): LDS row rho = 16 i + (l >> 2) holds column 2 (rho & 31) + (rho >> 5)
      const int rho = 16 * i + (lane >> 2);
      const int chunk = (lane & 3) ^ swz(rho);
      const int col = 2 * (rho & 31) + (rho >> 5);
      q[NGA + i] = (gfloat*)(g.B + (long long)min(n0 + col, g.N - 1) * g.ldb + k0 + chunk * 4);
    } else {                                           // B (K,N): copy i -> k-rows 4 i + (l >> 4), columns 4 (l & 15) .. +3
      const int k = 4 * i + (lane >> 4);
      q[NGA + i] = (gfloat*)(g.B + (long long)(k0 + k) * g.ldb + min(n0 + 4 * (lane & 15), g.N - 4));
    }
  }
}

template <bool TB>
__device__ __forceinline__ void stage(gfloat* (&q)[NGL], const WaveArgs& g, char* slot) {
  typedef __attribute__((address_space(3))) char lds_char;
#pragma unroll
  for (int i = 0; i < NGL; ++i) {
    __builtin_amdgcn_global_load_lds(q[i], (lds_char*)(slot + i * 1024), 16, 0, 0);   // wave-uniform base; the DMA adds lane * 16
    q[i] += (i >= NGA && TB) ? (long long)TK * g.ldb : TK;
  }
}

// k-step ks (8 k) of a slab: MFMA step e multiplies k = 8ks + e (lanes 0-31) and k = 8ks + 4 + e (lanes 32-63)
template <bool TB>
struct Frag {
  f32x4 a;             // row r, k = 8ks + 4h + e
  f32x4 bc[2];         // K-contiguous B: [tile] (e in the vector)
  f32x2 bt[4];         // K-major B: [e] (tile in the vector)
  __device__ __forceinline__ void load(const char* slot, int ks, int lane) {
    const int r = lane & 31, h = lane >> 5;
    a = *reinterpret_cast<const f32x4*>(slot + r * 64 + (((2 * ks + h) ^ swz(r)) << 4));
    const char* sb = slot + A_BYTES;
    if (!TB) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bc[j] = *reinterpret_cast<const f32x4*>(sb + (32 * j + r) * 64 + (((2 * ks + h) ^ swz(r)) << 4));   // swz(32j + r) == swz(r)
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) bt[e] = *reinterpret_cast<const f32x2*>(sb + (8 * ks + 4 * h + e) * 256 + r * 8);
    }
  }
  __device__ __forceinline__ float b(int j, int e) const { return TB ? bt[e][j] : bc[j][e]; }
};

__device__ __forceinline__ void wait_copies(int later) {   // all but the 6 * later youngest copies of this wave have landed
  if (later >= 4)      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (later == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if (later == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (later == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// one output row's two values of this lane: (+bias) (+C) (relu), 8-byte store when the row pitch allows it.
// VQF_GEMM_ACCUM: the old C values arrive in (c0, c1) -- fetched by c_prefetch() at kernel ENTRY.  Loading them here, between
// the stores of the previous rows, made the epilogue a chain of 16 dependent memory round trips (the compiler cannot move a
// load across a store it cannot disambiguate): 8 us of the LSTM's 58-us forward product.
__device__ __forceinline__ void put2(const WaveArgs& g, int row, int col, float v0, float v1, float b0, float b1, bool vec,
                                     float c0 = 0.f, float c1 = 0.f) {
  if (row >= g.M || col >= g.N) return;
  float* pc = g.C + (long long)row * g.ldc + col;
  const bool two = col + 1 < g.N;
  v0 += b0; v1 += b1;
  if (g.flags & VQF_GEMM_ACCUM) { v0 += c0; v1 += c1; }
  if (g.flags & VQF_GEMM_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
  if (vec && two) *reinterpret_cast<f32x2*>(pc) = f32x2{v0, v1};
  else { pc[0] = v0; if (two) pc[1] = v1; }
}

// the lane's 16 x 2 old C values of its 32 x 64 tile (rows (e & 3) + 8 (e >> 2) + 4h, columns 2 cm, 2 cm + 1): zeros unless
// VQF_GEMM_ACCUM; issued before the K loop, consumed after it
__device__ __forceinline__ void c_prefetch(const WaveArgs& g, int m0, int n0, int lane, bool vec, f32x2 (&cold)[16]) {
  const int cm = lane & 31, h = lane >> 5, col = n0 + 2 * cm;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    cold[e] = f32x2{0.f, 0.f};
    const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if ((g.flags & VQF_GEMM_ACCUM) && row < g.M && col < g.N) {
      const float* pc = g.C + (long long)row * g.ldc + col;
      if (vec && col + 1 < g.N) cold[e] = *reinterpret_cast<const f32x2*>(pc);
      else { cold[e][0] = pc[0]; if (col + 1 < g.N) cold[e][1] = pc[1]; }
    }
  }
}

template <bool TB, int WK>
__global__ void __launch_bounds__(NT, 1) gemm_f32_wave_kernel(const WaveArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // tile order: row tiles fastest, so the 4 waves of a WK = 1 workgroup (and neighbouring workgroups) stream the same
  // 64 columns of B; blocks b and b + 8 share an XCD: XCD-aware bijective remap of the workgroup index first
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int t = (WK == 1) ? 4 * wg + wave : wg;        // this wave's tile
  const int ntiles = g.tiles_m * g.tiles_n;
  const bool live = t < ntiles;                        // WK = 1: the last workgroup may have idle waves (wave-uniform)
  const int tm = live ? t % g.tiles_m : 0, tn = live ? t / g.tiles_m : 0;
  const int m0 = tm * WTM, n0 = tn * WTN;
  const int k0 = (WK == 1) ? 0 : wave * g.kpart;
  const int S = live ? g.kpart / TK : 0;
  char* my = smem + wave * WAVE_LDS;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  gfloat* q[NGL];
  init_src<TB>(q, g, m0, n0, k0, lane);
#pragma unroll
  for (int p = 0; p < NSLOT - 1; ++p)
    if (p < S) stage<TB>(q, g, my + p * SLOT_BYTES);
  const bool vec = ((g.ldc & 1) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 7) == 0);
  f32x2 cold[16];                                      // WK = 1: the old C values of an accumulating product, in flight
  if (WK == 1 && live) c_prefetch(g, m0, n0, lane, vec, cold);   // behind the first slabs' copies (vmcnt is in order)
  // Fragments are double-buffered ACROSS slabs: the 6 LDS reads of slab s+1 are issued before the 16 MFMAs of slab s
  // (two named sets, loop unrolled by two: a runtime-indexed set would live in scratch).
  Frag<TB> fa[2], fb[2];
  int slot = 0;                                        // slot of slab s
  // One slab: its 16 MFMAs with the next slab's LDS reads and the refill copies issued BETWEEN them, one auxiliary
  // instruction group per 64-cycle MFMA (a single wave feeds its SIMD's matrix pipe: anything issued in a block in front
  // of the MFMAs -- 6 copies cost ~330 issue cycles -- would leave the pipe idle for that long).
  auto half = [&](int s, Frag<TB> (&cur)[2], Frag<TB> (&nxt)[2]) {
    const bool more = s + 1 < S, refill = s + NSLOT - 1 < S;
    if (more) wait_copies(min(NSLOT - 3, S - 2 - s));  // my copies of slab s+1 (issued 4 slabs ago); up to 3 later slabs stay in flight
    const char* nx = my + (slot + 1 == NSLOT ? 0 : slot + 1) * SLOT_BYTES;
    char* rf = my + ((slot == 0) ? NSLOT - 1 : slot - 1) * SLOT_BYTES;   // slot of slab s-1 (read two slabs ago) takes slab s+5
    typedef __attribute__((address_space(3))) char lds_char;
#pragma unroll
    for (int n = 0; n < 16; ++n) {                     // MFMA n: k-step n >> 3, step e = (n >> 1) & 3, column tile n & 1
      const int ks = n >> 3, e = (n >> 1) & 3, j = n & 1;
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[ks].a[e], cur[ks].b(j, e), acc[j], 0, 0, 0);
      if (n == 0 && more) nxt[0].load(nx, 0, lane);
      if (n == 2 && more) nxt[1].load(nx, 1, lane);
      if (n >= 4 && n < 4 + NGL && refill) {
        const int i = n - 4;
        __builtin_amdgcn_global_load_lds(q[i], (lds_char*)(rf + i * 1024), 16, 0, 0);
        q[i] += (i >= NGA && TB) ? (long long)TK * g.ldb : TK;
      }
      __builtin_amdgcn_sched_barrier(0);               // keep this interleave
    }
    // the reads of slab s+1 returned long ago: draining the counter HERE costs nothing and tells hipcc that no LDS read is
    // pending at the loop head (else it waits lgkmcnt(0) in front of the next slab's first MFMA)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
  };
  if (S > 0) {
    wait_copies(min(NSLOT - 2, S - 1));                // slab 0
    fa[0].load(my, 0, lane);
    fa[1].load(my, 1, lane);
    __builtin_amdgcn_sched_barrier(0);                 // (same reason as in half(): no LDS read pending at the loop head)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int s = 0; s < S; s += 2) {
    half(s, fa, fb);
    if (s + 1 < S) half(s + 1, fb, fa);
  }

  // ---- epilogue: tile j, register e, lane (cm, h): row (e & 3) + 8 (e >> 2) + 4h, column 2 cm + j
  const int cm = lane & 31, h = lane >> 5;
  if (WK == 1) {
    if (!live) return;
    const int col = n0 + 2 * cm;
    const float b0 = (g.bias && col < g.N) ? g.bias[col] : 0.f;
    const float b1 = (g.bias && col + 1 < g.N) ? g.bias[col + 1] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      put2(g, m0 + (e & 3) + 8 * (e >> 2) + 4 * h, col, acc[0][e], acc[1][e], b0, b1, vec, cold[e][0], cold[e][1]);
  } else {
    // partial tiles through LDS (each wave's own region, free once its last slab has been read): [e][lane] float2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x2* mine = reinterpret_cast<f32x2*>(my);
#pragma unroll
    for (int e = 0; e < 16; ++e) mine[e * 64 + lane] = f32x2{acc[0][e], acc[1][e]};
    __syncthreads();
    if (!live) return;
    const int col = n0 + 2 * cm;
    const float b0 = (g.bias && col < g.N) ? g.bias[col] : 0.f;
    const float b1 = (g.bias && col + 1 < g.N) ? g.bias[col + 1] : 0.f;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {                   // wave w finishes registers e = w, w + 4, w + 8, w + 12
      const int e = wave + 4 * qd;
      f32x2 v = reinterpret_cast<const f32x2*>(smem)[e * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {                    // fixed order: ((wave0 + wave1) + wave2) + wave3
        const f32x2 p = reinterpret_cast<const f32x2*>(smem + w * WAVE_LDS)[e * 64 + lane];
        v[0] += p[0]; v[1] += p[1];
      }
      float c0 = 0.f, c1 = 0.f;                          // (WK = 4 with accumulate: 4 rows per wave, loaded here)
      const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if ((g.flags & VQF_GEMM_ACCUM) && row < g.M && col < g.N) {
        c0 = g.C[(long long)row * g.ldc + col];
        if (col + 1 < g.N) c1 = g.C[(long long)row * g.ldc + col + 1];
      }
      put2(g, row, col, v[0], v[1], b0, b1, vec, c0, c1);
    }
  }
}

// ---- WK = 1 with the B panel staged ONCE per workgroup (round 3) ---------------------------------------------------------
// The four waves of a WK = 1 workgroup own four row tiles of the SAME 64 columns, and each used to stream its own copy of
// that B slab: 6 KB per wave and slab, 384 MB through L2 per 512 x 4096 x 1024 product for 27 MB of operands, i.e. the
// kernel ran at the rate a CU can pull from L2 / Infinity Cache (58 us, 0.47 of the MFMA peak).  Here a slab's B image is
// ONE 4 KB region per workgroup, filled a quarter per wave (copy index = wave), so a workgroup moves 12 KB per slab
// instead of 24.  The price is one s_barrier per slab: it sits in the MIDDLE of the previous slab's MFMA stream (behind
// the wave's own counted vmcnt for the next slab), so the matrix pipe keeps draining while the waves meet.
//   RAW: a wave reads slab s+1's B only behind that barrier, which every wave enters after its own copies of slab s+1.
//   WAR: the slot of slab s-1 is refilled (slab s+5) during slab s, i.e. behind the barrier of slab s-1, which every wave
//        enters after lgkmcnt(0) for its reads of slab s-1 (issued and drained during slab s-2).
// NSL slots per operand ring (NSL - 1 slabs in flight): 12 KB of LDS per slot and workgroup (4 x 2 KB private A + 4 KB shared B)
constexpr int NGS = NGA + 1;                            // copies per lane and slab: 2 (own A rows) + 1 (a quarter of B)

__device__ __forceinline__ void wait_copies3(int later) {   // all but the 3 * later youngest copies of this wave have landed
  switch (later < 0 ? 0 : later) {                     // (vmcnt is a 6-bit immediate: 3 * 10 = 30 copies fit)
    case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1:  asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 2:  asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 3:  asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 4:  asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 5:  asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 6:  asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 7:  asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 8:  asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 9:  asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
  }
}

template <bool TB>
struct FragS {                                         // Frag with A and B in different LDS regions
  f32x4 a;
  f32x4 bc[2];
  f32x2 bt[4];
  __device__ __forceinline__ void load(const char* sa, const char* sb, int ks, int lane) {
    const int r = lane & 31, h = lane >> 5;
    a = *reinterpret_cast<const f32x4*>(sa + r * 64 + (((2 * ks + h) ^ swz(r)) << 4));
    if (!TB) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bc[j] = *reinterpret_cast<const f32x4*>(sb + (32 * j + r) * 64 + (((2 * ks + h) ^ swz(r)) << 4));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) bt[e] = *reinterpret_cast<const f32x2*>(sb + (8 * ks + 4 * h + e) * 256 + r * 8);
    }
  }
  __device__ __forceinline__ float b(int j, int e) const { return TB ? bt[e][j] : bc[j][e]; }
};

template <bool TB, int NSL>
__global__ void __launch_bounds__(NT, 1) gemm_f32_wave_shb_kernel(const WaveArgs g) {
  constexpr int SHB_A = NSL * A_BYTES;
  static_assert(NSL >= 3 && NSL <= 12, "at most 10 later slabs in the vmcnt ladder");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int t = 4 * wg + wave;                         // tiles_m % 4 == 0 (host): the 4 waves share the column tile
  const int tm = t % g.tiles_m, tn = t / g.tiles_m;
  const int m0 = tm * WTM, n0 = tn * WTN;
  const int S = g.kpart / TK;
  char* myA = smem + wave * SHB_A;
  char* shB = smem + 4 * SHB_A;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  gfloat* q[NGS];                                      // [0..1] my A rows, [2] my quarter of the workgroup's B slab
#pragma unroll
  for (int i = 0; i < NGA; ++i) {
    const int row = 16 * i + (lane >> 2);
    const int chunk = (lane & 3) ^ swz(row);
    q[i] = (gfloat*)(g.A + (long long)min(m0 + row, g.M - 1) * g.lda + chunk * 4);
  }
  if (!TB) {                                           // B (N,K): LDS row rho holds column 2 (rho & 31) + (rho >> 5); my rows 16 wave ..
    const int rho = 16 * wave + (lane >> 2);
    const int chunk = (lane & 3) ^ swz(rho);
    const int col = 2 * (rho & 31) + (rho >> 5);
    q[NGA] = (gfloat*)(g.B + (long long)min(n0 + col, g.N - 1) * g.ldb + chunk * 4);
  } else {                                             // B (K,N): my k-rows 4 wave + (l >> 4), columns 4 (l & 15) .. +3
    const int k = 4 * wave + (lane >> 4);
    q[NGA] = (gfloat*)(g.B + (long long)k * g.ldb + min(n0 + 4 * (lane & 15), g.N - 4));
  }
  auto copy = [&](int i, int sl) {                     // copy i of the slab going into slot sl
    lds_char* dst = (i < NGA) ? (lds_char*)(myA + sl * A_BYTES + i * 1024) : (lds_char*)(shB + sl * B_BYTES + wave * 1024);
    __builtin_amdgcn_global_load_lds(q[i], dst, 16, 0, 0);
    q[i] += (i >= NGA && TB) ? (long long)TK * g.ldb : TK;
  };
#pragma unroll
  for (int p = 0; p < NSL - 1; ++p)
    if (p < S) {
#pragma unroll
      for (int i = 0; i < NGS; ++i) copy(i, p);
    }
  const bool vec = ((g.ldc & 1) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 7) == 0);
  f32x2 cold[16];
  c_prefetch(g, m0, n0, lane, vec, cold);
  FragS<TB> fa[2], fb[2];
  int slot = 0;
  auto half = [&](int s, FragS<TB> (&cur)[2], FragS<TB> (&nxt)[2]) {
    const bool more = s + 1 < S, refill = s + NSL - 1 < S;
    const int nslot = (slot + 1 == NSL) ? 0 : slot + 1;
    const int rslot = (slot == 0) ? NSL - 1 : slot - 1;     // slot of slab s-1 takes slab s+NSL-1
#pragma unroll
    for (int n = 0; n < 16; ++n) {                     // MFMA n: k-step n >> 3, step e = (n >> 1) & 3, column tile n & 1
      const int ks = n >> 3, e = (n >> 1) & 3, j = n & 1;
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[ks].a[e], cur[ks].b(j, e), acc[j], 0, 0, 0);
      if (n < NGS && refill) copy(n, rslot);
      if (n == 6 && more) {                            // everybody's copies of slab s+1, then its fragment reads
        wait_copies3(min(NSL - 2, S - 2 - s));
        __builtin_amdgcn_s_barrier();
      }
      if (n == 7 && more) nxt[0].load(myA + nslot * A_BYTES, shB + nslot * B_BYTES, 0, lane);
      if (n == 9 && more) nxt[1].load(myA + nslot * A_BYTES, shB + nslot * B_BYTES, 1, lane);
      __builtin_amdgcn_sched_barrier(0);               // keep this interleave
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): no LDS read pending at the loop head
    __builtin_amdgcn_sched_barrier(0);
    slot = nslot;
  };
  if (S > 0) {
    wait_copies3(min(NSL - 2, S - 1));                 // my copies of slab 0
    __builtin_amdgcn_s_barrier();                      // ... and everybody else's
    fa[0].load(myA, shB, 0, lane);
    fa[1].load(myA, shB, 1, lane);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int s = 0; s < S; s += 2) {
    half(s, fa, fb);
    if (s + 1 < S) half(s + 1, fb, fa);
  }
  const int cm = lane & 31, h = lane >> 5;
  const int col = n0 + 2 * cm;
  const float b0 = (g.bias && col < g.N) ? g.bias[col] : 0.f;
  const float b1 = (g.bias && col + 1 < g.N) ? g.bias[col + 1] : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e)
    put2(g, m0 + (e & 3) + 8 * (e >> 2) + 4 * h, col, acc[0][e], acc[1][e], b0, b1, vec, cold[e][0], cold[e][1]);
}

template <bool TB, int NSL>
int launch_shb(const WaveArgs& g, int nwg, hipStream_t s) {
  static VqfDynLdsFlags attr = {};
  constexpr int SMEM_SHB = NSL * (4 * A_BYTES + B_BYTES);
  if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_wave_shb_kernel<TB, NSL>), SMEM_SHB, attr)) return e;
  VQF_LAUNCH(KID_GEMM_A0B0 + (TB ? 1 : 0), (gemm_f32_wave_shb_kernel<TB, NSL>), dim3(nwg), dim3(NT), SMEM_SHB, s, g);
  return vqf_last_error();
}

// ---- one LSTM step of the question encoder in ONE launch (mfb.py:69; host/functions.py::LstmBatchFn) --------------------------
//   gates_t (B, 4H) holds x_t W_ih^T + b on entry;  pre = gates_t + h_{t-1} W_hh^T;  i, f, o = sigmoid, g = tanh (PyTorch order
//   i, f, g, o);  c_t = f c_{t-1} + i g;  h_t = o tanh(c_t);  gates_t leaves ACTIVATED (kept for the backward).
// The shared-B kernel above with two changes.  (1) A wave's 64 output columns are 16 hidden units x 4 gates: the copy of the
// W_hh slab takes its source rows per LDS row (the DMA's source address is per lane, nothing else moves): LDS row rho = 32 j +
// cm holds W_hh row gate(j, cm) H + u0 + (cm & 15) with gate = 2 j + (cm >> 4) -- column tile 0 = (i | f), tile 1 = (g | o) of
// the same 16 units.  (2) The epilogue IS the cell: a lane activates its two pre-activations, lanes cm and cm ^ 16 hold the
// same unit and swap (f, o) for (i, g) with one cross-lane move each, lanes cm < 16 update c and h.  The old gate values, and
// c_{t-1}, are fetched at kernel entry like the accumulating product's C.  Per element the same arithmetic in the same order as
// vqf_gemm_f32(ACCUM) + vqf_lstm_cell_fwd: bit-identical, 14 launches and a 16 MB round trip per step fewer.
__device__ __forceinline__ float sigm_f(float x) { return 1.0f / (1.0f + expf(-x)); }     // lstm_cell.hip's sigm

__global__ void __launch_bounds__(NT, 1) gemm_f32_lstm_fwd_kernel(const WaveArgs g) {
  constexpr int NSL = 6;
  constexpr int SHB_A = NSL * A_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int t = 4 * wg + wave;                         // tiles_m % 4 == 0 (host): the 4 waves share the unit block
  const int tm = t % g.tiles_m, tn = t / g.tiles_m;
  const int m0 = tm * WTM, u0 = tn * 16, H = g.H;
  const int S = g.kpart / TK;
  char* myA = smem + wave * SHB_A;
  char* shB = smem + 4 * SHB_A;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  gfloat* q[NGS];
#pragma unroll
  for (int i = 0; i < NGA; ++i) {
    const int row = 16 * i + (lane >> 2);
    const int chunk = (lane & 3) ^ swz(row);
    q[i] = (gfloat*)(g.A + (long long)min(m0 + row, g.M - 1) * g.lda + chunk * 4);
  }
  {
    const int rho = 16 * wave + (lane >> 2);           // my quarter of the workgroup's B slab: LDS rows 16 wave .. +15
    const int chunk = (lane & 3) ^ swz(rho);
    const int gate = 2 * (rho >> 5) + ((rho & 31) >> 4);
    q[NGA] = (gfloat*)(g.B + (long long)(gate * H + u0 + (rho & 15)) * g.ldb + chunk * 4);
  }
  auto copy = [&](int i, int sl) {
    lds_char* dst = (i < NGA) ? (lds_char*)(myA + sl * A_BYTES + i * 1024) : (lds_char*)(shB + sl * B_BYTES + wave * 1024);
    __builtin_amdgcn_global_load_lds(q[i], dst, 16, 0, 0);
    q[i] += TK;
  };
#pragma unroll
  for (int p = 0; p < NSL - 1; ++p)
    if (p < S) {
#pragma unroll
      for (int i = 0; i < NGS; ++i) copy(i, p);
    }
  // the lane's old gate values (x W_ih^T + b) and, for the lanes that update the state, c_{t-1}: in flight behind the copies
  const int cm = lane & 31, hh = lane >> 5;
  const int col0 = (cm >> 4) * H + u0 + (cm & 15);     // tile 0: gate cm >> 4 (i | f); tile 1: + 2 H (g | o)
  float old0[16], old1[16], cpv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
    const float* pc = g.C + (long long)row * g.ldc + col0;
    old0[e] = pc[0];
    old1[e] = pc[2 * H];
    cpv[e] = (g.c_prev && cm < 16) ? g.c_prev[(long long)row * H + u0 + cm] : 0.f;
  }
  FragS<false> fa[2], fb[2];
  int slot = 0;
  auto half = [&](int s, FragS<false> (&cur)[2], FragS<false> (&nxt)[2]) {
    const bool more = s + 1 < S, refill = s + NSL - 1 < S;
    const int nslot = (slot + 1 == NSL) ? 0 : slot + 1;
    const int rslot = (slot == 0) ? NSL - 1 : slot - 1;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      const int ks = n >> 3, e = (n >> 1) & 3, j = n & 1;
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[ks].a[e], cur[ks].b(j, e), acc[j], 0, 0, 0);
      if (n < NGS && refill) copy(n, rslot);
      if (n == 6 && more) {
        wait_copies3(min(NSL - 2, S - 2 - s));
        __builtin_amdgcn_s_barrier();
      }
      if (n == 7 && more) nxt[0].load(myA + nslot * A_BYTES, shB + nslot * B_BYTES, 0, lane);
      if (n == 9 && more) nxt[1].load(myA + nslot * A_BYTES, shB + nslot * B_BYTES, 1, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    slot = nslot;
  };
  if (S > 0) {
    wait_copies3(min(NSL - 2, S - 1));
    __builtin_amdgcn_s_barrier();
    fa[0].load(myA, shB, 0, lane);
    fa[1].load(myA, shB, 1, lane);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int s = 0; s < S; s += 2) {
    half(s, fa, fb);
    if (s + 1 < S) half(s + 1, fb, fa);
  }
  // ---- the cell.  Lane (cm, hh), register e: row (e & 3) + 8 (e >> 2) + 4 hh, unit u0 + (cm & 15);
  //      tile 0 = gate i (cm < 16) | f (cm >= 16), tile 1 = gate g | o
  const bool lo = cm < 16;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
    const float a0 = sigm_f(acc[0][e] + old0[e]);                         // i | f
    const float p1 = acc[1][e] + old1[e];
    const float a1 = lo ? tanhf(p1) : sigm_f(p1);                         // g | o
    float* pc = g.C + (long long)row * g.ldc + col0;
    pc[0] = a0;
    pc[2 * H] = a1;
    const float f_act = __shfl_xor(a0, 16, 64), o_act = __shfl_xor(a1, 16, 64);
    if (lo) {
      const float c = f_act * cpv[e] + a0 * a1;
      g.c_out[(long long)row * H + u0 + cm] = c;
      g.h_out[(long long)row * H + u0 + cm] = o_act * tanhf(c);
    }
  }
}

int launch_lstm_fwd(const WaveArgs& g, int nwg, hipStream_t s) {
  static VqfDynLdsFlags attr = {};
  constexpr int SMEM_SHB = 6 * (4 * A_BYTES + B_BYTES);
  if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_lstm_fwd_kernel), SMEM_SHB, attr)) return e;
  VQF_LAUNCH(KID_LSTM_CELL_FWD, gemm_f32_lstm_fwd_kernel, dim3(nwg), dim3(NT), SMEM_SHB, s, g);
  return vqf_last_error();
}

template <bool TB, int WK>
int launch(const WaveArgs& g, int nwg, hipStream_t s) {
  static VqfDynLdsFlags attr = {};
  if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_f32_wave_kernel<TB, WK>), SMEM_WAVE, attr)) return e;
  VQF_LAUNCH(KID_GEMM_A0B0 + (TB ? 1 : 0), (gemm_f32_wave_kernel<TB, WK>), dim3(nwg), dim3(NT), SMEM_WAVE, s, g);
  return vqf_last_error();
}

}  // namespace

// 0 = this kernel does not apply (caller continues with gemm_f32.hip), 1 = launched (rc holds the status)
int vqf_gemm_f32_wave_try(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                          int ldc, const float* bias, int flags, hipStream_t s, int* rc) {
  if (vqf_opt(VQF_OPT_GEMM_F32_WAVE, 1) == 0) return 0;   // A/B switch: 0 disables this kernel
  if (ta || M > 1024 || (K % TK) || K < 256 || !aligned16(A) || !aligned16(B) || (lda % 4) || (ldb % 4)) return 0;
  if (tb && (N % 4)) return 0;
  // only where the 128x128 kernel cannot fill the chip without split-K slabs
  if ((long long)((M + 127) / 128) * ((N + 127) / 128) >= 256) return 0;
  WaveArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.flags = flags;
  g.tiles_m = (M + WTM - 1) / WTM;
  g.tiles_n = (N + WTN - 1) / WTN;
  const int tiles = g.tiles_m * g.tiles_n;
  const int nwg1 = (tiles + 3) / 4, nwg4 = tiles;
  const bool ok4 = (K % (4 * TK) == 0) && K / 4 >= 256;
  // one wave per tile and whole K when that fills the chip, else four K ranges per tile; neither -> not this kernel
  int wk = 0;
  // (one round only: with several rounds per CU the 128x128 kernel's split-K form measured faster, 512 x 5000 x 2048)
  if (nwg1 <= 256 && nwg1 >= 192) wk = 1;
  else if (ok4 && nwg4 <= 256 && nwg4 >= 192) wk = 4;
  if (!wk) return 0;
  g.kpart = (wk == 1) ? K : K / 4;
  vqf_prof_dims(M, N, K);
  vqf_stat_bump(VQF_STAT_GEMM_F32_WAVE);
  // option gemm_f32_wave: 1 = every wave streams its own B slab (round 2), 2 / default = B staged once per workgroup
  const bool shb = wk == 1 && (g.tiles_m % 4 == 0) && tiles % 4 == 0 && vqf_opt(VQF_OPT_GEMM_F32_WAVE, 2) != 1;
  if (shb && vqf_opt(VQF_OPT_GEMM_F32_WAVE, 2) == 3)   // A/B: 12 slots, 11 slabs in flight
    *rc = tb ? launch_shb<true, 12>(g, nwg1, s) : launch_shb<false, 12>(g, nwg1, s);
  else if (shb)     *rc = tb ? launch_shb<true, 6>(g, nwg1, s) : launch_shb<false, 6>(g, nwg1, s);
  else if (wk == 1) *rc = tb ? launch<true, 1>(g, nwg1, s) : launch<false, 1>(g, nwg1, s);
  else         *rc = tb ? launch<true, 4>(g, nwg4, s) : launch<false, 4>(g, nwg4, s);
  return 1;
}

// One fused LSTM step (see gemm_f32_lstm_fwd_kernel).  Supported: B % 128 == 0, H % 16 == 0, H >= 256, 16-byte aligned rows;
// else VQF_E_UNSUPPORTED and the caller runs vqf_gemm_f32(VQF_GEMM_ACCUM) + vqf_lstm_cell_fwd (the same bits).
extern "C" int vqf_lstm_step_supported(int B, int H) {
  return B > 0 && H >= 256 && (B % 128) == 0 && (H % 16) == 0 && vqf_opt(VQF_OPT_GEMM_F32_WAVE, 2) != 0;
}
extern "C" int vqf_lstm_step_fwd(const float* h_prev, const float* w_hh, float* gates, const float* c_prev, int B, int H,
                                 float* c_out, float* h_out, void* stream) {
  if (!h_prev || !w_hh || !gates || !c_out || !h_out || B <= 0 || H <= 0) return VQF_E_BADARG;
  if (!vqf_lstm_step_supported(B, H)) return VQF_E_UNSUPPORTED;
  if (!aligned16(h_prev) || !aligned16(w_hh) || !aligned16(gates) || !aligned16(c_out) || !aligned16(h_out) ||
      (c_prev && !aligned16(c_prev)))
    return VQF_E_ALIGN;
  {
    // round 5: one workgroup per CU on 16x16x4 tiles whose four column tiles are a unit's four gates (gemm_f32_n80.hip) where the
    // shape gives one round (B = 512, H = 1024: 256 tiles); option gemm_f32_n80 = 0: the per-wave form below
    int rc = VQF_OK;
    if (vqf_lstm_step16_try(h_prev, w_hh, gates, c_prev, B, H, c_out, h_out, (hipStream_t)stream, &rc)) return rc;
  }
  WaveArgs g = {};
  g.A = h_prev; g.B = w_hh; g.C = gates; g.bias = nullptr;
  g.M = B; g.N = 4 * H; g.K = H; g.lda = H; g.ldb = H; g.ldc = 4 * H; g.flags = VQF_GEMM_ACCUM;
  g.tiles_m = B / WTM;
  g.tiles_n = H / 16;
  g.kpart = H;
  g.c_prev = c_prev; g.c_out = c_out; g.h_out = h_out; g.H = H;
  vqf_prof_dims(B, 4 * H, H);
  vqf_stat_bump(VQF_STAT_GEMM_F32_WAVE);
  return launch_lstm_fwd(g, g.tiles_m * g.tiles_n / 4, (hipStream_t)stream);
}
