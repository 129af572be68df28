// bf16 x bf16 -> fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16), gfx950 only.
//
//   C[m,n] (fp32) = sum_k Aop[m,k] * Bop[n,k] (+ bias[n]) (relu)      A, B stored as bf16
//
// BASELINE config 3 (MHBCoAtt, "bf16"): bf16 storage for the image tensor / activations /
// weights of the large projections, fp32 accumulation and fp32 everywhere else.  Same
// decomposition as gemm_f32.hip (128x128 block tile, 4 wave64, 64x64 per wave, 4 workgroups per
// CU, register-prefetched staging, two LDS buffers, one barrier per K-tile, XCD-aware grouped
// tile order, deterministic split-K), with BK = 32:
//   * K-contiguous operands: LDS image [row][32+8] bf16 (80-byte rows), one ds_read_b128 per
//     MFMA operand: lane (r, h) holds k = 16s + 8h .. +7, exactly the 32x32x16 operand map.
//   * K-major operands (both operands of the weight gradient, K = N*196; the weight in dgrad):
//     LDS image [k][128+32] bf16 and the hardware transposing read ds_read_b64_tr_b16: a 16-lane
//     group reads a 4(k) x 16(row) block and each lane receives the 4 k-values of ITS row; two
//     such reads make one operand.  The 320-byte row stride puts the 4 k-rows of a block on
//     disjoint bank quarters.
// Requirements (else VQF_E_UNSUPPORTED and the caller uses the fp32 kernel): 16-byte aligned
// bases, leading dimensions % 8 == 0; K-contiguous operand: K % 8 == 0; K-major operand: its row
// extent % 8 == 0 and >= 8.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;                     // storage type at the ABI
typedef const bf16_t __attribute__((address_space(1))) gbf16;
typedef const f32x4 __attribute__((address_space(1))) gvec16;   // any 16-byte chunk

constexpr int BM = 128, BN = 128, BK = 32, NTHREADS = 256;
constexpr int LD_RK = BK + 8;                 // elements; 80-byte rows
constexpr int LD_KR = BM + 32;                // elements; 320-byte k-rows
constexpr int OP_ELEMS = (BM * LD_RK > BK * LD_KR) ? BM * LD_RK : BK * LD_KR;   // 5120
constexpr int STAGE_ELEMS = 2 * OP_ELEMS;
constexpr int SMEM_BYTES = 2 * STAGE_ELEMS * 2;      // 40,960 B
constexpr int NLD = BM * BK / 8 / NTHREADS;          // 16-byte loads per thread per operand (2)

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  float* C;
  const float* bias;
  float* slab;
  int M, N, K, lda, ldb, ldc;
  int flags;
  int kchunk;
  int tiles_m, tiles_n;
  const float* rowscale;   // epilogue: C = rowscale[row / rps] * acc + bias (vqf_gemm_bf16_rowscale), or nullptr
  int rps;
  int* cnt;                // split-K combined in the launch: arrival counters, one per tile (common.h), or nullptr
};

// thread -> (row, k) of its i-th 16-byte chunk
template <bool T>
__device__ __forceinline__ void chunk_rc(int f, int& r, int& k) {
  if (!T) { r = f >> 2; k = (f & 3) << 3; }        // 4 chunks of 8 k per row
  else    { k = f >> 4; r = (f & 15) << 3; }       // 16 chunks of 8 rows per k-row
}

template <bool T>
__device__ __forceinline__ void init_ptrs(const bf16_t* p0, int ld, int r0, int R, int k0, int tid,
                                          gbf16* (&q)[NLD]) {
  gbf16* p = (gbf16*)p0;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int r, k;
    chunk_rc<T>(tid + NTHREADS * i, r, k);
    if (!T) q[i] = p + (long long)min(r0 + r, R - 1) * ld + (k0 + k);
    else    q[i] = p + (long long)(k0 + k) * ld + min(r0 + r, R - 8);
  }
}

// loads tile t (k range [k0, k0+BK)); chunks beyond kend are zero (K tail of the last tile)
template <bool T>
__device__ __forceinline__ void load_tile(gbf16* (&q)[NLD], int ld, int k0, int kend, int tid,
                                          f32x4 (&v)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int r, k;
    chunk_rc<T>(tid + NTHREADS * i, r, k);
    const bool ok = (k0 + k) < kend;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (ok) x = *(gvec16*)q[i];
    v[i] = x;
    q[i] += T ? (long long)BK * ld : BK;
  }
}

template <bool T>
__device__ __forceinline__ void store_tile(bf16_t* s, int tid, const f32x4 (&v)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int r, k;
    chunk_rc<T>(tid + NTHREADS * i, r, k);
    if (!T) *reinterpret_cast<f32x4*>(s + r * LD_RK + k) = v[i];
    else    *reinterpret_cast<f32x4*>(s + k * LD_KR + r) = v[i];
  }
}

// operand fragment of one 32-row sub-tile for MFMA k-step ks (16 k): lane (r=l&31, h=l>>5)
// holds k = 16*ks + 8*h + j, j = 0..7
template <bool T>
__device__ __forceinline__ bf16x8 read_frag(const bf16_t* s, int row0, int ks, int lane) {
  if (!T) {
    const int r = lane & 31, h = lane >> 5;
    return *reinterpret_cast<const bf16x8*>(s + (row0 + r) * LD_RK + 16 * ks + 8 * h);
  }
  // K-major image: 16-lane group g covers rows row0 + 16*(g&1) .. +15 and k-block 16*ks + 8*(g>>1);
  // lane 4q+p of the group points at k-row q, rows 4p..4p+3 of the block and RECEIVES its own
  // row's 4 k-values (ds_read_b64_tr_b16).
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int kb = 16 * ks + 8 * (g >> 1);
  const bf16_t* a0 = s + (kb + q) * LD_KR + row0 + 16 * (g & 1) + 4 * p;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * LD_KR));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool TA, bool TB, bool EDGE>
__device__ __forceinline__ void mainloop(const GemmArgs& g, bf16_t* smem, const bf16_t* gA,
                                         const bf16_t* gB, int m0, int n0, int kbeg, int kend,
                                         int tid, int lane, int wr, int wc, f32x16 (&acc)[2][2]) {
  const int ntiles = (kend - kbeg + BK - 1) / BK;
  bool v10 = true, v01 = true;
  if (EDGE) {
    v10 = (m0 + wr * 64 + 32) < g.M;
    v01 = (n0 + wc * 64 + 32) < g.N;
  }
  gbf16* pa[NLD];
  gbf16* pb[NLD];
  f32x4 ra[NLD], rb[NLD];
  init_ptrs<TA>(gA, g.lda, m0, g.M, kbeg, tid, pa);
  init_ptrs<TB>(gB, g.ldb, n0, g.N, kbeg, tid, pb);
  load_tile<TA>(pa, g.lda, kbeg, kend, tid, ra);
  load_tile<TB>(pb, g.ldb, kbeg, kend, tid, rb);
  store_tile<TA>(smem, tid, ra);
  store_tile<TB>(smem + OP_ELEMS, tid, rb);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const bf16_t* sA = smem + (t & 1) * STAGE_ELEMS;
    const bf16_t* sB = sA + OP_ELEMS;
    const bool more = (t + 1) < ntiles;
    if (more) {
      const int k0 = kbeg + (t + 1) * BK;
      load_tile<TA>(pa, g.lda, k0, kend, tid, ra);
      load_tile<TB>(pb, g.ldb, k0, kend, tid, rb);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (!EDGE) {
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = read_frag<TA>(sA, wr * 64 + i * 32, ks, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = read_frag<TB>(sB, wc * 64 + j * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      } else {
        // the transposing read needs EXEC all ones: fragments are read unconditionally (the LDS
        // image is always fully written), only the MFMAs of outside sub-tiles are skipped
        const bf16x8 a0 = read_frag<TA>(sA, wr * 64, ks, lane);
        const bf16x8 a1 = read_frag<TA>(sA, wr * 64 + 32, ks, lane);
        const bf16x8 b0 = read_frag<TB>(sB, wc * 64, ks, lane);
        const bf16x8 b1 = read_frag<TB>(sB, wc * 64 + 32, ks, lane);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        if (v10) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        if (v01) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      }
    }
    if (more) {
      bf16_t* d = smem + ((t + 1) & 1) * STAGE_ELEMS;
      store_tile<TA>(d, tid, ra);
      store_tile<TB>(d + OP_ELEMS, tid, rb);
    }
    __syncthreads();
  }
}

template <bool TA, bool TB>
__global__ void __launch_bounds__(NTHREADS, 4) gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_bf[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

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

  const bool all_valid = (m0 + wr * 64 + 32) < g.M && (n0 + wc * 64 + 32) < g.N;
  if (all_valid)
    mainloop<TA, TB, false>(g, smem_bf, g.A, g.B, m0, n0, kbeg, kend, tid, lane, wr, wc, acc);
  else
    mainloop<TA, TB, true>(g, smem_bf, g.A, g.B, m0, n0, kbeg, kend, tid, lane, wr, wc, acc);

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool to_slab = g.slab != nullptr;
  float* out = to_slab ? g.slab + (long long)blockIdx.y * g.M * g.N : g.C;
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
          float v = (g.rowscale ? acc[i][j][r] * g.rowscale[row / g.rps] : acc[i][j][r]) + bv;
          if (!to_slab) {
            if (g.flags & VQF_GEMM_ACCUM) v += *pc;
            if (g.flags & VQF_GEMM_RELU) v = fmaxf(v, 0.f);
          }
          *pc = v;
        }
      }
    }
  }
  if (to_slab && g.cnt) {      // split-K combined in this launch by the tile's last arriver (common.h)
    const VqfSplitkTile st = {g.cnt, g.slab, g.C, g.bias, g.M, g.N, g.ldc, g.flags};
    vqf_splitk_combine<BM, BN, NTHREADS>(st, bid, (int)gridDim.y, m0, n0, tid, reinterpret_cast<float*>(smem_bf));
  }
}

template <bool TA, bool TB>
int launch(const GemmArgs& g, dim3 grid, hipStream_t s) {
  static VqfDynLdsFlags attr = {};
  if (int e = vqf_set_dyn_lds(reinterpret_cast<const void*>(&gemm_bf16_kernel<TA, TB>), SMEM_BYTES, attr)) return e;
  VQF_LAUNCH(KID_GEMM_BF16, (gemm_bf16_kernel<TA, TB>), grid, dim3(NTHREADS), SMEM_BYTES, s, g);
  return vqf_last_error();
}

// y[r, c] = bf16(x[r, c]) for c < C, 0 for C <= c < ldy   (round-to-nearest-even, NaN kept).  One launch whatever R (round 4:
// a workgroup per row with 8 columns per thread -- half of its threads idle at 1024 columns -- and one launch per 65535 rows)
__global__ void cast_bf16_kernel(const float* __restrict__ x, int R, int C, int ldx,
                                 bf16_t* __restrict__ y, int ldy) {
  const unsigned W8 = (unsigned)ldy >> 3;
  const unsigned long long total = (unsigned long long)R * W8, stride = (unsigned long long)gridDim.x * blockDim.x;
  const bool vec = ((ldx & 3) == 0) && aligned16_dev(x);
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const unsigned r = (unsigned)(i / W8);
    const int c8 = (int)(i - (unsigned long long)r * W8) * 8;
    const float* xr = x + (long long)r * ldx;
    __bf16 o[8];
    if (c8 + 7 < C && vec) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + c8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(xr + c8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = (__bf16)a[j]; o[4 + j] = (__bf16)b[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (c8 + j < C) ? (__bf16)xr[c8 + j] : (__bf16)0.0f;
    }
    *reinterpret_cast<f32x4*>(y + (long long)r * ldy + c8) = *reinterpret_cast<const f32x4*>(o);
  }
}

}  // namespace

extern "C" {

size_t vqf_gemm_bf16_ws_bytes(int ta, int tb, int M, int N, int K) {
  return vqf_gemm_bf16_big_ws_bytes(ta, tb, M, N, K);
}

static int gemm_bf16_impl(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C,
                          int ldc, const float* bias, int flags, const float* rowscale, int rps, void* ws, size_t ws_bytes,
                          void* stream);

int vqf_gemm_bf16(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B,
                  int ldb, float* C, int ldc, const float* bias, int flags, void* ws,
                  size_t ws_bytes, void* stream) {
  return gemm_bf16_impl(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, nullptr, 1, ws, ws_bytes, stream);
}

int vqf_gemm_bf16_rowscale(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C,
                           int ldc, const float* bias, int flags, const float* rowscale, int rows_per_scale, void* stream) {
  if (!rowscale || rows_per_scale <= 0) return VQF_E_BADARG;
  if (flags & (VQF_GEMM_ACCUM | VQF_GEMM_OUT_BF16)) return VQF_E_UNSUPPORTED;
  return gemm_bf16_impl(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, rowscale, rows_per_scale, nullptr, 0, stream);
}

static int gemm_bf16_impl(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C,
                          int ldc, const float* bias, int flags, const float* rowscale, int rps, void* ws, size_t ws_bytes,
                          void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc < N)
    return VQF_E_BADARG;
  if (!aligned16(A) || !aligned16(B) || (lda % 8) || (ldb % 8)) return VQF_E_UNSUPPORTED;
  if (!ta && (K % 8)) return VQF_E_UNSUPPORTED;
  if (!tb && (K % 8)) return VQF_E_UNSUPPORTED;
  if (ta && ((M % 8) || M < 8)) return VQF_E_UNSUPPORTED;
  if (tb && ((N % 8) || N < 8)) return VQF_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  {
    int rc = VQF_OK;      // the two big projections take the 256x256-tile kernel
    if (vqf_gemm_bf16_big_try(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, rowscale, rps, ws, ws_bytes, s, &rc)) return rc;
    if (flags & VQF_GEMM_OUT_BF16) return VQF_E_UNSUPPORTED;   // bf16 output exists in the large-tile kernel only
  }
  GemmArgs g;
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.bias = bias; g.slab = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.flags = flags;
  g.rowscale = rowscale; g.rps = rps; g.cnt = nullptr;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + BN - 1) / BN;
  const long long tiles = (long long)g.tiles_m * g.tiles_n;
  const int ktiles = (K + BK - 1) / BK;
  int splits = 1;
  const int slots = 512;
  if (ws && !rowscale && tiles < 1024 && ktiles >= 32) {
    double best = 1e30;
    for (int sp = 1; sp <= 16; ++sp) {
      if (ktiles / sp < 16) break;
      if ((size_t)sp * M * N * sizeof(float) > ws_bytes) break;
      const double blocks = (double)tiles * sp;
      const double rounds = (double)((long long)((blocks + slots - 1) / slots));
      const double cost = rounds / sp * (1.0 + 0.01 * sp);
      if (cost < best - 1e-12) { best = cost; splits = sp; }
    }
  }
  const int kt_per = (ktiles + splits - 1) / splits;
  g.kchunk = kt_per * BK;
  splits = (K + g.kchunk - 1) / g.kchunk;
  if (splits > 1) {
    g.slab = (float*)ws;
    if (vqf_opt(VQF_OPT_GEMM_SPLITK_FUSED, 0) == 1) g.cnt = vqf_splitk_counters((int)tiles);
  }
  if (g_vqf_prof_on) vqf_prof_dims(M, N, K);
  vqf_stat_bump(VQF_STAT_GEMM_BF16_TILE128);
  dim3 grid((unsigned)tiles, (unsigned)splits);
  int rc;
  if (!ta && !tb) rc = launch<false, false>(g, grid, s);
  else if (!ta && tb) rc = launch<false, true>(g, grid, s);
  else if (ta && !tb) rc = launch<true, false>(g, grid, s);
  else rc = launch<true, true>(g, grid, s);
  if (rc != VQF_OK) {
    vqf_splitk_counters_clear(g.cnt, (int)tiles, s);
    return rc;
  }
  if (splits > 1 && !g.cnt) rc = vqf_splitk_reduce((const float*)ws, splits, M, N, C, ldc, bias, flags, s);
  return rc;
}

int vqf_cast_f32_bf16(const float* x, int R, int C, int ldx, void* y, int ldy, void* stream) {
  if (!x || !y || R <= 0 || C <= 0 || ldx < C || ldy < C) return VQF_E_BADARG;
  if ((ldy % 8) || !aligned16(y)) return VQF_E_ALIGN;
  const unsigned long long total = (unsigned long long)R * (unsigned)(ldy / 8);
  unsigned long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  VQF_LAUNCH(KID_CAST_BF16, cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, R, C, ldx, (bf16_t*)y, ldy);
  if (int rc = vqf_last_error()) return rc;
  return VQF_OK;
}

}  // extern "C"
