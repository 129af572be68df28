// HieCoAtten's co-attention ladder (hieCoAtten.py:32-49), the stages whose products have a tiny inner or outer extent
// (T = words per question, 14): as MFMA GEMMs they are (196 x 512) x 14 outer-product updates and 14-row reductions that
// stream a (N, L, E) tensor at a quarter of the memory rate.  Here each of them is ONE streaming pass over the rows of the
// (N*L, E) tensors -- 16-byte coalesced accesses, the per-sample (T, E) and (T, L) operands in LDS, the T-row accumulations in
// registers -- with the neighbouring element-wise stage fused in:
//
//   mode FWD   out[l,:] = dropout(tanh(a[l,:] + sum_t U[t,l] V[t,:]))         Hv = drop(tanh(img_ + C^T que_))      :38-39
//              part[t,:] += U[t,l] a[l,:]                                      ti = C img_  (the question side's input) :45
//   mode HEAD  out[l,:] = dl[l] w[:] sc (1 - (a[l,:]/sc)^2)                    d(img_ + tq): backward of :38-40 from the
//              part[t,:] += U[t,l] out[l,:];  wpart[:] += dl[l] a[l,:]         logit gradient (never materialising dHv);
//                                                                              C dtq -> dque_;  dl^T Hv -> d fc_Whv.weight
//   mode ADD   out[l,:] = a[l,:] + sum_t U[t,l] V[t,:]                         dimg_ += C^T dti                      (of :45)
//   mode LEFT  out[l,:] = sum_t U[t,l] V[t,:];  part[t,:] += U[t,l] z[l,:]     dCv = daff^T Cq;  dCq = daff Cv       (of :32)
//              ADD / LEFT: colpart[:] += out[l,:] -- the two halves of [dCv | dimg_] are final here, their column sums are the
//              bias gradients of fc_Wbv / fc_Wv (one partial row per workgroup instead of a 205 MB column-sum pass)
//
// a, z, out: rows m = n*L + l of 2-D tensors with row strides (column blocks of the concatenated-weight products, functions.
// HieCoreFn); U (N, T, L) contiguous; V rows n*T + t with stride ldv.  part: (S, N, T, E) partial sums over the S row chunks of
// a sample, summed afterwards in chunk order (deterministic, no atomics).  T <= 16, E % 4 == 0, (E / 4) divides 256.
#include "common.h"

namespace {

// Threads per workgroup: 1024.  Round 4 ran 256-thread workgroups on quarter-sample chunks: every workgroup re-loaded its
// sample's (T, E) operand (28 KB for 98 KB of streamed rows: the 18-31 % over-fetch of profiles/r04_pmc_hie.txt), wrote a partial
// slab the slab-sum launch read back, and a CU held 32 KB of loads in flight.  One 1024-thread workgroup per SAMPLE (N >= the CU
// count; smaller batches cut the sample into chunks as before) loads the operand once, needs no partial slabs (S == 1: the T-row
// sums go straight to their destination, optionally on top of another tensor) and keeps the same 16 waves per CU.
constexpr int HT = 1024;
constexpr int TMAX_ALL = 16;

enum { MODE_FWD = 0, MODE_HEAD = 1, MODE_ADD = 2, MODE_LEFT = 3 };

struct HieArgs {
  const float* a; int lda;
  const float* z; int ldz;
  float* out; int ldo;
  const float* U;
  const float* V; int ldv;
  float* part; int ldp;                              // partial sums: (S, N*T) rows of pitch ldp (S > 1: ldp == E, contiguous slabs)
  const float* padd; int ldpa;                       // S == 1 only: the sums are written ON TOP of these rows (or nullptr)
  const uint8_t* keep; uint64_t seed; uint32_t thr; float inv_keep;
  const float* dl; const float* w; float* wpart; int ldw;   // HEAD: logit gradient (N*L), head weight (E), partial rows (S*N) of pitch ldw >= E + 4
  float* colpart; int ldcp;                          // ADD / LEFT: per-workgroup column sums of `out` (rows s*N + n, pitch ldcp) or nullptr
  int N, L, E, T, S, Lc;
};

__device__ __forceinline__ void keep4v(const uint8_t* __restrict__ keep, uint64_t seed, uint32_t thr, float inv_keep,
                                       long long i4, f32x4& sc) {
#ifdef VQF_HIE_DIAG_NOPHILOX                           // (diagnostic build: what the Philox draw costs the passes; wrong masks)
  if (!keep && thr != 0u) { sc = f32x4{inv_keep, inv_keep, inv_keep, inv_keep}; return; }
#endif
  if (keep) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(keep + 4 * i4);
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[j] = ((w >> (8 * j)) & 0xFFu) ? inv_keep : 0.f;
  } else if (thr != 0u) {
    const uint4 r = philox4x32_10((uint64_t)i4, seed);       // the draw of the flat element-wise kernels (elementwise.hip)
    sc[0] = r.x >= thr ? inv_keep : 0.f;
    sc[1] = r.y >= thr ? inv_keep : 0.f;
    sc[2] = r.z >= thr ? inv_keep : 0.f;
    sc[3] = r.w >= thr ? inv_keep : 0.f;
  } else {
    sc = f32x4{1.f, 1.f, 1.f, 1.f};
  }
}

// TMAX: compile-time bound of T (8, 14 or 16: the T-row accumulators are 4 TMAX registers; at 16 the HEAD mode spilled under the
// 128-register budget of a 1024-thread workgroup)
template <int MODE, int TMAX>
__global__ void __launch_bounds__(HT) hie_stream_kernel(const HieArgs g) {
  // The rank-T sums of this kernel may contract to fused multiply-adds (the library is built with -ffp-contract=off so that the
  // GEMM kernels and their references add in ONE stated order; these passes have no bit-level counterpart -- the batched-GEMM
  // form they replace rounds differently anyway -- and are VALU-sensitive: 112 multiply-add pairs per 16 bytes streamed.
  // Measured with the whole file contracted: forward 77 -> 68 us, head backward 67 -> 62, rank_left 60 -> 52).
#ifndef VQF_HIE_NO_CONTRACT                           // (A/B build switch)
#pragma clang fp contract(fast)
#endif
  const int HT = blockDim.x;                                // (shadows the constant: small problems launch narrower workgroups)
  constexpr bool RANK = MODE != MODE_HEAD;                  // out has the rank-T term sum_t U[t,l] V[t,:]
  constexpr bool ACC = MODE != MODE_ADD;                    // part[t,:] += U[t,l] (a | out | z)[l,:]
  extern __shared__ float smem[];
  const int E = g.E, T = g.T, L = g.L;
  const int CT = E >> 2, RS = HT / CT;
  const int n = blockIdx.y, s = blockIdx.x;
  const int l0 = s * g.Lc, l1 = min(L, l0 + g.Lc), rows = l1 - l0;
  const int tid = threadIdx.x, c4 = tid % CT, rs = tid / CT;
  float* Vs = smem;                                          // [T][E]        (RANK)
  float* Us = Vs + (RANK ? T * E : 0);                       // [T][Lc]
  float* red = smem;                                         // [TG][RS][E]: re-uses the operand images behind the main loop
  if (rows <= 0) return;
  if (RANK)
    for (int i = tid; i < T * CT; i += HT) {
      const int t = i / CT, c = i - t * CT;
      *reinterpret_cast<f32x4*>(Vs + t * E + 4 * c) = *reinterpret_cast<const f32x4*>(g.V + (long long)(n * T + t) * g.ldv + 4 * c);
    }
  for (int i = tid; i < T * rows; i += HT) {
    const int t = i / rows, r = i - t * rows;
    Us[t * g.Lc + r] = g.U[((long long)n * T + t) * L + l0 + r];
  }
  __syncthreads();

  f32x4 tacc[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) tacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 wacc = {0.f, 0.f, 0.f, 0.f};
  float dlsum = 0.f;
  f32x4 wv = {0.f, 0.f, 0.f, 0.f};
  constexpr bool COLS = MODE == MODE_ADD || MODE == MODE_LEFT;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  if (MODE == MODE_HEAD) wv = *reinterpret_cast<const f32x4*>(g.w + 4 * c4);

  // NR rows per trip: every row's load is issued before any is consumed.  The pass is latency-bound, not VALU- or LDS-bound (28
  // FMAs and one 16-byte LDS read per element and t are a tenth of the pass): with two rows per trip a CU's 16 waves kept 32 KB
  // in flight, 2.3-2.7 TB/s over the (N*L, E) tensors (profiles/r04_pmc_hie.txt); four rows double that.
  constexpr int NR = 2;        // (four rows per trip: 133-154 VGPRs, fewer waves, measured slower: gpurun_out/r05/b_c4_a.json)
  // The next trip's rows are requested BEFORE this trip's arithmetic (round 5): a wave's trips were a chain of load latency +
  // compute (13 of them per sample at 2-3 us each); with the prefetch the latency of trip i+1 hides behind the compute of trip i.
  long long mn[NR];
  f32x4 xn[NR];
  float dn[NR];
  auto fetch = [&](int r) {
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const bool h = r + q * RS < rows;
      mn[q] = (long long)n * L + l0 + (h ? r + q * RS : min(r, rows - 1));
      xn[q] = MODE != MODE_LEFT ? vqf_ld_stream(reinterpret_cast<const f32x4*>(g.a + mn[q] * g.lda + 4 * c4))
                                : vqf_ld_stream(reinterpret_cast<const f32x4*>(g.z + mn[q] * g.ldz + 4 * c4));
      if (MODE == MODE_HEAD) dn[q] = g.dl[mn[q]];
    }
  };
  if (rs < rows) fetch(rs);
  for (int r = rs; r < rows; r += NR * RS) {
    long long m[NR];
    bool has[NR];
    f32x4 x[NR], o[NR];
    float dd[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      has[q] = r + q * RS < rows;
      m[q] = mn[q];
      x[q] = xn[q];
      dd[q] = dn[q];
    }
    if (r + NR * RS < rows) fetch(r + NR * RS);
    if (MODE == MODE_HEAD) {
      const float keep_q = 1.0f / g.inv_keep;
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const float d = dd[q];
        f32x4 sc;
        keep4v(g.keep, g.seed, g.thr, g.inv_keep, m[q] * CT + c4, sc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = sc[j] > 0.f ? x[q][j] * keep_q : 0.f;       // tanh value back from the stored tanh * keep / (1 - p)
          o[q][j] = d * wv[j] * sc[j] * (1.0f - t * t);
        }
        if (has[q]) {
          wacc += x[q] * d;
          if (c4 == 0) dlsum += d;
        }
      }
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < T) {
#pragma unroll
          for (int q = 0; q < NR; ++q)
            if (has[q]) tacc[t] += o[q] * Us[t * g.Lc + r + q * RS];
        }
    } else {
#pragma unroll
      for (int q = 0; q < NR; ++q) o[q] = MODE == MODE_LEFT ? f32x4{0.f, 0.f, 0.f, 0.f} : x[q];
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < T) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(Vs + t * E + 4 * c4);
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            const float u = has[q] ? Us[t * g.Lc + r + q * RS] : 0.f;      // (u == 0 when the row does not exist)
            o[q] += v * u;
            if (MODE == MODE_FWD || MODE == MODE_LEFT) tacc[t] += x[q] * u;
          }
        }
      if (MODE == MODE_FWD) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          f32x4 sc;
          keep4v(g.keep, g.seed, g.thr, g.inv_keep, m[q] * CT + c4, sc);
#pragma unroll
#ifdef VQF_HIE_DIAG_NOTANH
          for (int j = 0; j < 4; ++j) o[q][j] = o[q][j] * sc[j];
#else
          for (int j = 0; j < 4; ++j) o[q][j] = vqf_tanh_fast(o[q][j]) * sc[j];
#endif
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q)
      if (has[q]) {
        *reinterpret_cast<f32x4*>(g.out + m[q] * g.ldo + 4 * c4) = o[q];
        if (COLS) csum += o[q];
      }
  }

  // `red` re-uses the LDS of the (T, E) / (T, Lc) operand images: nobody reads them any more behind this barrier
  __syncthreads();
  if (ACC) {
    // fold the RS row slots of the workgroup (fixed slot order), TG values of t per round through TG x RS x E floats of LDS:
    // four rounds of two barriers for T = 14 (round 4 folded one t at a time: 28 barriers across 16 waves)
    constexpr int TG = 4;
    float* dst = g.part + (((long long)s * g.N + n) * T) * g.ldp;
    const float* add = g.padd ? g.padd + (long long)n * T * g.ldpa : nullptr;
#pragma unroll
    for (int rd = 0; rd < (TMAX + TG - 1) / TG; ++rd) {
      if (rd * TG < T) {                                       // (uniform over the workgroup)
        if (rd) __syncthreads();
#pragma unroll
        for (int tg = 0; tg < TG; ++tg)
          if (rd * TG + tg < TMAX)
            *reinterpret_cast<f32x4*>(red + ((tg * RS + rs) * E) + 4 * c4) = tacc[rd * TG + tg < TMAX ? rd * TG + tg : 0];
        __syncthreads();
        for (int tg = rs; tg < TG; tg += RS) {                 // row slot tg adds the RS slots of t = rd * TG + tg
          const int t = rd * TG + tg;
          if (t < T) {
            f32x4 sum = *reinterpret_cast<const f32x4*>(red + (tg * RS) * E + 4 * c4);
            for (int q = 1; q < RS; ++q) sum += *reinterpret_cast<const f32x4*>(red + (tg * RS + q) * E + 4 * c4);
            if (add) sum = *reinterpret_cast<const f32x4*>(add + (long long)t * g.ldpa + 4 * c4) + sum;
            *reinterpret_cast<f32x4*>(dst + (long long)t * g.ldp + 4 * c4) = sum;
          }
        }
      }
    }
  }
  if (COLS && g.colpart) {                                     // (uniform) column sums of this workgroup's rows, row slots in order
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + rs * E + 4 * c4) = csum;
    __syncthreads();
    if (rs == 0) {
      f32x4 sum = *reinterpret_cast<const f32x4*>(red + 4 * c4);
      for (int q = 1; q < RS; ++q) sum += *reinterpret_cast<const f32x4*>(red + q * E + 4 * c4);
      *reinterpret_cast<f32x4*>(g.colpart + ((long long)s * g.N + n) * g.ldcp + 4 * c4) = sum;
    }
  }
  if (MODE == MODE_HEAD) {
    float* wrow = g.wpart + ((long long)s * g.N + n) * g.ldw;
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + rs * E + 4 * c4) = wacc;
    __syncthreads();
    if (rs == 0) {
      f32x4 sum = *reinterpret_cast<const f32x4*>(red + 4 * c4);
      for (int q = 1; q < RS; ++q) sum += *reinterpret_cast<const f32x4*>(red + q * E + 4 * c4);
      *reinterpret_cast<f32x4*>(wrow + 4 * c4) = sum;
    }
    __syncthreads();
    if (c4 == 0) red[rs] = dlsum;
    __syncthreads();
    if (tid == 0) {
      float sum = 0.f;
      for (int q = 0; q < RS; ++q) sum += red[q];
      wrow[E] = sum; wrow[E + 1] = 0.f; wrow[E + 2] = 0.f; wrow[E + 3] = 0.f;
    }
  }
}

// out[r, c] = (add ? add[r, c] : 0) + sum_{s < S} part[s][r][c]   (slabs summed in chunk order; rows of add / out may be strided)
__global__ void slab_sum_kernel(const float* __restrict__ part, int S, int R, int W4, const float* __restrict__ add, int lda,
                                float* __restrict__ out, int ldo) {
  const unsigned n4 = (unsigned)R * (unsigned)W4;
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const unsigned r = i / (unsigned)W4, c4 = i - r * (unsigned)W4;
    f32x4 v = add ? *reinterpret_cast<const f32x4*>(add + (long long)r * lda + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < S; ++q) v += *reinterpret_cast<const f32x4*>(part + ((long long)q * n4 + i) * 4);
    *reinterpret_cast<f32x4*>(out + (long long)r * ldo + 4 * c4) = v;
  }
}

// ---- the affinity products: C[n,t,l] = sum_e X[n,t,e] Y[n,l,e] (hieCoAtten.py:32 and its gradient) -------------------------------
// The inner-product shape of the ladder: T = 14 rows against L = 196, K = E.  On the 128x128-tile batched GEMM a 14-row A panel
// wastes 8/9 of every MFMA and the launch took 50 us for 103 MB of Y rows (2 TB/s).  Here a wave owns 16 rows of Y
// (v_mfma_f32_16x16x4_f32: A = the sample's (16, E) X image in LDS, rows T..15 zero; B = the wave's 16 Y rows straight from
// memory, each lane 32 contiguous bytes per 32-wide k step, eight steps requested before the first is consumed), one
// workgroup per sample (chunks of row groups for small batches: outputs are disjoint, nothing to reduce).  Optional second
// pair (X2, Y2) accumulated behind the first (dC = dti img_^T + que_ dtq^T in one launch) and the element-wise neighbour in the
// epilogue: EPI 1 = dropout(tanh(.)) (:32-33), EPI 2 = its backward given the forward's output.  The dropout index of (n,t,l)
// is that of the contiguous (N*T, L) tensor: same masks as vqf_tanh_dropout_fwd / _bwd.
struct AffArgs {
  const float* x1; int ldx1; const float* y1; int ldy1;
  const float* x2; int ldx2; const float* y2; int ldy2;
  const float* yprev; float* out;
  const uint8_t* keep; uint64_t seed; uint32_t thr; float inv_keep;
  int N, L, E, T;
};

__device__ __forceinline__ float keep1(const uint8_t* __restrict__ keep, uint64_t seed, uint32_t thr, float inv_keep, long long idx) {
  if (keep) return keep[idx] ? inv_keep : 0.f;
  if (thr == 0u) return 1.0f;
  const uint4 r = philox4x32_10((uint64_t)(idx >> 2), seed);
  const int j = (int)(idx & 3);
  const uint32_t v = j == 0 ? r.x : j == 1 ? r.y : j == 2 ? r.z : r.w;
  return v >= thr ? inv_keep : 0.f;
}

template <int EPI>
__global__ void __launch_bounds__(1024) hie_affinity_kernel(const AffArgs g) {
  extern __shared__ float smem[];
  const int E = g.E, T = g.T, L = g.L, ES = E + 4;           // + 4: the 16 rows of a b128 fragment read fall on distinct banks
  const int n = blockIdx.y, tid = threadIdx.x, W = blockDim.x >> 6, wave = tid >> 6, lane = tid & 63;
  const int npair = g.x2 ? 2 : 1, CT = E >> 2;
  for (int p = 0; p < npair; ++p) {
    const float* x = p ? g.x2 : g.x1;
    const int ldx = p ? g.ldx2 : g.ldx1;
    float* Xs = smem + p * 16 * ES;
    for (int i = tid; i < 16 * CT; i += blockDim.x) {
      const int t = i / CT, c = i - t * CT;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (t < T) v = *reinterpret_cast<const f32x4*>(x + (long long)(n * T + t) * ldx + 4 * c);
      *reinterpret_cast<f32x4*>(Xs + t * ES + 4 * c) = v;
    }
  }
  __syncthreads();
  const int G = (L + 15) >> 4;
  const int r = lane & 15, kq = lane >> 4;
  for (int grp = blockIdx.x * W + wave; grp < G; grp += gridDim.x * W) {
    const int l = grp * 16 + r;
    const long long row = (long long)n * L + (l < L ? l : L - 1);      // rows past L: a valid row, result not stored
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < npair; ++p) {
      const float* y = (p ? g.y2 : g.y1) + row * (p ? g.ldy2 : g.ldy1) + 8 * kq;
      const float* xs = smem + p * 16 * ES + r * ES + 8 * kq;
      for (int k0 = 0; k0 < E; k0 += 256) {
        f32x4 b[16];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + 32 * u < E) {                                       // (uniform)
            b[2 * u] = vqf_ld_stream(reinterpret_cast<const f32x4*>(y + k0 + 32 * u));
            b[2 * u + 1] = vqf_ld_stream(reinterpret_cast<const f32x4*>(y + k0 + 32 * u + 4));
          }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + 32 * u < E) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(xs + k0 + 32 * u);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(xs + k0 + 32 * u + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b[2 * u][j], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b[2 * u + 1][j], acc, 0, 0, 0);
          }
      }
    }
    if (l < L) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = 4 * kq + j;                                      // D: rows 4 (lane / 16) + j, column lane % 16
        if (t < T) {
          const long long idx = ((long long)n * T + t) * L + l;
          float v = acc[j];
          if (EPI == 1) {
            v = vqf_tanh_fast(v) * keep1(g.keep, g.seed, g.thr, g.inv_keep, idx);
          } else if (EPI == 2) {
            const float sc = keep1(g.keep, g.seed, g.thr, g.inv_keep, idx);
            const float th = sc > 0.f ? g.yprev[idx] * (1.0f / g.inv_keep) : 0.f;
            v = v * sc * (1.0f - th * th);
          }
          g.out[idx] = v;
        }
      }
    }
  }
}

VqfDynLdsFlags g_aff_lds[3];

bool shape_ok(int N, int L, int E, int T) {
  if (N <= 0 || L <= 0 || E <= 0 || T <= 0 || N > 65535 || T > TMAX_ALL || (E % 4)) return false;
  const int CT = E / 4;
  return CT <= 256 && 256 % CT == 0;
}

// threads per workgroup: 1024, or fewer when a chunk has too few rows to give every row slot two rows per trip
int threads_for(int E, int Lc) {
  const int CT = E / 4;
  int nt = HT;
  while (nt > 256 && (nt / CT) * 2 > Lc) nt >>= 1;
  return nt;
}

int chunks_for(int N, int L) {
  const int cus = vqf_cu_count() > 0 ? vqf_cu_count() : 256;
  int S = (cus + N - 1) / N;                  // one workgroup per CU; a whole sample per workgroup once N >= the CU count
  if (S > (L + 7) / 8) S = (L + 7) / 8;       // at least 8 rows per chunk
  return S < 1 ? 1 : S;
}

size_t lds_bytes(int mode, int E, int T, int Lc) {
  const int RS = threads_for(E, Lc) / (E / 4);
  const size_t ops_f = (size_t)(mode != MODE_HEAD ? T * E : 0) + (size_t)T * Lc;     // operand images of the main loop ...
  const size_t red_f = (size_t)4 * RS * E;                                           // ... re-used by the fold (TG = 4 values of t per round)
  return sizeof(float) * (ops_f > red_f ? ops_f : red_f);
}

int launch(int mode, HieArgs& g, const uint8_t* keep, uint64_t seed, float p, hipStream_t s) {
  if (!shape_ok(g.N, g.L, g.E, g.T)) return VQF_E_UNSUPPORTED;
  g.S = chunks_for(g.N, g.L);
  g.Lc = (g.L + g.S - 1) / g.S;
  g.S = (g.L + g.Lc - 1) / g.Lc;
  const size_t lds = lds_bytes(mode, g.E, g.T, g.Lc);
  if (lds > 64 * 1024) return VQF_E_UNSUPPORTED;
  if (g.S > 1 && (g.padd || (g.part && g.ldp != g.E))) return VQF_E_BADARG;     // several chunks: contiguous slabs, summed by vqf_hie_slab_sum
  const int nt = threads_for(g.E, g.Lc);
  g.keep = keep; g.seed = seed;
  g.thr = (keep || p == 0.f) ? 0u : drop_threshold_host(p);
  g.inv_keep = (keep || p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
  const dim3 grid(g.S, g.N);
#define HIE_GO(TM)                                                                                              \
  switch (mode) {                                                                                               \
    case MODE_FWD:  VQF_LAUNCH(KID_HIE_FWD, (hie_stream_kernel<MODE_FWD, TM>), grid, dim3(nt), lds, s, g); break;   \
    case MODE_HEAD: VQF_LAUNCH(KID_HIE_HEAD, (hie_stream_kernel<MODE_HEAD, TM>), grid, dim3(nt), lds, s, g); break; \
    case MODE_ADD:  VQF_LAUNCH(KID_HIE_ADD, (hie_stream_kernel<MODE_ADD, TM>), grid, dim3(nt), lds, s, g); break;   \
    default:        VQF_LAUNCH(KID_HIE_LEFT, (hie_stream_kernel<MODE_LEFT, TM>), grid, dim3(nt), lds, s, g); break; \
  }
  if (g.T <= 8) { HIE_GO(8) } else if (g.T <= 14) { HIE_GO(14) } else { HIE_GO(16) }
#undef HIE_GO
  return vqf_last_error();
}

bool rows_ok(const float* p, int ld, int E) { return p && aligned16(p) && ld >= E && (ld % 4) == 0; }

}  // namespace

extern "C" {

int vqf_hie_stream_supported(int N, int L, int E, int T) {
  if (!shape_ok(N, L, E, T)) return 0;
  const int S = chunks_for(N, L), Lc = (L + S - 1) / S;
  return lds_bytes(MODE_FWD, E, T, Lc) <= 64 * 1024;
}

int vqf_hie_chunks(int N, int L) {
  const int S = chunks_for(N, L), Lc = (L + S - 1) / S;
  return (L + Lc - 1) / Lc;
}

int vqf_hie_slab_sum(const float* part, int S, int R, int W, const float* add, int lda, float* out, int ldo, void* stream) {
  if (!part || !out || S <= 0 || R <= 0 || W <= 0 || ldo < W || (add && lda < W)) return VQF_E_BADARG;
  if ((W % 4) || (ldo % 4) || (add && (lda % 4)) || (long long)R * (W / 4) >= (1LL << 31)) return VQF_E_UNSUPPORTED;
  if (!aligned16(part) || !aligned16(out) || (add && !aligned16(add))) return VQF_E_ALIGN;
  long long blocks = ((long long)R * (W / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  VQF_LAUNCH(KID_HIE_SLABSUM, slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, part, S, R, W / 4, add,
             lda, out, ldo);
  return vqf_last_error();
}

int vqf_hie_hv_fwd(const float* a, int lda, const float* C, const float* V, int ldv, const uint8_t* keep, uint64_t seed,
                   float p_drop, int N, int L, int E, int T, float* out, int ldo, float* part, int ldp, void* stream) {
  if (!C || !rows_ok(part, ldp, E) || !rows_ok(a, lda, E) || !rows_ok(V, ldv, E) || !rows_ok(out, ldo, E) || p_drop < 0.f ||
      p_drop >= 1.f)
    return VQF_E_BADARG;
  HieArgs g = {};
  g.a = a; g.lda = lda; g.out = out; g.ldo = ldo; g.U = C; g.V = V; g.ldv = ldv; g.part = part; g.ldp = ldp;
  g.N = N; g.L = L; g.E = E; g.T = T;
  return launch(MODE_FWD, g, keep, seed, p_drop, (hipStream_t)stream);
}

int vqf_hie_head_bwd(const float* hv, int ldh, const float* dl, const float* w, const float* C, const uint8_t* keep,
                     uint64_t seed, float p_drop, int N, int L, int E, int T, float* out, int ldo, float* part, int ldp,
                     const float* padd, int ldpa, float* wpart, int ldw, void* stream) {
  if (!C || !dl || !w || !rows_ok(part, ldp, E) || (padd && !rows_ok(padd, ldpa, E)) || !wpart || !aligned16(wpart) || ldw < E + 4 ||
      (ldw % 4) || !rows_ok(hv, ldh, E) || !rows_ok(out, ldo, E) || !aligned16(w) || p_drop < 0.f || p_drop >= 1.f)
    return VQF_E_BADARG;
  HieArgs g = {};
  g.a = hv; g.lda = ldh; g.out = out; g.ldo = ldo; g.U = C; g.part = part; g.ldp = ldp; g.padd = padd; g.ldpa = ldpa;
  g.dl = dl; g.w = w; g.wpart = wpart; g.ldw = ldw;
  g.N = N; g.L = L; g.E = E; g.T = T;
  return launch(MODE_HEAD, g, keep, seed, p_drop, (hipStream_t)stream);
}

int vqf_hie_rank_add(const float* a, int lda, const float* U, const float* V, int ldv, int N, int L, int E, int T, float* out,
                     int ldo, float* colpart, int ldcp, void* stream) {
  if (!U || !rows_ok(a, lda, E) || !rows_ok(V, ldv, E) || !rows_ok(out, ldo, E) || (colpart && !rows_ok(colpart, ldcp, E)))
    return VQF_E_BADARG;
  HieArgs g = {};
  g.a = a; g.lda = lda; g.out = out; g.ldo = ldo; g.U = U; g.V = V; g.ldv = ldv; g.colpart = colpart; g.ldcp = ldcp;
  g.N = N; g.L = L; g.E = E; g.T = T;
  return launch(MODE_ADD, g, nullptr, 0, 0.f, (hipStream_t)stream);
}

int vqf_hie_rank_left(const float* U, const float* V, int ldv, const float* z, int ldz, int N, int L, int E, int T, float* out,
                      int ldo, float* part, int ldp, float* colpart, int ldcp, void* stream) {
  if (!U || !rows_ok(part, ldp, E) || !rows_ok(z, ldz, E) || !rows_ok(V, ldv, E) || !rows_ok(out, ldo, E) ||
      (colpart && !rows_ok(colpart, ldcp, E)))
    return VQF_E_BADARG;
  HieArgs g = {};
  g.z = z; g.ldz = ldz; g.out = out; g.ldo = ldo; g.U = U; g.V = V; g.ldv = ldv; g.part = part; g.ldp = ldp;
  g.colpart = colpart; g.ldcp = ldcp;
  g.N = N; g.L = L; g.E = E; g.T = T;
  return launch(MODE_LEFT, g, nullptr, 0, 0.f, (hipStream_t)stream);
}

int vqf_hie_affinity_supported(int N, int L, int E, int T, int pairs) {
  if (N <= 0 || N > 65535 || L <= 0 || T <= 0 || T > 16 || E <= 0 || (E % 32) || pairs < 1 || pairs > 2) return 0;
  return (size_t)pairs * 16 * (E + 4) * sizeof(float) <= 160 * 1024;
}

int vqf_hie_affinity(const float* x1, int ldx1, const float* y1, int ldy1, const float* x2, int ldx2, const float* y2, int ldy2,
                     int epi, const float* yprev, const uint8_t* keep, uint64_t seed, float p_drop, int N, int L, int E, int T,
                     float* out, void* stream) {
  const int pairs = x2 ? 2 : 1;
  if (!out || !rows_ok(x1, ldx1, E) || !rows_ok(y1, ldy1, E) || (!x2) != (!y2) || (x2 && (!rows_ok(x2, ldx2, E) || !rows_ok(y2, ldy2, E))) ||
      epi < 0 || epi > 2 || (epi == 2 && !yprev) || p_drop < 0.f || p_drop >= 1.f)
    return VQF_E_BADARG;
  if (!vqf_hie_affinity_supported(N, L, E, T, pairs)) return VQF_E_UNSUPPORTED;
  AffArgs g = {};
  g.x1 = x1; g.ldx1 = ldx1; g.y1 = y1; g.ldy1 = ldy1; g.x2 = x2; g.ldx2 = ldx2; g.y2 = y2; g.ldy2 = ldy2;
  g.yprev = yprev; g.out = out; g.N = N; g.L = L; g.E = E; g.T = T;
  g.keep = epi ? keep : nullptr; g.seed = seed;
  g.thr = (!epi || keep || p_drop == 0.f) ? 0u : drop_threshold_host(p_drop);
  g.inv_keep = (epi && (keep || p_drop > 0.f)) ? 1.0f / (1.0f - p_drop) : 1.0f;
  const int cus = vqf_cu_count() > 0 ? vqf_cu_count() : 256;
  const int G = (L + 15) / 16;
  int S = (cus + N - 1) / N;                       // whole samples per workgroup once N >= the CU count
  if (S > G) S = G;
  int W = (G + S - 1) / S;
  if (W > 16) W = 16;
  const int lds = pairs * 16 * (E + 4) * (int)sizeof(float);
  const void* fn = epi == 0 ? (const void*)hie_affinity_kernel<0> : epi == 1 ? (const void*)hie_affinity_kernel<1> : (const void*)hie_affinity_kernel<2>;
  if (lds > 64 * 1024) {
    const int rc = vqf_set_dyn_lds(fn, lds, g_aff_lds[epi]);
    if (rc != VQF_OK) return rc;
  }
  const dim3 grid(S, N), block(64 * W);
  hipStream_t s = (hipStream_t)stream;
  switch (epi) {
    case 0:  VQF_LAUNCH(KID_HIE_AFF, hie_affinity_kernel<0>, grid, block, lds, s, g); break;
    case 1:  VQF_LAUNCH(KID_HIE_AFF, hie_affinity_kernel<1>, grid, block, lds, s, g); break;
    default: VQF_LAUNCH(KID_HIE_AFF, hie_affinity_kernel<2>, grid, block, lds, s, g); break;
  }
  return vqf_last_error();
}

}  // extern "C"
