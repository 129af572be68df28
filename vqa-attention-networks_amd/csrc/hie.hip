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
//
// a, z, out: rows m = n*L + l of 2-D tensors with row strides (column blocks of the concatenated-weight products, functions.
// HieCoreFn); U (N, T, L) contiguous; V rows n*T + t with stride ldv.  part: (S, N, T, E) partial sums over the S row chunks of
// a sample, summed afterwards in chunk order (deterministic, no atomics).  T <= 16, E % 4 == 0, (E / 4) divides 256.
#include "common.h"

namespace {

constexpr int HT = 256;        // threads per workgroup
constexpr int TMAX = 16;

enum { MODE_FWD = 0, MODE_HEAD = 1, MODE_ADD = 2, MODE_LEFT = 3 };

struct HieArgs {
  const float* a; int lda;
  const float* z; int ldz;
  float* out; int ldo;
  const float* U;
  const float* V; int ldv;
  float* part;
  const uint8_t* keep; uint64_t seed; uint32_t thr; float inv_keep;
  const float* dl; const float* w; float* wpart;     // HEAD: logit gradient (N*L), head weight (E), partial rows (S*N, E + 4)
  int N, L, E, T, S, Lc;
};

__device__ __forceinline__ void keep4v(const uint8_t* __restrict__ keep, uint64_t seed, uint32_t thr, float inv_keep,
                                       long long i4, f32x4& sc) {
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

template <int MODE>
__global__ void __launch_bounds__(HT) hie_stream_kernel(const HieArgs g) {
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
  float* red = Us + T * g.Lc;                                // [RS][E]       (ACC)
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
  if (MODE == MODE_HEAD) wv = *reinterpret_cast<const f32x4*>(g.w + 4 * c4);

  // two rows per trip: both rows' loads are issued before either is consumed
  for (int r = rs; r < rows; r += 2 * RS) {
    const int rA = r, rB = r + RS;
    const bool hasB = rB < rows;
    const long long mA = (long long)n * L + l0 + rA, mB = hasB ? mA + RS : mA;
    f32x4 xA = {0.f, 0.f, 0.f, 0.f}, xB = xA, zA = xA, zB = xA;
    if (MODE != MODE_LEFT) {
      xA = *reinterpret_cast<const f32x4*>(g.a + mA * g.lda + 4 * c4);
      xB = *reinterpret_cast<const f32x4*>(g.a + mB * g.lda + 4 * c4);
    } else {
      zA = *reinterpret_cast<const f32x4*>(g.z + mA * g.ldz + 4 * c4);
      zB = *reinterpret_cast<const f32x4*>(g.z + mB * g.ldz + 4 * c4);
    }
    f32x4 oA, oB;
    if (MODE == MODE_HEAD) {
      const float dA = g.dl[mA], dB = g.dl[mB];
      const float keep_q = 1.0f / g.inv_keep;
      f32x4 scA, scB;
      keep4v(g.keep, g.seed, g.thr, g.inv_keep, mA * CT + c4, scA);
      keep4v(g.keep, g.seed, g.thr, g.inv_keep, mB * CT + c4, scB);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float tA = scA[j] > 0.f ? xA[j] * keep_q : 0.f;       // tanh value back from the stored tanh * keep / (1 - p)
        const float tB = scB[j] > 0.f ? xB[j] * keep_q : 0.f;
        oA[j] = dA * wv[j] * scA[j] * (1.0f - tA * tA);
        oB[j] = dB * wv[j] * scB[j] * (1.0f - tB * tB);
      }
      wacc += xA * dA;
      if (c4 == 0) dlsum += dA;
      if (hasB) {
        wacc += xB * dB;
        if (c4 == 0) dlsum += dB;
      }
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < T) {
          tacc[t] += oA * Us[t * g.Lc + rA];
          if (hasB) tacc[t] += oB * Us[t * g.Lc + rB];
        }
    } else {
      oA = xA;
      oB = xB;
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < T) {
          const float uA = Us[t * g.Lc + rA], uB = hasB ? Us[t * g.Lc + rB] : 0.f;
          const f32x4 v = *reinterpret_cast<const f32x4*>(Vs + t * E + 4 * c4);
          oA += v * uA;
          oB += v * uB;
          if (MODE == MODE_FWD) { tacc[t] += xA * uA; tacc[t] += xB * uB; }       // (uB == 0 when the row does not exist)
          if (MODE == MODE_LEFT) { tacc[t] += zA * uA; tacc[t] += zB * uB; }
        }
      if (MODE == MODE_FWD) {
        f32x4 scA, scB;
        keep4v(g.keep, g.seed, g.thr, g.inv_keep, mA * CT + c4, scA);
        keep4v(g.keep, g.seed, g.thr, g.inv_keep, mB * CT + c4, scB);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          oA[j] = vqf_tanh_fast(oA[j]) * scA[j];
          oB[j] = vqf_tanh_fast(oB[j]) * scB[j];
        }
      }
    }
    *reinterpret_cast<f32x4*>(g.out + mA * g.ldo + 4 * c4) = oA;
    if (hasB) *reinterpret_cast<f32x4*>(g.out + mB * g.ldo + 4 * c4) = oB;
  }

  if (ACC) {
    // fold the RS row slots of the workgroup (fixed order), one t at a time through 4 KB of LDS
    float* dst = g.part + (((long long)s * g.N + n) * T) * E;
    for (int t = 0; t < T; ++t) {
      f32x4 v = tacc[0];
#pragma unroll
      for (int q = 1; q < TMAX; ++q)
        if (q == t) v = tacc[q];
      __syncthreads();
      *reinterpret_cast<f32x4*>(red + rs * E + 4 * c4) = v;
      __syncthreads();
      if (rs == 0) {
        f32x4 sum = *reinterpret_cast<const f32x4*>(red + 4 * c4);
        for (int q = 1; q < RS; ++q) sum += *reinterpret_cast<const f32x4*>(red + q * E + 4 * c4);
        *reinterpret_cast<f32x4*>(dst + (long long)t * E + 4 * c4) = sum;
      }
    }
  }
  if (MODE == MODE_HEAD) {
    float* wrow = g.wpart + ((long long)s * g.N + n) * (E + 4);
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

bool shape_ok(int N, int L, int E, int T) {
  if (N <= 0 || L <= 0 || E <= 0 || T <= 0 || N > 65535 || T > TMAX || (E % 4)) return false;
  const int CT = E / 4;
  return CT <= HT && HT % CT == 0;
}

int chunks_for(int N, int L) {
  int S = (1024 + N - 1) / N;                 // >= ~1024 workgroups: four per CU
  if (S > (L + 7) / 8) S = (L + 7) / 8;       // at least 8 rows per chunk
  return S < 1 ? 1 : S;
}

size_t lds_bytes(int mode, int E, int T, int Lc) {
  const int RS = HT / (E / 4);
  return sizeof(float) * ((size_t)(mode != MODE_HEAD ? T * E : 0) + (size_t)T * Lc + (size_t)RS * E + 16);
}

int launch(int mode, HieArgs& g, const uint8_t* keep, uint64_t seed, float p, hipStream_t s) {
  if (!shape_ok(g.N, g.L, g.E, g.T)) return VQF_E_UNSUPPORTED;
  g.S = chunks_for(g.N, g.L);
  g.Lc = (g.L + g.S - 1) / g.S;
  g.S = (g.L + g.Lc - 1) / g.Lc;
  const size_t lds = lds_bytes(mode, g.E, g.T, g.Lc);
  if (lds > 64 * 1024) return VQF_E_UNSUPPORTED;
  g.keep = keep; g.seed = seed;
  g.thr = (keep || p == 0.f) ? 0u : drop_threshold_host(p);
  g.inv_keep = (keep || p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
  const dim3 grid(g.S, g.N);
  switch (mode) {
    case MODE_FWD:  VQF_LAUNCH(KID_HIE_FWD, hie_stream_kernel<MODE_FWD>, grid, dim3(HT), lds, s, g); break;
    case MODE_HEAD: VQF_LAUNCH(KID_HIE_HEAD, hie_stream_kernel<MODE_HEAD>, grid, dim3(HT), lds, s, g); break;
    case MODE_ADD:  VQF_LAUNCH(KID_HIE_ADD, hie_stream_kernel<MODE_ADD>, grid, dim3(HT), lds, s, g); break;
    default:        VQF_LAUNCH(KID_HIE_LEFT, hie_stream_kernel<MODE_LEFT>, grid, dim3(HT), lds, s, g); break;
  }
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
                   float p_drop, int N, int L, int E, int T, float* out, int ldo, float* part, void* stream) {
  if (!C || !part || !rows_ok(a, lda, E) || !rows_ok(V, ldv, E) || !rows_ok(out, ldo, E) || p_drop < 0.f || p_drop >= 1.f)
    return VQF_E_BADARG;
  HieArgs g = {};
  g.a = a; g.lda = lda; g.out = out; g.ldo = ldo; g.U = C; g.V = V; g.ldv = ldv; g.part = part;
  g.N = N; g.L = L; g.E = E; g.T = T;
  return launch(MODE_FWD, g, keep, seed, p_drop, (hipStream_t)stream);
}

int vqf_hie_head_bwd(const float* hv, int ldh, const float* dl, const float* w, const float* C, const uint8_t* keep,
                     uint64_t seed, float p_drop, int N, int L, int E, int T, float* out, int ldo, float* part, float* wpart,
                     void* stream) {
  if (!C || !dl || !w || !part || !wpart || !rows_ok(hv, ldh, E) || !rows_ok(out, ldo, E) || !aligned16(w) || p_drop < 0.f ||
      p_drop >= 1.f)
    return VQF_E_BADARG;
  HieArgs g = {};
  g.a = hv; g.lda = ldh; g.out = out; g.ldo = ldo; g.U = C; g.part = part; g.dl = dl; g.w = w; g.wpart = wpart;
  g.N = N; g.L = L; g.E = E; g.T = T;
  return launch(MODE_HEAD, g, keep, seed, p_drop, (hipStream_t)stream);
}

int vqf_hie_rank_add(const float* a, int lda, const float* U, const float* V, int ldv, int N, int L, int E, int T, float* out,
                     int ldo, void* stream) {
  if (!U || !rows_ok(a, lda, E) || !rows_ok(V, ldv, E) || !rows_ok(out, ldo, E)) return VQF_E_BADARG;
  HieArgs g = {};
  g.a = a; g.lda = lda; g.out = out; g.ldo = ldo; g.U = U; g.V = V; g.ldv = ldv;
  g.N = N; g.L = L; g.E = E; g.T = T;
  return launch(MODE_ADD, g, nullptr, 0, 0.f, (hipStream_t)stream);
}

int vqf_hie_rank_left(const float* U, const float* V, int ldv, const float* z, int ldz, int N, int L, int E, int T, float* out,
                      int ldo, float* part, void* stream) {
  if (!U || !part || !rows_ok(z, ldz, E) || !rows_ok(V, ldv, E) || !rows_ok(out, ldo, E)) return VQF_E_BADARG;
  HieArgs g = {};
  g.z = z; g.ldz = ldz; g.out = out; g.ldo = ldo; g.U = U; g.V = V; g.ldv = ldv; g.part = part;
  g.N = N; g.L = L; g.E = E; g.T = T;
  return launch(MODE_LEFT, g, nullptr, 0, 0.f, (hipStream_t)stream);
}

}  // extern "C"
