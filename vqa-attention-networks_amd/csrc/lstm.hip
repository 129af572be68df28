// LSTM recursion of the question encoder for SMALL per-step batches.
//
// mhb_coAtt.py:27-36,72-74 builds the LSTM with batch_first=True but feeds it (T,N,.), so the
// reference recurs over the MINIBATCH axis: S = N = 512 sequential steps, each a (T=14) x 1024 by
// 1024 x 4096 product.  A library RNN spends ~35 ms per training step on that (3000+ launches of
// tiny GEMM / point-wise kernels); here one step is ONE kernel:
//
//   forward  step s:  pre = h_{s-1} W_hh^T (+ xw_s)  ->  i,f,o = sigmoid, g = tanh
//                     c_s = f c_{s-1} + i g,  h_s = o tanh(c_s)
//   backward step s:  dh = dhs_s + dG_{s+1} W_hh ;  dc, dG_s (pre-activation grads) ;  dc carry
//
// A workgroup owns 4 hidden units = 16 gate columns = one 16x16 MFMA tile (256 workgroups at
// H = 1024, one per CU); its waves split the reduction dimension and v_mfma_f32_16x16x4_f32 does
// the cross-lane sums (a VALU + shuffle formulation measured 21-28 us per step, of which most was
// the 340-shuffle reduction tree per wave).  Forward: every lane first issues ALL its W_hh loads and
// reads its h_{s-1} fragments (B x H fp32, 57 KB at B=14, shared by all workgroups) straight from L2
// (staging them in LDS measured 10.6 instead of 8.9 us per step).  Backward: dG_{s+1}
// (B x 4H) and the W_hh^T rows are read straight from L2.  W_hh (16 MB) stays resident in the
// Infinity Cache across the 512 steps.  The input
// projection xw = x W_ih^T + b_ih + b_hh and all weight gradients (dW_ih, dW_hh, dx) are plain
// GEMMs over the whole sequence and use vqf_gemm_f32.
// PyTorch gate order i,f,g,o; zero initial state (mfb.py:69, mhb_coAtt.py:72-74 pass no hx).
// Constraints: B <= 32, H in {256, 512, 768, 1024} (else VQF_E_UNSUPPORTED: the caller keeps nn.LSTM).
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef VQF_LSTM_DIRECT
#define VQF_LSTM_DIRECT 1
#endif
constexpr int UPB = 4;           // hidden units per workgroup = 16 gate columns = one MFMA 16x16 tile
constexpr int FWD_WAVES = 4;     // forward: K = H split over 4 waves
constexpr int BWD_WAVES = 8;     // backward: K = 4H split over 8 waves

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// One step is a (B <= 16) x K x 16 product; v_mfma_f32_16x16x4_f32 does the cross-lane reduction
// that a VALU formulation needs ~340 shuffles per wave for.  Operand maps (guide section 3):
//   A[i = lane & 15][k = lane >> 4],  B[k = lane >> 4][j = lane & 15],
//   D: col = lane & 15, row = 4 * (lane >> 4) + reg.
// Lane (r, g) loads a float4 of 4 consecutive k at k0 + 4g for row r of each operand; MFMA number jj
// of that chunk consumes element jj of both (same k on both sides).
//
// forward: A = h_{s-1} (rows b), B = W_hh rows of this workgroup's 16 gate columns n = gate*4 + u.
template <int KI>
__global__ void __launch_bounds__(64 * FWD_WAVES)
lstm_step_fwd_kernel(const float* __restrict__ xw_s, const float* __restrict__ w_hh,
                     const float* __restrict__ h_prev, const float* __restrict__ c_prev, int B,
                     float* __restrict__ h_out, float* __restrict__ c_out,
                     float* __restrict__ gates_out) {
  constexpr int H = 256 * KI, KW = H / FWD_WAVES, NC = KW / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
#if VQF_LSTM_DIRECT
  float* part = smem;                        // [FWD_WAVES][2][16][16]   (two 16-row batch halves)
#else
  float* hbuf = smem;                        // [B][H]
  float* part = smem + 32 * H;               // [FWD_WAVES][2][16][16]   (two 16-row batch halves)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * UPB;
  const int nh = (B + 15) >> 4;              // batch halves (1 or 2)

  if (h_prev) {
    f32x4 wv[NC];
    const float* wr = w_hh + (long long)((r >> 2) * H + u0 + (r & 3)) * H + wave * KW + 4 * g;
#pragma unroll
    for (int c = 0; c < NC; ++c) wv[c] = *reinterpret_cast<const f32x4*>(wr + 16 * c);
#if !VQF_LSTM_DIRECT
    for (int i = tid * 4; i < B * H; i += 64 * FWD_WAVES * 4)
      *reinterpret_cast<f32x4*>(hbuf + i) = *reinterpret_cast<const f32x4*>(h_prev + i);
    __syncthreads();
#endif
    for (int hb = 0; hb < nh; ++hb) {
      const int b = hb * 16 + r;
#if VQF_LSTM_DIRECT
      // A fragments straight from L2 (h_{s-1} is 56 KB, shared by all workgroups): no LDS staging, so the
      // kernel needs 4 KB of LDS and can share a CU with the image-projection GEMM of the side stream
      const float* hr = h_prev + (long long)(b < B ? b : 0) * H + wave * KW + 4 * g;
#else
      const float* hr = hbuf + (b < B ? b : 0) * H + wave * KW + 4 * g;
#endif
      const float keep = b < B ? 1.f : 0.f;
      f32x4 hv[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) hv[c] = *reinterpret_cast<const f32x4*>(hr + 16 * c);
      f32x4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < NC; c += 2) {
        const f32x4 h0 = hv[c] * keep, h1 = hv[c + 1] * keep;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(h0[jj], wv[c][jj], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(h1[jj], wv[c + 1][jj], acc1, 0, 0, 0);
        }
      }
      float* pw = part + ((wave * 2 + hb) * 16) * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) pw[(4 * g + q) * 16 + r] = acc0[q] + acc1[q];     // D[b = 4g+q][n = r]
    }
  }
  __syncthreads();
  if (tid < B * UPB) {
    const int b = tid / UPB, u = tid % UPB, col = u0 + u;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (h_prev) {
#pragma unroll
      for (int w = 0; w < FWD_WAVES; ++w) {
        const float* pw = part + ((w * 2 + (b >> 4)) * 16 + (b & 15)) * 16 + u;
#pragma unroll
        for (int k = 0; k < 4; ++k) pre[k] += pw[4 * k];
      }
    }
    const float* x = xw_s + (long long)b * 4 * H + col;
    const float gi = sigmoidf_(pre[0] + x[0]);
    const float gf = sigmoidf_(pre[1] + x[H]);
    const float gg = tanhf(pre[2] + x[2 * H]);
    const float go = sigmoidf_(pre[3] + x[3 * H]);
    const float cp = c_prev ? c_prev[(long long)b * H + col] : 0.f;
    const float c = gf * cp + gi * gg;
    const float h = go * tanhf(c);
    h_out[(long long)b * H + col] = h;
    c_out[(long long)b * H + col] = c;
    float* gt = gates_out + (long long)b * 4 * H + col;
    gt[0] = gi; gt[H] = gf; gt[2 * H] = gg; gt[3 * H] = go;
  }
}

// backward step s: dh_carry[b][u] = sum_j dG_{s+1}[b][j] W_hh[j][u] -> A = dG_{s+1} (rows b, read
// straight from L2), B = rows u0..u0+3 of W_hh^T (columns 4..15 of the tile are zero).
template <int KI>
__global__ void __launch_bounds__(64 * BWD_WAVES)
lstm_step_bwd_kernel(const float* __restrict__ dhs_s, const float* __restrict__ dg_next,
                     const float* __restrict__ w_hh_t, const float* __restrict__ gates_s,
                     const float* __restrict__ c_s, const float* __restrict__ c_prev,
                     float* __restrict__ dc_carry, int B, float* __restrict__ dg_s) {
  constexpr int H = 256 * KI, H4 = 4 * H, JW = H4 / BWD_WAVES, NC = JW / 16, CG = 8;   // CG chunks per group
  __shared__ float part[BWD_WAVES][2][16][UPB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * UPB;
  const int nh = (B + 15) >> 4;

  if (dg_next) {
    const bool wok = r < UPB;
    const float* wr = w_hh_t + (long long)(u0 + (wok ? r : 0)) * H4 + wave * JW + 4 * g;
    for (int hb = 0; hb < nh; ++hb) {
      const int b = hb * 16 + r;
      const bool aok = b < B;
      const float* ar = dg_next + (long long)(aok ? b : 0) * H4 + wave * JW + 4 * g;
      f32x4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      for (int c0 = 0; c0 < NC; c0 += CG) {
        f32x4 av[CG], wv[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
          av[c] = *reinterpret_cast<const f32x4*>(ar + 16 * (c0 + c));
          wv[c] = *reinterpret_cast<const f32x4*>(wr + 16 * (c0 + c));
        }
#pragma unroll
        for (int c = 0; c < CG; ++c) {
          if (!aok) av[c] = f32x4{0, 0, 0, 0};
          if (!wok) wv[c] = f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int c = 0; c < CG; c += 2)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][jj], wv[c][jj], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c + 1][jj], wv[c + 1][jj], acc1, 0, 0, 0);
          }
      }
      if (r < UPB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) part[wave][hb][4 * g + q][r] = acc0[q] + acc1[q];   // D[b = 4g+q][u = r]
      }
    }
  }
  __syncthreads();
  if (tid < B * UPB) {
    const int b = tid / UPB, u = tid % UPB, col = u0 + u;
    const long long bh = (long long)b * H + col;
    float dhc = 0.f;
    if (dg_next) {
#pragma unroll
      for (int w = 0; w < BWD_WAVES; ++w) dhc += part[w][b >> 4][b & 15][u];
    }
    const float dh = dhs_s[bh] + dhc;
    const float* gt = gates_s + (long long)b * H4 + col;
    const float gi = gt[0], gf = gt[H], gg = gt[2 * H], go = gt[3 * H];
    const float tc = tanhf(c_s[bh]);
    const float cp = c_prev ? c_prev[bh] : 0.f;
    const float dc = dc_carry[bh] + dh * go * (1.0f - tc * tc);
    float* d = dg_s + (long long)b * H4 + col;
    d[0] = dc * gg * gi * (1.0f - gi);
    d[H] = dc * cp * gf * (1.0f - gf);
    d[2 * H] = dc * gi * (1.0f - gg * gg);
    d[3 * H] = dh * tc * go * (1.0f - go);
    dc_carry[bh] = dc * gf;
  }
}

template <typename K>
int set_smem(K kern, size_t bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? VQF_OK : (int)e;
}

template <int KI>
int run_fwd(const float* xw, const float* w_hh, int S, int B, float* hs, float* cs, float* gates,
            hipStream_t s) {
  constexpr int H = 256 * KI;
#if VQF_LSTM_DIRECT
  const size_t smem = (size_t)FWD_WAVES * 2 * 256 * sizeof(float);
#else
  const size_t smem = ((size_t)32 * H + (size_t)FWD_WAVES * 2 * 256) * sizeof(float);
#endif
  int rc = set_smem(lstm_step_fwd_kernel<KI>, smem);
  if (rc) return rc;
  const long long bh = (long long)B * H;
  dim3 grid(H / UPB);
  (void)hipGetLastError();
  if (g_vqf_prof_on) { vqf_prof_dims(S, B, H); vqf_prof_begin(KID_LSTM_FWD, s); }   // one bracket per sequence
  for (int t = 0; t < S; ++t) {
    const float* hp = t ? hs + (t - 1) * bh : nullptr;
    const float* cp = t ? cs + (t - 1) * bh : nullptr;
    hipLaunchKernelGGL(lstm_step_fwd_kernel<KI>, grid, dim3(64 * FWD_WAVES), smem, s, xw + t * 4 * bh, w_hh, hp, cp, B,
                       hs + t * bh, cs + t * bh, gates + t * 4 * bh);
  }
  if (g_vqf_prof_on) vqf_prof_end(KID_LSTM_FWD, s);
  return vqf_last_error();
}

template <int KI>
int run_bwd(const float* dhs, const float* gates, const float* cs, const float* w_hh_t, int S, int B,
            float* dgates, float* dc_carry, hipStream_t s) {
  constexpr int H = 256 * KI;
  const long long bh = (long long)B * H;
  dim3 grid(H / UPB);
  (void)hipGetLastError();
  if (g_vqf_prof_on) { vqf_prof_dims(S, B, H); vqf_prof_begin(KID_LSTM_BWD, s); }
  for (int t = S - 1; t >= 0; --t) {
    const float* dgn = (t + 1 < S) ? dgates + (long long)(t + 1) * 4 * bh : nullptr;
    const float* cp = t ? cs + (t - 1) * bh : nullptr;
    hipLaunchKernelGGL(lstm_step_bwd_kernel<KI>, grid, dim3(64 * BWD_WAVES), 0, s, dhs + t * bh, dgn, w_hh_t,
                       gates + t * 4 * bh, cs + t * bh, cp, dc_carry, B, dgates + t * 4 * bh);
  }
  if (g_vqf_prof_on) vqf_prof_end(KID_LSTM_BWD, s);
  return vqf_last_error();
}

}  // namespace

extern "C" {

int vqf_lstm_seq_supported(int B, int H) {
  return (B >= 1 && B <= 32 && (H == 256 || H == 512 || H == 768 || H == 1024)) ? 1 : 0;
}

int vqf_lstm_seq_fwd(const float* xw, const float* w_hh, int S, int B, int H, float* hs, float* cs,
                     float* gates, void* stream) {
  if (!xw || !w_hh || !hs || !cs || !gates || S <= 0) return VQF_E_BADARG;
  if (!vqf_lstm_seq_supported(B, H)) return VQF_E_UNSUPPORTED;
  if (!aligned16(xw) || !aligned16(w_hh) || !aligned16(hs)) return VQF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  switch (H / 256) {
    case 1: return run_fwd<1>(xw, w_hh, S, B, hs, cs, gates, s);
    case 2: return run_fwd<2>(xw, w_hh, S, B, hs, cs, gates, s);
    case 3: return run_fwd<3>(xw, w_hh, S, B, hs, cs, gates, s);
    default: return run_fwd<4>(xw, w_hh, S, B, hs, cs, gates, s);
  }
}

int vqf_lstm_seq_bwd(const float* dhs, const float* gates, const float* cs, const float* w_hh_t, int S,
                     int B, int H, float* dgates, float* dc_carry, void* stream) {
  if (!dhs || !gates || !cs || !w_hh_t || !dgates || !dc_carry || S <= 0) return VQF_E_BADARG;
  if (!vqf_lstm_seq_supported(B, H)) return VQF_E_UNSUPPORTED;
  if (!aligned16(dgates) || !aligned16(w_hh_t)) return VQF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(dc_carry, 0, (size_t)B * H * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  switch (H / 256) {
    case 1: return run_bwd<1>(dhs, gates, cs, w_hh_t, S, B, dgates, dc_carry, s);
    case 2: return run_bwd<2>(dhs, gates, cs, w_hh_t, S, B, dgates, dc_carry, s);
    case 3: return run_bwd<3>(dhs, gates, cs, w_hh_t, S, B, dgates, dc_carry, s);
    default: return run_bwd<4>(dhs, gates, cs, w_hh_t, S, B, dgates, dc_carry, s);
  }
}

}  // extern "C"
