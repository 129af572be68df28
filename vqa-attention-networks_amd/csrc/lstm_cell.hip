// Point-wise LSTM cell stages for LARGE per-step batches (the question encoder in its regular
// orientation: mfb.py:68-70, T = 14 steps of a B = N = 512 row batch; also MHB, mhb_coAtt.py:182-183, and
// MHBCoAtt with fix_lstm_orientation).  There the recurrent product h_{t-1} W_hh^T is a real GEMM
// ((512 x 1024) x (1024 x 4096)) and runs on vqf_gemm_f32 with the accumulate flag into the
// pre-computed input projection; these two kernels do everything between the GEMMs of consecutive
// steps in one pass each (HBM-bound, B x 4H floats).  PyTorch gate order i,f,g,o.
#include "common.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// gates (B,4H): pre-activations in, ACTIVATED gates out (kept for the backward)
__global__ void __launch_bounds__(256) lstm_cell_fwd_kernel(float* __restrict__ gates, const float* __restrict__ c_prev,
                                                            int B, int H, float* __restrict__ c_out,
                                                            float* __restrict__ h_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // over B * H/4 float4s
  const int H4v = H >> 2;
  if (i >= (long long)B * H4v) return;
  const int b = (int)(i / H4v), j = (int)(i % H4v) * 4;
  float* g = gates + (long long)b * 4 * H + j;
  f32x4 gi = *reinterpret_cast<f32x4*>(g), gf = *reinterpret_cast<f32x4*>(g + H);
  f32x4 gg = *reinterpret_cast<f32x4*>(g + 2 * H), go = *reinterpret_cast<f32x4*>(g + 3 * H);
  f32x4 cp = {0, 0, 0, 0};
  if (c_prev) cp = *reinterpret_cast<const f32x4*>(c_prev + (long long)b * H + j);
  f32x4 c, h;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    gi[e] = sigm(gi[e]); gf[e] = sigm(gf[e]); gg[e] = tanhf(gg[e]); go[e] = sigm(go[e]);
    c[e] = gf[e] * cp[e] + gi[e] * gg[e];
    h[e] = go[e] * tanhf(c[e]);
  }
  *reinterpret_cast<f32x4*>(g) = gi; *reinterpret_cast<f32x4*>(g + H) = gf;
  *reinterpret_cast<f32x4*>(g + 2 * H) = gg; *reinterpret_cast<f32x4*>(g + 3 * H) = go;
  *reinterpret_cast<f32x4*>(c_out + (long long)b * H + j) = c;
  *reinterpret_cast<f32x4*>(h_out + (long long)b * H + j) = h;
}

// dh = dhs_t (+ dh_carry);  dc = dc_carry + dh o (1 - tanh(c)^2);  dG = pre-activation gradients;  dc_carry = dc f
__global__ void __launch_bounds__(256) lstm_cell_bwd_kernel(const float* __restrict__ dhs_t,
                                                            const float* __restrict__ dh_carry,
                                                            const float* __restrict__ gates,
                                                            const float* __restrict__ c_t,
                                                            const float* __restrict__ c_prev, int first, int B, int H,
                                                            float* __restrict__ dc_carry, float* __restrict__ dG) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int H4v = H >> 2;
  if (i >= (long long)B * H4v) return;
  const int b = (int)(i / H4v), j = (int)(i % H4v) * 4;
  const long long bh = (long long)b * H + j;
  const float* g = gates + (long long)b * 4 * H + j;
  const f32x4 gi = *reinterpret_cast<const f32x4*>(g), gf = *reinterpret_cast<const f32x4*>(g + H);
  const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 2 * H), go = *reinterpret_cast<const f32x4*>(g + 3 * H);
  f32x4 dh = *reinterpret_cast<const f32x4*>(dhs_t + bh);
  if (dh_carry) dh += *reinterpret_cast<const f32x4*>(dh_carry + bh);
  const f32x4 c = *reinterpret_cast<const f32x4*>(c_t + bh);
  f32x4 cp = {0, 0, 0, 0}, dcc = {0, 0, 0, 0};
  if (c_prev) cp = *reinterpret_cast<const f32x4*>(c_prev + bh);
  if (!first) dcc = *reinterpret_cast<const f32x4*>(dc_carry + bh);      // first = last time step: no carry yet
  f32x4 d0, d1, d2, d3, dco;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float tc = tanhf(c[e]);
    const float dc = dcc[e] + dh[e] * go[e] * (1.0f - tc * tc);
    d0[e] = dc * gg[e] * gi[e] * (1.0f - gi[e]);
    d1[e] = dc * cp[e] * gf[e] * (1.0f - gf[e]);
    d2[e] = dc * gi[e] * (1.0f - gg[e] * gg[e]);
    d3[e] = dh[e] * tc * go[e] * (1.0f - go[e]);
    dco[e] = dc * gf[e];
  }
  float* d = dG + (long long)b * 4 * H + j;
  *reinterpret_cast<f32x4*>(d) = d0; *reinterpret_cast<f32x4*>(d + H) = d1;
  *reinterpret_cast<f32x4*>(d + 2 * H) = d2; *reinterpret_cast<f32x4*>(d + 3 * H) = d3;
  *reinterpret_cast<f32x4*>(dc_carry + bh) = dco;
}

}  // namespace

extern "C" {

int vqf_lstm_cell_fwd(float* gates, const float* c_prev, int B, int H, float* c_out, float* h_out, void* stream) {
  if (!gates || !c_out || !h_out || B <= 0 || H <= 0) return VQF_E_BADARG;
  if (H % 4) return VQF_E_UNSUPPORTED;
  if (!aligned16(gates) || !aligned16(c_out) || !aligned16(h_out) || (c_prev && !aligned16(c_prev))) return VQF_E_ALIGN;
  const long long n = (long long)B * (H / 4);
  vqf_prof_dims(B, H, 0);
  VQF_LAUNCH(KID_LSTM_CELL_FWD, lstm_cell_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
             (hipStream_t)stream, gates, c_prev, B, H, c_out, h_out);
  return vqf_last_error();
}

int vqf_lstm_cell_bwd(const float* dhs_t, const float* dh_carry, const float* gates, const float* c_t,
                      const float* c_prev, int first, int B, int H, float* dc_carry, float* dG, void* stream) {
  if (!dhs_t || !gates || !c_t || !dc_carry || !dG || B <= 0 || H <= 0) return VQF_E_BADARG;
  if (H % 4) return VQF_E_UNSUPPORTED;
  if (!aligned16(dhs_t) || !aligned16(gates) || !aligned16(c_t) || !aligned16(dc_carry) || !aligned16(dG) ||
      (dh_carry && !aligned16(dh_carry)) || (c_prev && !aligned16(c_prev)))
    return VQF_E_ALIGN;
  const long long n = (long long)B * (H / 4);
  vqf_prof_dims(B, H, 0);
  VQF_LAUNCH(KID_LSTM_CELL_BWD, lstm_cell_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
             (hipStream_t)stream, dhs_t, dh_carry, gates, c_t, c_prev, first, B, H, dc_carry, dG);
  return vqf_last_error();
}

}  // extern "C"
