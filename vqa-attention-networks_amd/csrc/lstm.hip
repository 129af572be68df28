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
// the 340-shuffle reduction tree per wave).
//
// A step is a chain of L2 round trips, so the operands are kept in FRAGMENT-MAJOR layouts: the 16
// bytes lane l needs for chunk c of an operand sit at ((c * 64) + l) * 16, i.e. every wave-wide load
// is one contiguous 1 KB read of 8 whole cache lines instead of 16 half lines from 16 rows
// (7.8 -> 6.0 us per forward step, 13.5 -> 9.9 us per backward step):
//   Wf  [q][kc][lane][4]   forward  B operand of workgroup q: W_hh[gate*H + 4q + u][16kc + 4g + e],
//                          lane = 16g + (gate*4 + u)          (packed once per sequence)
//   Wb  [q][jc][g][u][4]   backward B operand: W_hh[16jc + 4g + e][4q + u]   (tile columns 4..15 are 0)
//   hf  [half][kc][lane][4]   h_s  as the next step's A operand: h[16 half + r][16kc + 4g + e], lane = 16g + r
//   dgf [half][jc][lane][4]   dG_s as the previous step's A operand
// hf / dgf are written by the point-wise tail of each step next to the regular (S,B,.) outputs
// (double-buffered, rows >= B stay zero).  The input projection xw = x W_ih^T + b_ih + b_hh and all
// weight gradients (dW_ih, dW_hh, dx) are plain GEMMs over the whole sequence (vqf_gemm_f32).
// Other forms that were measured: h_{s-1} staged in LDS (10.6 us per forward step), operands read
// in their natural row-major layouts (7.8 / 13.5 us), a v_mfma_f32_4x4x1_16B_f32 backward that wastes
// no tile columns (17.0 us: the step is bound by its load chain, not by the 3.4 us of MFMAs), 8 or 16
// hidden units per workgroup in the backward (14.5 / 17.1 us), and a whole-sequence persistent
// kernel (not in the product build since ABI 5: tools/variants/lstm_persist.hip).
// PyTorch gate order i,f,g,o; zero initial state (mfb.py:69, mhb_coAtt.py:72-74 pass no hx).
// Constraints: B <= 32, H in {256, 512, 768, 1024} (else VQF_E_UNSUPPORTED: the caller keeps nn.LSTM).
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int UPB = 4;           // hidden units per workgroup = 16 gate columns = one MFMA 16x16 tile
constexpr int FWD_WAVES = 4;     // forward: K = H split over 4 waves
constexpr int BWD_WAVES = 8;     // backward: K = 4H split over 8 waves

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- one-time packing of W_hh into the two fragment-major images ---------------------------------
__global__ void pack_w_fwd_kernel(const float* __restrict__ w_hh, int H, float* __restrict__ wf) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // float4 index
  const int KC = H / 16;
  if (i >= (long long)(H / UPB) * KC * 64) return;
  const int lane = (int)(i & 63), kc = (int)((i >> 6) % KC), q = (int)((i >> 6) / KC);
  const int r = lane & 15, g = lane >> 4;
  const float* src = w_hh + (long long)((r >> 2) * H + q * UPB + (r & 3)) * H + 16 * kc + 4 * g;
  reinterpret_cast<f32x4*>(wf)[i] = *reinterpret_cast<const f32x4*>(src);
}
__global__ void pack_w_bwd_kernel(const float* __restrict__ w_hh, int H, float* __restrict__ wb) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // float4 index
  const int JC = 4 * H / 16;
  if (i >= (long long)(H / UPB) * JC * 16) return;
  const int u = (int)(i & 3), g = (int)((i >> 2) & 3), jc = (int)((i >> 4) % JC), q = (int)((i >> 4) / JC);
  const float* src = w_hh + (long long)(16 * jc + 4 * g) * H + q * UPB + u;
  f32x4 v = {src[0], src[H], src[2 * H], src[3 * H]};
  reinterpret_cast<f32x4*>(wb)[i] = v;
}

// offset (floats) of element (b, k) inside a fragment-major A image with KC chunks of 16 per row
__device__ __forceinline__ long long frag_off(int b, int k, int KC) {
  return (((long long)(b >> 4) * KC + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (b & 15)) * 4 + (k & 3);
}

// One step is a (B <= 16 per half) x K x 16 product.  Operand maps of v_mfma_f32_16x16x4_f32:
//   A[i = lane & 15][k = lane >> 4],  B[k = lane >> 4][j = lane & 15],
//   D: col = lane & 15, row = 4 * (lane >> 4) + reg.
// Lane (r, g) holds 4 consecutive k (k0 + 4g .. +3) of row r of each operand per 16-wide chunk; MFMA
// number e of that chunk consumes element e of both (same k on both sides).
template <int KI>
__global__ void __launch_bounds__(64 * FWD_WAVES)
lstm_step_fwd_kernel(const float* __restrict__ xw_s, const float* __restrict__ wf,
                     const float* __restrict__ hf_prev, const float* __restrict__ c_prev, int B,
                     float* __restrict__ h_out, float* __restrict__ c_out,
                     float* __restrict__ gates_out, float* __restrict__ hf_out) {
  constexpr int H = 256 * KI, KC = H / 16, NC = KC / FWD_WAVES;
  __shared__ float part[FWD_WAVES][2][16][16];       // [wave][half][b][n = gate*4 + u]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * UPB;
  const int nh = (B + 15) >> 4;              // batch halves (1 or 2)

  // the (b,u) owner threads fetch their point-wise operands first: xw streams from HBM (~2 us), and
  // that latency hides behind the operand loads and the MFMA chain instead of following them
  const bool own = tid < B * UPB;
  const int ob = tid / UPB, ou = tid % UPB, ocol = u0 + ou;
  float xq[4] = {0.f, 0.f, 0.f, 0.f}, cpq = 0.f;
  if (own) {
    const float* x = xw_s + (long long)ob * 4 * H + ocol;
    xq[0] = x[0]; xq[1] = x[H]; xq[2] = x[2 * H]; xq[3] = x[3 * H];
    if (c_prev) cpq = c_prev[(long long)ob * H + ocol];
  }

  if (hf_prev) {
    f32x4 wv[NC];
    const float* wp = wf + (((long long)blockIdx.x * KC + wave * NC) * 64 + lane) * 4;
#pragma unroll
    for (int c = 0; c < NC; ++c) wv[c] = *reinterpret_cast<const f32x4*>(wp + 256 * c);
    for (int hb = 0; hb < nh; ++hb) {
      const float* hp = hf_prev + (((long long)hb * KC + wave * NC) * 64 + lane) * 4;
      f32x4 hv[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) hv[c] = *reinterpret_cast<const f32x4*>(hp + 256 * c);
      __builtin_amdgcn_sched_barrier(0);     // all loads in flight before the first MFMA
      f32x4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < NC; c += 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[c][e], wv[c][e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[c + 1][e], wv[c + 1][e], acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) part[wave][hb][4 * g + q][r] = acc0[q] + acc1[q];     // D[b = 4g+q][n = r]
    }
  }
  __syncthreads();
  if (own) {
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (hf_prev) {
#pragma unroll
      for (int w = 0; w < FWD_WAVES; ++w)
#pragma unroll
        for (int k = 0; k < 4; ++k) pre[k] += part[w][ob >> 4][ob & 15][4 * k + ou];
    }
    const float gi = sigmoidf_(pre[0] + xq[0]);
    const float gf = sigmoidf_(pre[1] + xq[1]);
    const float gg = tanhf(pre[2] + xq[2]);
    const float go = sigmoidf_(pre[3] + xq[3]);
    const float c = gf * cpq + gi * gg;
    const float h = go * tanhf(c);
    h_out[(long long)ob * H + ocol] = h;
    hf_out[frag_off(ob, ocol, KC)] = h;
    c_out[(long long)ob * H + ocol] = c;
    float* gt = gates_out + (long long)ob * 4 * H + ocol;
    gt[0] = gi; gt[H] = gf; gt[2 * H] = gg; gt[3 * H] = go;
  }
}

// backward step s: dh_carry[b][u] = sum_j dG_{s+1}[b][j] W_hh[j][u] -> A = dG_{s+1} (rows b),
// B = rows u0..u0+3 of W_hh^T (columns 4..15 of the tile are zero: lanes r >= 4 load nothing).
template <int KI>
__global__ void __launch_bounds__(64 * BWD_WAVES)
lstm_step_bwd_kernel(const float* __restrict__ dhs_s, const float* __restrict__ dgf_next,
                     const float* __restrict__ wb, const float* __restrict__ gates_s,
                     const float* __restrict__ c_s, const float* __restrict__ c_prev,
                     float* __restrict__ dc_carry, int B, float* __restrict__ dg_s,
                     float* __restrict__ dgf_out) {
  constexpr int H = 256 * KI, H4 = 4 * H, JC = H4 / 16, NC = JC / BWD_WAVES, CG = NC < 8 ? NC : 8;
  __shared__ float part[BWD_WAVES][2][16][UPB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * UPB;
  const int nh = (B + 15) >> 4;

  const bool own = tid < B * UPB;            // point-wise operands first (see the forward kernel)
  const int ob = tid / UPB, ou = tid % UPB, ocol = u0 + ou;
  const long long obh = (long long)ob * H + ocol;
  float q_dh = 0.f, q_gi = 0.f, q_gf = 0.f, q_gg = 0.f, q_go = 0.f, q_c = 0.f, q_cp = 0.f, q_dc = 0.f;
  if (own) {
    q_dh = dhs_s[obh];
    const float* gt = gates_s + (long long)ob * H4 + ocol;
    q_gi = gt[0]; q_gf = gt[H]; q_gg = gt[2 * H]; q_go = gt[3 * H];
    q_c = c_s[obh];
    q_cp = c_prev ? c_prev[obh] : 0.f;
    q_dc = dc_carry[obh];
  }

  if (dgf_next) {
    const bool wok = r < UPB;
    const float* wp = wb + (((long long)blockIdx.x * JC + wave * NC) * 16 + g * 4 + (wok ? r : 0)) * 4;
    for (int hb = 0; hb < nh; ++hb) {
      const float* ap = dgf_next + (((long long)hb * JC + wave * NC) * 64 + lane) * 4;
      f32x4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      for (int c0 = 0; c0 < NC; c0 += CG) {
        f32x4 av[CG], wv[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
          av[c] = *reinterpret_cast<const f32x4*>(ap + 256 * (c0 + c));
          wv[c] = wok ? *reinterpret_cast<const f32x4*>(wp + 64 * (c0 + c)) : f32x4{0, 0, 0, 0};
        }
        __builtin_amdgcn_sched_barrier(0);   // the whole group of loads in flight before the first MFMA
#pragma unroll
        for (int c = 0; c < CG; c += 2)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][e], wv[c][e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c + 1][e], wv[c + 1][e], acc1, 0, 0, 0);
          }
      }
      if (r < UPB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) part[wave][hb][4 * g + q][r] = acc0[q] + acc1[q];   // D[b = 4g+q][u = r]
      }
    }
  }
  __syncthreads();
  if (own) {
    float dhc = 0.f;
    if (dgf_next) {
#pragma unroll
      for (int w = 0; w < BWD_WAVES; ++w) dhc += part[w][ob >> 4][ob & 15][ou];
    }
    const float dh = q_dh + dhc;
    const float gi = q_gi, gf = q_gf, gg = q_gg, go = q_go;
    const float tc = tanhf(q_c);
    const float dc = q_dc + dh * go * (1.0f - tc * tc);
    const float d0 = dc * gg * gi * (1.0f - gi);
    const float d1 = dc * q_cp * gf * (1.0f - gf);
    const float d2 = dc * gi * (1.0f - gg * gg);
    const float d3 = dh * tc * go * (1.0f - go);
    float* d = dg_s + (long long)ob * H4 + ocol;
    d[0] = d0; d[H] = d1; d[2 * H] = d2; d[3 * H] = d3;
    dgf_out[frag_off(ob, ocol, JC)] = d0;
    dgf_out[frag_off(ob, H + ocol, JC)] = d1;
    dgf_out[frag_off(ob, 2 * H + ocol, JC)] = d2;
    dgf_out[frag_off(ob, 3 * H + ocol, JC)] = d3;
    dc_carry[obh] = dc * gf;
  }
}

// ================================================================================================
// bf16-operand variant of the recursion (bf16 mode of MHBCoAtt, BASELINE config 3): W_hh and the
// recurrent operand (h_{s-1} / dG_{s+1}) enter v_mfma_f32_16x16x32_bf16 as bf16, accumulation, gates,
// cell state and every stored tensor stay fp32.  Same decomposition and fragment-major layouts with
// 32-k chunks (a lane's 8 bf16 = 16 bytes): half the operand bytes per step and 1/8 of the MFMA
// instructions (the 75 %-empty backward tile then costs 0.2 instead of 3.4 us).
//   A[i = lane & 15][k = 8 (lane >> 4) .. +7],  B[k = 8 (lane >> 4) .. +7][j = lane & 15],  D as above.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void pack_w_fwd_bf16_kernel(const float* __restrict__ w_hh, int H, __bf16* __restrict__ wf) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // 16-byte (8 x bf16) index
  const int KC = H / 32;
  if (i >= (long long)(H / UPB) * KC * 64) return;
  const int lane = (int)(i & 63), kc = (int)((i >> 6) % KC), q = (int)((i >> 6) / KC);
  const int r = lane & 15, g = lane >> 4;
  const float* src = w_hh + (long long)((r >> 2) * H + q * UPB + (r & 3)) * H + 32 * kc + 8 * g;
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (__bf16)src[e];
  reinterpret_cast<bf16x8*>(wf)[i] = v;
}
__global__ void pack_w_bwd_bf16_kernel(const float* __restrict__ w_hh, int H, __bf16* __restrict__ wb) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int JC = 4 * H / 32;
  if (i >= (long long)(H / UPB) * JC * 16) return;
  const int u = (int)(i & 3), g = (int)((i >> 2) & 3), jc = (int)((i >> 4) % JC), q = (int)((i >> 4) / JC);
  const float* src = w_hh + (long long)(32 * jc + 8 * g) * H + q * UPB + u;
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (__bf16)src[(long long)e * H];
  reinterpret_cast<bf16x8*>(wb)[i] = v;
}

// offset (bf16 elements) of element (b, k) in a fragment-major bf16 A image with KC chunks of 32 per row
__device__ __forceinline__ long long frag_off16(int b, int k, int KC) {
  return (((long long)(b >> 4) * KC + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (b & 15)) * 8 + (k & 7);
}

template <int KI>
__global__ void __launch_bounds__(64 * FWD_WAVES)
lstm_step_fwd_bf16_kernel(const float* __restrict__ xw_s, const __bf16* __restrict__ wf,
                          const __bf16* __restrict__ hf_prev, const float* __restrict__ c_prev, int B,
                          float* __restrict__ h_out, float* __restrict__ c_out,
                          float* __restrict__ gates_out, __bf16* __restrict__ hf_out) {
  constexpr int H = 256 * KI, KC = H / 32, NC = KC / FWD_WAVES;
  __shared__ float part[FWD_WAVES][2][16][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * UPB;
  const int nh = (B + 15) >> 4;
  const bool own = tid < B * UPB;
  const int ob = tid / UPB, ou = tid % UPB, ocol = u0 + ou;
  float xq[4] = {0.f, 0.f, 0.f, 0.f}, cpq = 0.f;
  if (own) {
    const float* x = xw_s + (long long)ob * 4 * H + ocol;
    xq[0] = x[0]; xq[1] = x[H]; xq[2] = x[2 * H]; xq[3] = x[3 * H];
    if (c_prev) cpq = c_prev[(long long)ob * H + ocol];
  }
  if (hf_prev) {
    bf16x8 wv[NC];
    const __bf16* wp = wf + (((long long)blockIdx.x * KC + wave * NC) * 64 + lane) * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) wv[c] = *reinterpret_cast<const bf16x8*>(wp + 512 * c);
    for (int hb = 0; hb < nh; ++hb) {
      const __bf16* hp = hf_prev + (((long long)hb * KC + wave * NC) * 64 + lane) * 8;
      bf16x8 hv[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) hv[c] = *reinterpret_cast<const bf16x8*>(hp + 512 * c);
      __builtin_amdgcn_sched_barrier(0);
      f32x4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < NC; c += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hv[c], wv[c], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hv[c + 1], wv[c + 1], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) part[wave][hb][4 * g + q][r] = acc0[q] + acc1[q];
    }
  }
  __syncthreads();
  if (own) {
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (hf_prev) {
#pragma unroll
      for (int w = 0; w < FWD_WAVES; ++w)
#pragma unroll
        for (int k = 0; k < 4; ++k) pre[k] += part[w][ob >> 4][ob & 15][4 * k + ou];
    }
    const float gi = sigmoidf_(pre[0] + xq[0]);
    const float gf = sigmoidf_(pre[1] + xq[1]);
    const float gg = tanhf(pre[2] + xq[2]);
    const float go = sigmoidf_(pre[3] + xq[3]);
    const float c = gf * cpq + gi * gg;
    const float h = go * tanhf(c);
    h_out[(long long)ob * H + ocol] = h;
    hf_out[frag_off16(ob, ocol, KC)] = (__bf16)h;
    c_out[(long long)ob * H + ocol] = c;
    float* gt = gates_out + (long long)ob * 4 * H + ocol;
    gt[0] = gi; gt[H] = gf; gt[2 * H] = gg; gt[3 * H] = go;
  }
}

template <int KI>
__global__ void __launch_bounds__(64 * BWD_WAVES)
lstm_step_bwd_bf16_kernel(const float* __restrict__ dhs_s, const __bf16* __restrict__ dgf_next,
                          const __bf16* __restrict__ wb, const float* __restrict__ gates_s,
                          const float* __restrict__ c_s, const float* __restrict__ c_prev,
                          float* __restrict__ dc_carry, int B, float* __restrict__ dg_s,
                          __bf16* __restrict__ dgf_out) {
  constexpr int H = 256 * KI, H4 = 4 * H, JC = H4 / 32, NC = JC / BWD_WAVES;
  __shared__ float part[BWD_WAVES][2][16][UPB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * UPB;
  const int nh = (B + 15) >> 4;
  const bool own = tid < B * UPB;
  const int ob = tid / UPB, ou = tid % UPB, ocol = u0 + ou;
  const long long obh = (long long)ob * H + ocol;
  float q_dh = 0.f, q_gi = 0.f, q_gf = 0.f, q_gg = 0.f, q_go = 0.f, q_c = 0.f, q_cp = 0.f, q_dc = 0.f;
  if (own) {
    q_dh = dhs_s[obh];
    const float* gt = gates_s + (long long)ob * H4 + ocol;
    q_gi = gt[0]; q_gf = gt[H]; q_gg = gt[2 * H]; q_go = gt[3 * H];
    q_c = c_s[obh];
    q_cp = c_prev ? c_prev[obh] : 0.f;
    q_dc = dc_carry[obh];
  }
  if (dgf_next) {
    const bool wok = r < UPB;
    const __bf16* wp = wb + (((long long)blockIdx.x * JC + wave * NC) * 16 + g * 4 + (wok ? r : 0)) * 8;
    bf16x8 wv[NC];
    const bf16x8 zero = __builtin_bit_cast(bf16x8, f32x4{0, 0, 0, 0});
#pragma unroll
    for (int c = 0; c < NC; ++c) wv[c] = wok ? *reinterpret_cast<const bf16x8*>(wp + 128 * c) : zero;
    for (int hb = 0; hb < nh; ++hb) {
      const __bf16* ap = dgf_next + (((long long)hb * JC + wave * NC) * 64 + lane) * 8;
      bf16x8 av[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) av[c] = *reinterpret_cast<const bf16x8*>(ap + 512 * c);
      __builtin_amdgcn_sched_barrier(0);
      f32x4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < NC; c += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[c], wv[c], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[c + 1], wv[c + 1], acc1, 0, 0, 0);
      }
      if (r < UPB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) part[wave][hb][4 * g + q][r] = acc0[q] + acc1[q];
      }
    }
  }
  __syncthreads();
  if (own) {
    float dhc = 0.f;
    if (dgf_next) {
#pragma unroll
      for (int w = 0; w < BWD_WAVES; ++w) dhc += part[w][ob >> 4][ob & 15][ou];
    }
    const float dh = q_dh + dhc;
    const float gi = q_gi, gf = q_gf, gg = q_gg, go = q_go;
    const float tc = tanhf(q_c);
    const float dc = q_dc + dh * go * (1.0f - tc * tc);
    const float d0 = dc * gg * gi * (1.0f - gi);
    const float d1 = dc * q_cp * gf * (1.0f - gf);
    const float d2 = dc * gi * (1.0f - gg * gg);
    const float d3 = dh * tc * go * (1.0f - go);
    float* d = dg_s + (long long)ob * H4 + ocol;
    d[0] = d0; d[H] = d1; d[2 * H] = d2; d[3 * H] = d3;
    dgf_out[frag_off16(ob, ocol, JC)] = (__bf16)d0;
    dgf_out[frag_off16(ob, H + ocol, JC)] = (__bf16)d1;
    dgf_out[frag_off16(ob, 2 * H + ocol, JC)] = (__bf16)d2;
    dgf_out[frag_off16(ob, 3 * H + ocol, JC)] = (__bf16)d3;
    dc_carry[obh] = dc * gf;
  }
}

// workspace: [ packed W_hh : 4H*H floats ][ fragment image 0 ][ fragment image 1 ]
size_t frag_floats(int H, bool bwd) { return (size_t)2 * ((bwd ? 4 * H : H) / 16) * 64 * 4; }   // 2 batch halves
size_t ws_need(int H, bool bwd) { return ((size_t)4 * H * H + 2 * frag_floats(H, bwd)) * sizeof(float); }

template <int KI>
int run_fwd_bf16(const float* xw, const float* w_hh, int S, int B, float* hs, float* cs, float* gates,
                 float* ws, hipStream_t s) {
  constexpr int H = 256 * KI;
  __bf16* wf = (__bf16*)ws;                                        // same workspace, half of it used
  __bf16* hf0 = (__bf16*)(ws + (size_t)4 * H * H);
  const size_t fe = (size_t)2 * (H / 32) * 64 * 8;                 // bf16 elements per fragment image
  __bf16* hf[2] = {hf0, hf0 + fe};
  hipError_t e = hipMemsetAsync(hf0, 0, 2 * fe * sizeof(__bf16), s);
  if (e != hipSuccess) return (int)e;
  const long long bh = (long long)B * H;
  dim3 grid(H / UPB);
  (void)hipGetLastError();
  const bool bracket = g_vqf_prof_on && (vqf_prof_dims(S, B, H), vqf_prof_begin(KID_LSTM_FWD, s));
  const long long n16 = (long long)(H / UPB) * (H / 32) * 64;
  hipLaunchKernelGGL(pack_w_fwd_bf16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, w_hh, H, wf);
  for (int t = 0; t < S; ++t) {
    const __bf16* hp = t ? hf[(t - 1) & 1] : nullptr;
    const float* cp = t ? cs + (t - 1) * bh : nullptr;
    hipLaunchKernelGGL(lstm_step_fwd_bf16_kernel<KI>, grid, dim3(64 * FWD_WAVES), 0, s, xw + t * 4 * bh,
                       (const __bf16*)wf, hp, cp, B, hs + t * bh, cs + t * bh, gates + t * 4 * bh, hf[t & 1]);
  }
  if (bracket) vqf_prof_end(KID_LSTM_FWD, s);
  return vqf_last_error();
}

template <int KI>
int run_bwd_bf16(const float* dhs, const float* gates, const float* cs, const float* w_hh, int S, int B,
                 float* dgates, float* dc_carry, float* ws, hipStream_t s) {
  constexpr int H = 256 * KI;
  __bf16* wb = (__bf16*)ws;
  __bf16* gf0 = (__bf16*)(ws + (size_t)4 * H * H);
  const size_t fe = (size_t)2 * (4 * H / 32) * 64 * 8;
  __bf16* gf[2] = {gf0, gf0 + fe};
  hipError_t e = hipMemsetAsync(gf0, 0, 2 * fe * sizeof(__bf16), s);
  if (e != hipSuccess) return (int)e;
  const long long bh = (long long)B * H;
  dim3 grid(H / UPB);
  (void)hipGetLastError();
  const bool bracket = g_vqf_prof_on && (vqf_prof_dims(S, B, H), vqf_prof_begin(KID_LSTM_BWD, s));
  const long long n16 = (long long)(H / UPB) * (4 * H / 32) * 16;
  hipLaunchKernelGGL(pack_w_bwd_bf16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, w_hh, H, wb);
  for (int t = S - 1; t >= 0; --t) {
    const __bf16* dgn = (t + 1 < S) ? gf[(t + 1) & 1] : nullptr;
    const float* cp = t ? cs + (t - 1) * bh : nullptr;
    hipLaunchKernelGGL(lstm_step_bwd_bf16_kernel<KI>, grid, dim3(64 * BWD_WAVES), 0, s, dhs + t * bh, dgn,
                       (const __bf16*)wb, gates + t * 4 * bh, cs + t * bh, cp, dc_carry, B, dgates + t * 4 * bh,
                       gf[t & 1]);
  }
  if (bracket) vqf_prof_end(KID_LSTM_BWD, s);
  return vqf_last_error();
}

template <int KI>
int run_fwd(const float* xw, const float* w_hh, int S, int B, float* hs, float* cs, float* gates,
            float* ws, hipStream_t s) {
  constexpr int H = 256 * KI;
  float* wf = ws;
  float* hf[2] = {ws + (size_t)4 * H * H, ws + (size_t)4 * H * H + frag_floats(H, false)};
  hipError_t e = hipMemsetAsync(hf[0], 0, 2 * frag_floats(H, false) * sizeof(float), s);   // rows >= B stay 0
  if (e != hipSuccess) return (int)e;
  const long long bh = (long long)B * H;
  dim3 grid(H / UPB);
  (void)hipGetLastError();
  const bool bracket = g_vqf_prof_on && (vqf_prof_dims(S, B, H), vqf_prof_begin(KID_LSTM_FWD, s));   // one bracket per sequence
  const long long n4 = (long long)(H / UPB) * (H / 16) * 64;
  hipLaunchKernelGGL(pack_w_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, w_hh, H, wf);
  for (int t = 0; t < S; ++t) {
    const float* hp = t ? hf[(t - 1) & 1] : nullptr;
    const float* cp = t ? cs + (t - 1) * bh : nullptr;
    hipLaunchKernelGGL(lstm_step_fwd_kernel<KI>, grid, dim3(64 * FWD_WAVES), 0, s, xw + t * 4 * bh,
                       (const float*)wf, hp, cp, B, hs + t * bh, cs + t * bh, gates + t * 4 * bh, hf[t & 1]);
  }
  if (bracket) vqf_prof_end(KID_LSTM_FWD, s);
  return vqf_last_error();
}

template <int KI>
int run_bwd(const float* dhs, const float* gates, const float* cs, const float* w_hh, int S, int B,
            float* dgates, float* dc_carry, float* ws, hipStream_t s) {
  constexpr int H = 256 * KI;
  float* wb = ws;
  float* gf[2] = {ws + (size_t)4 * H * H, ws + (size_t)4 * H * H + frag_floats(H, true)};
  hipError_t e = hipMemsetAsync(gf[0], 0, 2 * frag_floats(H, true) * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  const long long bh = (long long)B * H;
  dim3 grid(H / UPB);
  (void)hipGetLastError();
  const bool bracket = g_vqf_prof_on && (vqf_prof_dims(S, B, H), vqf_prof_begin(KID_LSTM_BWD, s));
  const long long n4 = (long long)(H / UPB) * (4 * H / 16) * 16;
  hipLaunchKernelGGL(pack_w_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, w_hh, H, wb);
  for (int t = S - 1; t >= 0; --t) {
    const float* dgn = (t + 1 < S) ? gf[(t + 1) & 1] : nullptr;
    const float* cp = t ? cs + (t - 1) * bh : nullptr;
    hipLaunchKernelGGL(lstm_step_bwd_kernel<KI>, grid, dim3(64 * BWD_WAVES), 0, s, dhs + t * bh, dgn,
                       (const float*)wb, gates + t * 4 * bh, cs + t * bh, cp, dc_carry, B, dgates + t * 4 * bh,
                       gf[t & 1]);
  }
  if (bracket) vqf_prof_end(KID_LSTM_BWD, s);
  return vqf_last_error();
}

}  // namespace

extern "C" {

int vqf_lstm_seq_supported(int B, int H) {
  return (B >= 1 && B <= 32 && (H == 256 || H == 512 || H == 768 || H == 1024)) ? 1 : 0;
}

size_t vqf_lstm_seq_ws_bytes(int B, int H) {
  return vqf_lstm_seq_supported(B, H) ? ws_need(H, true) : 0;
}

int vqf_lstm_seq_fwd(const float* xw, const float* w_hh, int S, int B, int H, float* hs, float* cs,
                     float* gates, int flags, void* ws, size_t ws_bytes, void* stream) {
  if (!xw || !w_hh || !hs || !cs || !gates || !ws || S <= 0) return VQF_E_BADARG;
  if (!vqf_lstm_seq_supported(B, H)) return VQF_E_UNSUPPORTED;
  if (!aligned16(xw) || !aligned16(w_hh) || !aligned16(hs) || !aligned16(ws)) return VQF_E_ALIGN;
  if (ws_bytes < ws_need(H, false)) return VQF_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (flags & VQF_LSTM_BF16) {
    switch (H / 256) {
      case 1: return run_fwd_bf16<1>(xw, w_hh, S, B, hs, cs, gates, (float*)ws, s);
      case 2: return run_fwd_bf16<2>(xw, w_hh, S, B, hs, cs, gates, (float*)ws, s);
      case 3: return run_fwd_bf16<3>(xw, w_hh, S, B, hs, cs, gates, (float*)ws, s);
      default: return run_fwd_bf16<4>(xw, w_hh, S, B, hs, cs, gates, (float*)ws, s);
    }
  }
  switch (H / 256) {
    case 1: return run_fwd<1>(xw, w_hh, S, B, hs, cs, gates, (float*)ws, s);
    case 2: return run_fwd<2>(xw, w_hh, S, B, hs, cs, gates, (float*)ws, s);
    case 3: return run_fwd<3>(xw, w_hh, S, B, hs, cs, gates, (float*)ws, s);
    default: return run_fwd<4>(xw, w_hh, S, B, hs, cs, gates, (float*)ws, s);
  }
}

int vqf_lstm_seq_bwd(const float* dhs, const float* gates, const float* cs, const float* w_hh, int S,
                     int B, int H, float* dgates, float* dc_carry, int flags, void* ws, size_t ws_bytes,
                     void* stream) {
  if (!dhs || !gates || !cs || !w_hh || !dgates || !dc_carry || !ws || S <= 0) return VQF_E_BADARG;
  if (!vqf_lstm_seq_supported(B, H)) return VQF_E_UNSUPPORTED;
  if (!aligned16(dgates) || !aligned16(w_hh) || !aligned16(ws)) return VQF_E_ALIGN;
  if (ws_bytes < ws_need(H, true)) return VQF_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(dc_carry, 0, (size_t)B * H * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  if (flags & VQF_LSTM_BF16) {
    switch (H / 256) {
      case 1: return run_bwd_bf16<1>(dhs, gates, cs, w_hh, S, B, dgates, dc_carry, (float*)ws, s);
      case 2: return run_bwd_bf16<2>(dhs, gates, cs, w_hh, S, B, dgates, dc_carry, (float*)ws, s);
      case 3: return run_bwd_bf16<3>(dhs, gates, cs, w_hh, S, B, dgates, dc_carry, (float*)ws, s);
      default: return run_bwd_bf16<4>(dhs, gates, cs, w_hh, S, B, dgates, dc_carry, (float*)ws, s);
    }
  }
  switch (H / 256) {
    case 1: return run_bwd<1>(dhs, gates, cs, w_hh, S, B, dgates, dc_carry, (float*)ws, s);
    case 2: return run_bwd<2>(dhs, gates, cs, w_hh, S, B, dgates, dc_carry, (float*)ws, s);
    case 3: return run_bwd<3>(dhs, gates, cs, w_hh, S, B, dgates, dc_carry, (float*)ws, s);
    default: return run_bwd<4>(dhs, gates, cs, w_hh, S, B, dgates, dc_carry, (float*)ws, s);
  }
}

}  // extern "C"
