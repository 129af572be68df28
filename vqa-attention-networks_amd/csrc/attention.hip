// Attention heads of the co-attention ladder (HBM-bound, wave64 reductions).
//
//   question side (S = T tokens):  mfb.py:81-89   / mhb_coAtt.py:83-91          (G = 2 glimpses)
//   image side    (S = 196 regions, C = 2048): mfb.py:114-123 / mhb_coAtt.py:113-121   (G = 2)
//   HieCoAtten av/aq (hieCoAtten.py:40-42,47-49) and Attention_1 (modules.py:60-65)     (G = 1)
//
// att_logits_*   : hidden (M,Hh) -> G logits per row, and its backward, optionally through the
//                  preceding ReLU (one pass over the hidden activations).
// glimpse_pool_* : softmax over the S positions of a sample (or the reference's singleton-axis
//                  softmax == 1, mfb.py:84,118) and the G glimpse-weighted sums over the (N,S,C)
//                  feature tensor.  The image tensor is streamed exactly once per pass with 16-byte
//                  coalesced loads along C; softmax rows (S <= 1024) live in LDS and are reduced
//                  with wavefront shuffles.
#include "common.h"

// one block per sample, one wave per position: 16 waves per block keep 2 blocks x 16 waves on a CU when N is only
// a few hundred (256 threads: 0.26 ms on the 822 MB image pass, 3.1 TB/s)
#ifndef VQF_GLIMPSE_FWD_ROWS
#define VQF_GLIMPSE_FWD_ROWS 4
#endif
// threads per sample of the pooling backward: 256 (four waves, eight 16-byte loads in flight per lane) streams the image grid at
// 6.3-6.5 TB/s, 512: 6.1-6.3, 1024 (round 4): 5.9-6.0 -- few resident waves stream best (tools/hbm_kernels_ab.py, gpurun_out/r05)
#ifndef VQF_GLIMPSE_BWD_THREADS
#define VQF_GLIMPSE_BWD_THREADS 256
#endif

namespace {

constexpr int MAXS = 1024;

// ---- logits[m,g] = hid[m,:] . w[g,:] + b[g]; one wave per row ------------------------------
// LIN: hid = relu(pre + b1) with pre LINEAR in the layer's input (the co-attention conv of the un-normalised fusion output,
// scaled per sample in the GEMM epilogue).  lin[m,g] = sum_j w[g,j] * (hid[m,j] - b1[j]) over the columns with hid > 0 is
// then the part of the logit that is linear in that input: the backward of F.normalize needs sum(Y * dY) per sample, which
// equals sum_g dlogits[m,g] * lin[m,g] and so costs no pass over the (N*L, 1000) tensors (vqf_l2_norm_bwd_coef_lin).
template <int G, bool LIN>
__global__ void att_logits_fwd_kernel(const float* __restrict__ hid, const float* __restrict__ w2,
                                      const float* __restrict__ b2, const float* __restrict__ b1, int M, int Hh,
                                      float* __restrict__ logits, float* __restrict__ lin) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const float* h = hid + (long long)row * Hh;
  float a[G], al[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { a[g] = 0.f; al[g] = 0.f; }
  if ((Hh & 3) == 0 && aligned16_dev(h) && aligned16_dev(w2) && (!LIN || aligned16_dev(b1))) {
    for (int c = lane * 4; c < Hh; c += 256) {
      const f32x4 x = vqf_ld_stream(reinterpret_cast<const f32x4*>(h + c));
      f32x4 xl = x;
      if (LIN) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) xl[j] = x[j] > 0.f ? x[j] - bb[j] : 0.f;
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(w2 + g * Hh + c);
        a[g] += x[0] * u[0] + x[1] * u[1] + x[2] * u[2] + x[3] * u[3];
        if (LIN) al[g] += xl[0] * u[0] + xl[1] * u[1] + xl[2] * u[2] + xl[3] * u[3];
      }
    }
  } else {
    for (int c = lane; c < Hh; c += 64)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        a[g] += h[c] * w2[g * Hh + c];
        if (LIN) al[g] += (h[c] > 0.f ? h[c] - b1[c] : 0.f) * w2[g * Hh + c];
      }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    a[g] = wave_sum(a[g]);
    if (LIN) al[g] = wave_sum(al[g]);
    if (lane == 0) {
      logits[(long long)G * row + g] = a[g] + b2[g];
      if (LIN) lin[(long long)G * row + g] = al[g];
    }
  }
}

// ---- backward of the G-logit head, optionally through the ReLU in front of it ----------------
// block = 256 threads, thread = 4 consecutive hidden columns (per 1024-column chunk); a block folds
// `lb` rows (att_bwd_rows_per_block: 128 for the image grid's 100 352 rows; fewer for short inputs -- the question head's 7 168
// rows as 56 blocks of 128 ran at 1 TB/s, 60 us; 448 blocks of 16: see DESIGN) and writes one partial slab row of width (G+1)*Hh + 4:
//   part[b][g*Hh + j]      sum_m dl[m,g] * hid[m,j]            (g < G)
//   part[b][G*Hh + j]      sum_m dhid_pre[m,j]
//   part[b][(G+1)*Hh + g]  sum_m dl[m,g]
constexpr int LB_ROWS = 128;
inline int att_bwd_rows_per_block(int M) {            // a multiple of the kernel's 4-row trips, ~512 blocks or more when M allows
  int lb = ((M / 512 + 3) / 4) * 4;
  return lb < 16 ? 16 : (lb > LB_ROWS ? LB_ROWS : lb);
}
#ifndef VQF_ATT_BWD_ST_NT
#define VQF_ATT_BWD_ST_NT 0    // 1: the hidden-layer gradient is stored non-temporal (A/B; measured 0.176-0.184 ms against 0.158-0.160 with plain stores)
#endif

// rowscale != nullptr: the STORED dhid_pre rows are multiplied by rowscale[row / rps] (the per-sample 1/norm of the layer's
// un-normalised input: its weight gradient and dgrad GEMMs then need no scaling); the bias partial sums stay unscaled.
// OBF16: the stored rows are bf16 (round-to-nearest-even), `dhid_pre` points to (M, Hh) bf16 storage -- the A operand of the
// layer's bf16 weight-gradient / input-gradient GEMMs (BASELINE config 3) without an fp32 round trip and a cast launch; the
// partial sums stay fp32 and are those of the fp32 values (Hh % 4 == 0, 8-byte aligned).
template <int G, bool RELU, bool OBF16 = false>
__global__ void att_logits_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ hid,
                                      const float* __restrict__ w2, int M, int Hh,
                                      float* __restrict__ dhid_pre, float* __restrict__ part,
                                      const float* __restrict__ rowscale, int rps, int lb) {
  const int r0 = blockIdx.x * lb, r1 = min(M, r0 + lb);
  const int pw = (G + 1) * Hh + 4;
  float* prow = part + (long long)blockIdx.x * pw;
  const bool vec = ((Hh & 3) == 0) && aligned16_dev(hid) && aligned16_dev(dhid_pre);
  for (int c = threadIdx.x * 4; c < Hh; c += 1024) {
    const int nc = min(4, Hh - c);
    float u[G][4], s[G][4], sb[4] = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u[g][j] = j < nc ? w2[g * Hh + c + j] : 0.f;
        s[g][j] = 0.f;
      }
    // RB rows per trip: their loads are issued together (one 16-byte load per row and thread is all a row needs, so a
    // thread that walks row by row keeps 1 KB per wave in flight: 3.4 TB/s; four rows at a time: see DESIGN)
    constexpr int RB = 4;
    for (int rb = r0; rb < r1; rb += RB) {
      float d[RB][G], x[RB][4], rs[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int r = min(rb + q, r1 - 1);               // the tail trip re-reads the last row (its results are not used)
#pragma unroll
        for (int g = 0; g < G; ++g) d[q][g] = dl[(long long)G * r + g];
        rs[q] = rowscale ? rowscale[r / rps] : 1.0f;
        const float* hp = hid + (long long)r * Hh + c;
        if (vec) {
          const f32x4 xv = vqf_ld_stream(reinterpret_cast<const f32x4*>(hp));
          x[q][0] = xv[0]; x[q][1] = xv[1]; x[q][2] = xv[2]; x[q][3] = xv[3];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) x[q][j] = j < nc ? hp[j] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int r = rb + q;
        if (r >= r1) break;
        float gp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = 0.f;
#pragma unroll
          for (int g = 0; g < G; ++g) { t += d[q][g] * u[g][j]; s[g][j] += d[q][g] * x[q][j]; }
          gp[j] = (!RELU || x[q][j] > 0.f) ? t : 0.f;
          sb[j] += gp[j];
        }
        if (OBF16) {
          __bf16 ob[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) ob[j] = (__bf16)(gp[j] * rs[q]);
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dhid_pre) + (long long)r * Hh + c) = *reinterpret_cast<const uint2*>(ob);
          continue;
        }
        float* o = dhid_pre + (long long)r * Hh + c;
        if (vec) {
          f32x4 gv = {gp[0] * rs[q], gp[1] * rs[q], gp[2] * rs[q], gp[3] * rs[q]};
          if (VQF_ATT_BWD_ST_NT) vqf_st_stream(reinterpret_cast<f32x4*>(o), gv); else *reinterpret_cast<f32x4*>(o) = gv;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (j < nc) o[j] = gp[j] * rs[q];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nc) {
#pragma unroll
        for (int g = 0; g < G; ++g) prow[g * Hh + c + j] = s[g][j];
        prow[G * Hh + c + j] = sb[j];
      }
  }
  if (threadIdx.x < G) {
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += dl[(long long)G * r + threadIdx.x];
    prow[(G + 1) * Hh + threadIdx.x] = a;
  }
}

__global__ void split_reduced_row_kernel(const float* __restrict__ red, int n0, int n1, int n2, float* __restrict__ d0,
                                         float* __restrict__ d1, float* __restrict__ d2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n0) d0[i] = red[i];
  else if (i < n0 + n1) { if (d1) d1[i - n0] = red[i]; }
  else if (i < n0 + n1 + n2) d2[i - n0 - n1] = red[i];
}

// feature element loaders: the pooled tensor is fp32 or (bf16 feature storage) bf16
template <typename FT> __device__ __forceinline__ f32x4 load4(const FT* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) {
  return vqf_ld_stream(reinterpret_cast<const f32x4*>(p));
}
template <> __device__ __forceinline__ f32x4 load4<__bf16>(const __bf16* p) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);        // 4 bf16 = 8 bytes
  f32x4 o;
  o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
  o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
  return o;
}

// ---- softmax over S + G glimpse sums ---------------------------------------------------------
// grid (ceil(C/1024), N); thread = 4 consecutive channels
template <int G, typename FT>
__global__ void glimpse_pool_fwd_kernel(const FT* __restrict__ feat,
                                        const float* __restrict__ logits, int N, int S, int C,
                                        int unit, float* __restrict__ wts,
                                        float* __restrict__ pooled) {
  __shared__ float w[G][MAXS];
  const int n = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave < G) {
    const int g = wave;
    const float* lg = logits + (long long)n * S * G + g;
    if (unit) {
      for (int s = lane; s < S; s += 64) w[g][s] = 1.0f;
    } else {
      float mx = -INFINITY;
      for (int s = lane; s < S; s += 64) mx = fmaxf(mx, lg[G * s]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int s = lane; s < S; s += 64) { const float e = expf(lg[G * s] - mx); w[g][s] = e; sum += e; }
      sum = wave_sum(sum);
      const float rs = 1.0f / sum;
      for (int s = lane; s < S; s += 64) w[g][s] *= rs;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && wts)
    for (int i = tid; i < G * S; i += blockDim.x)
      wts[(long long)n * G * S + i] = w[i / S][i % S];

  const int c = (blockIdx.x * blockDim.x + tid) * 4;
  if (c >= C) return;
  const FT* f = feat + (long long)n * S * C + c;
  const bool vec = ((C & 3) == 0) && aligned16_dev(feat) && aligned16_dev(pooled);
  f32x4 a[G];
#pragma unroll
  for (int g = 0; g < G; ++g) a[g] = f32x4{0, 0, 0, 0};
  if (vec) {
    int s = 0;
    // VQF_GLIMPSE_FWD_ROWS rows per trip, every load issued before the first is consumed (the sums keep the row order: same bits)
    for (; s + VQF_GLIMPSE_FWD_ROWS - 1 < S; s += VQF_GLIMPSE_FWD_ROWS) {
      f32x4 x[VQF_GLIMPSE_FWD_ROWS];
#pragma unroll
      for (int u = 0; u < VQF_GLIMPSE_FWD_ROWS; ++u) x[u] = load4<FT>(f + (long long)(s + u) * C);
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int u = 0; u < VQF_GLIMPSE_FWD_ROWS; ++u) a[g] += x[u] * w[g][s + u];
      }
    }
    for (; s < S; ++s) {
      const f32x4 x = load4<FT>(f + (long long)s * C);
#pragma unroll
      for (int g = 0; g < G; ++g) a[g] += x * w[g][s];
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
      *reinterpret_cast<f32x4*>(pooled + (long long)n * G * C + g * C + c) = a[g];
  } else {
    const int nc = min(4, C - c);
    for (int s = 0; s < S; ++s)
      for (int j = 0; j < nc; ++j) {
        const float x = (float)f[(long long)s * C + j];
#pragma unroll
        for (int g = 0; g < G; ++g) a[g][j] += x * w[g][s];
      }
    for (int j = 0; j < nc; ++j)
#pragma unroll
      for (int g = 0; g < G; ++g) pooled[(long long)n * G * C + g * C + c + j] = a[g][j];
  }
}

// block per sample; wave per position s (strided); then the softmax backward.
// dwts_extra (N,G,S) or null: gradient arriving through the returned attention weights.
template <int G, typename FT>
__global__ void glimpse_pool_bwd_kernel(const float* __restrict__ dpooled,
                                        const float* __restrict__ dwts_extra,
                                        const FT* __restrict__ feat,
                                        const float* __restrict__ wts, int N, int S, int C, int unit,
                                        float* __restrict__ dlogits, float* __restrict__ dfeat) {
  __shared__ float dw[G][MAXS];
  __shared__ float ws[G][MAXS];
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  for (int i = tid; i < G * S; i += blockDim.x) ws[i / S][i % S] = wts[(long long)n * G * S + i];
  __syncthreads();
  const float* dp = dpooled + (long long)n * G * C;
  const bool vec = ((C & 3) == 0) && aligned16_dev(feat) && aligned16_dev(dpooled) &&
                   (dfeat == nullptr || aligned16_dev(dfeat));
  if (vec && dfeat == nullptr && C <= 2048) {
    // the image grid is data (no dfeat): dpooled's G x C values of the sample stay in registers (8 x 16 bytes per glimpse and
    // lane), a row is eight independent 16-byte loads per lane issued together, and nothing else touches memory in the row loop
    constexpr int KC = 8;
    f32x4 pr[G][KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int c = lane * 4 + 256 * k;
#pragma unroll
      for (int g = 0; g < G; ++g) pr[g][k] = c < C ? *reinterpret_cast<const f32x4*>(dp + g * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int s = wave; s < S; s += nwave) {
      const FT* f = feat + ((long long)n * S + s) * C;
      f32x4 x[KC];
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int c = lane * 4 + 256 * k;
        x[k] = c < C ? load4<FT>(f + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      float a[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        a[g] = 0.f;
#pragma unroll
        for (int k = 0; k < KC; ++k) a[g] += x[k][0] * pr[g][k][0] + x[k][1] * pr[g][k][1] + x[k][2] * pr[g][k][2] + x[k][3] * pr[g][k][3];
        a[g] = wave_sum(a[g]);
        if (lane == 0) dw[g][s] = a[g] + (dwts_extra ? dwts_extra[((long long)n * G + g) * S + s] : 0.f);
      }
    }
  } else
  for (int s = wave; s < S; s += nwave) {
    const FT* f = feat + ((long long)n * S + s) * C;
    float* df = dfeat ? dfeat + ((long long)n * S + s) * C : nullptr;
    float a[G];
#pragma unroll
    for (int g = 0; g < G; ++g) a[g] = 0.f;
    if (vec) {
      for (int c = lane * 4; c < C; c += 256) {
        const f32x4 x = load4<FT>(f + c);
        f32x4 o = {0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const f32x4 p = *reinterpret_cast<const f32x4*>(dp + g * C + c);
          a[g] += x[0] * p[0] + x[1] * p[1] + x[2] * p[2] + x[3] * p[3];
          o += p * ws[g][s];
        }
        if (df) *reinterpret_cast<f32x4*>(df + c) = o;
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float x = (float)f[c];
        float o = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) { a[g] += x * dp[g * C + c]; o += ws[g][s] * dp[g * C + c]; }
        if (df) df[c] = o;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      a[g] = wave_sum(a[g]);
      if (lane == 0) dw[g][s] = a[g] + (dwts_extra ? dwts_extra[((long long)n * G + g) * S + s] : 0.f);
    }
  }
  __syncthreads();
  if (wave < G) {
    const int g = wave;
    float dot = 0.f;
    if (!unit) {
      for (int s = lane; s < S; s += 64) dot += ws[g][s] * dw[g][s];
      dot = wave_sum(dot);
    }
    for (int s = lane; s < S; s += 64) {
      // softmax over a singleton axis: y == 1 and dy - sum(dy*y) == 0 exactly
      const float d = unit ? 0.f : ws[g][s] * (dw[g][s] - dot);
      dlogits[((long long)n * S + s) * G + g] = d;
    }
  }
}

}  // namespace

extern "C" {

int vqf_att_logits_fwd(const float* hid, const float* w2, const float* b2, int M, int Hh, int G,
                       float* logits, void* stream) {
  if (!hid || !w2 || !b2 || !logits || M <= 0 || Hh <= 0) return VQF_E_BADARG;
  if (G != 1 && G != 2) return VQF_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const float* nob1 = nullptr;
  float* nolin = nullptr;
  if (G == 2)
    VQF_LAUNCH(KID_ATT_LOGITS_FWD, (att_logits_fwd_kernel<2, false>), dim3((M + 3) / 4), dim3(256), 0, s, hid,
               w2, b2, nob1, M, Hh, logits, nolin);
  else
    VQF_LAUNCH(KID_ATT_LOGITS_FWD, (att_logits_fwd_kernel<1, false>), dim3((M + 3) / 4), dim3(256), 0, s, hid,
               w2, b2, nob1, M, Hh, logits, nolin);
  return vqf_last_error();
}

int vqf_att_logits_fwd_lin(const float* hid, const float* w2, const float* b2, const float* b1, int M, int Hh, int G,
                           float* logits, float* lin, void* stream) {
  if (!hid || !w2 || !b2 || !b1 || !logits || !lin || M <= 0 || Hh <= 0) return VQF_E_BADARG;
  if (G != 1 && G != 2) return VQF_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (G == 2)
    VQF_LAUNCH(KID_ATT_LOGITS_FWD, (att_logits_fwd_kernel<2, true>), dim3((M + 3) / 4), dim3(256), 0, s, hid,
               w2, b2, b1, M, Hh, logits, lin);
  else
    VQF_LAUNCH(KID_ATT_LOGITS_FWD, (att_logits_fwd_kernel<1, true>), dim3((M + 3) / 4), dim3(256), 0, s, hid,
               w2, b2, b1, M, Hh, logits, lin);
  return vqf_last_error();
}

size_t vqf_att_logits_bwd_ws_bytes(int M, int Hh) {
  if (M <= 0 || Hh <= 0) return 0;
  const int lb = att_bwd_rows_per_block(M);
  return (size_t)((M + lb - 1) / lb + 1 + VQF_REDUCE_SPLITS) * (size_t)(3 * Hh + 4) * sizeof(float);
}

int vqf_att_logits_bwd(const float* dlogits, const float* hid, const float* w2, int M, int Hh, int G,
                       int relu_mask, float* dhid_pre, float* dw2, float* db2, float* dbias1,
                       void* ws, size_t ws_bytes, void* stream) {
  return vqf_att_logits_bwd_rowscale(dlogits, hid, w2, nullptr, 1, M, Hh, G, relu_mask, dhid_pre, dw2, db2, dbias1, ws,
                                     ws_bytes, stream);
}

static int att_logits_bwd_impl(const float* dlogits, const float* hid, const float* w2, const float* rowscale,
                               int rows_per_scale, int M, int Hh, int G, int relu_mask, float* dhid_pre, int out_bf16, float* dw2,
                               float* db2, float* dbias1, void* ws, size_t ws_bytes, void* stream);

int vqf_att_logits_bwd_rowscale(const float* dlogits, const float* hid, const float* w2, const float* rowscale,
                                int rows_per_scale, int M, int Hh, int G, int relu_mask, float* dhid_pre, float* dw2,
                                float* db2, float* dbias1, void* ws, size_t ws_bytes, void* stream) {
  return att_logits_bwd_impl(dlogits, hid, w2, rowscale, rows_per_scale, M, Hh, G, relu_mask, dhid_pre, 0, dw2, db2, dbias1, ws,
                             ws_bytes, stream);
}

int vqf_att_logits_bwd_rowscale_obf16(const float* dlogits, const float* hid, const float* w2, const float* rowscale,
                                      int rows_per_scale, int M, int Hh, int G, void* dhid_pre_bf16, float* dw2, float* db2,
                                      float* dbias1, void* ws, size_t ws_bytes, void* stream) {
  if (G != 2 || (Hh % 4)) return VQF_E_UNSUPPORTED;
  if (((uintptr_t)dhid_pre_bf16) & 7) return VQF_E_ALIGN;
  return att_logits_bwd_impl(dlogits, hid, w2, rowscale, rows_per_scale, M, Hh, G, 1, (float*)dhid_pre_bf16, 1, dw2, db2, dbias1,
                             ws, ws_bytes, stream);
}

static int att_logits_bwd_impl(const float* dlogits, const float* hid, const float* w2, const float* rowscale,
                               int rows_per_scale, int M, int Hh, int G, int relu_mask, float* dhid_pre, int out_bf16, float* dw2,
                               float* db2, float* dbias1, void* ws, size_t ws_bytes, void* stream) {
  if (!dlogits || !hid || !w2 || !dhid_pre || !dw2 || !db2 || M <= 0 || Hh <= 0 || rows_per_scale <= 0)
    return VQF_E_BADARG;
  if (G != 1 && G != 2) return VQF_E_UNSUPPORTED;
  if (!ws || ws_bytes < vqf_att_logits_bwd_ws_bytes(M, Hh)) return VQF_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int lb = att_bwd_rows_per_block(M);
  const int nb = (M + lb - 1) / lb;
  const int pw = (G + 1) * Hh + 4;
  float* part = (float*)ws;
  float* red = part + (size_t)nb * pw;     // reduced row [pw]
#define VQF_LB(G_, R_)                                                                           \
  VQF_LAUNCH(KID_ATT_LOGITS_BWD, (att_logits_bwd_kernel<G_, R_>), dim3(nb), dim3(256), 0, s,     \
             dlogits, hid, w2, M, Hh, dhid_pre, part, rowscale, rows_per_scale, lb)
  if (out_bf16)    VQF_LAUNCH(KID_ATT_LOGITS_BWD, (att_logits_bwd_kernel<2, true, true>), dim3(nb), dim3(256), 0, s, dlogits, hid, w2, M,
                              Hh, dhid_pre, part, rowscale, rows_per_scale, lb);
  else if (G == 2) { if (relu_mask) VQF_LB(2, true); else VQF_LB(2, false); }
  else             { if (relu_mask) VQF_LB(1, true); else VQF_LB(1, false); }
#undef VQF_LB
  int rc = vqf_last_error();
  if (rc) return rc;
  rc = vqf_colreduce_2stage(part, nb, pw, red, red + pw, s);
  if (rc) return rc;
  // the reduced row [dw2 (G*Hh) | dbias1 (Hh) | db2 (G)] -> its three destinations, one launch (was three device copies)
  hipLaunchKernelGGL(split_reduced_row_kernel, dim3(((G + 1) * Hh + G + 255) / 256), dim3(256), 0, s, red, G * Hh, Hh, G, dw2,
                     dbias1, db2);
  return vqf_last_error();
}

}  // extern "C"

namespace {
// The same for a NARROW feature tensor (C / 4 <= 256 threads wide: HieCoAtten's C = 512 at batch 256): the kernel above gives
// such a sample one workgroup of which half the threads have a channel group and walk all S rows alone -- two active waves per
// CU, 8 KB in flight: 3.2 TB/s.  Here 1024 threads = RS row slots x C / 4 channel groups; slot rs sums rows rs, rs + RS, ... (two
// rows per trip in flight), the slots are folded through LDS in slot order: deterministic, a different association than the
// single chain of the wide kernel (same value to fp32 rounding).
template <int G, typename FT>
__global__ void __launch_bounds__(1024) glimpse_pool_fwd_rows_kernel(const FT* __restrict__ feat, const float* __restrict__ logits,
                                                                     int N, int S, int C, int unit, float* __restrict__ wts,
                                                                     float* __restrict__ pooled) {
  extern __shared__ float smem_g[];
  float* w = smem_g;                                   // [G][S]
  float* red = smem_g + G * S;                         // [RS][G][C]
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave < G) {
    const int g = wave;
    const float* lg = logits + (long long)n * S * G + g;
    if (unit) {
      for (int s = lane; s < S; s += 64) w[g * S + s] = 1.0f;
    } else {
      float mx = -INFINITY;
      for (int s = lane; s < S; s += 64) mx = fmaxf(mx, lg[G * s]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int s = lane; s < S; s += 64) { const float e = expf(lg[G * s] - mx); w[g * S + s] = e; sum += e; }
      sum = wave_sum(sum);
      const float rs = 1.0f / sum;
      for (int s = lane; s < S; s += 64) w[g * S + s] *= rs;
    }
  }
  __syncthreads();
  if (wts)
    for (int i = tid; i < G * S; i += blockDim.x) wts[(long long)n * G * S + i] = w[i];
  const int CT = C >> 2, RS = blockDim.x / CT;
  const int c4 = tid % CT, rs = tid / CT;
  const FT* f = feat + (long long)n * S * C + 4 * c4;
  f32x4 a[G];
#pragma unroll
  for (int g = 0; g < G; ++g) a[g] = f32x4{0, 0, 0, 0};
  int s = rs;
  for (; s + RS < S; s += 2 * RS) {
    const f32x4 x0 = load4<FT>(f + (long long)s * C), x1 = load4<FT>(f + (long long)(s + RS) * C);
#pragma unroll
    for (int g = 0; g < G; ++g) { a[g] += x0 * w[g * S + s]; a[g] += x1 * w[g * S + s + RS]; }
  }
  if (s < S) {
    const f32x4 x0 = load4<FT>(f + (long long)s * C);
#pragma unroll
    for (int g = 0; g < G; ++g) a[g] += x0 * w[g * S + s];
  }
#pragma unroll
  for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(red + ((rs * G + g) * C) + 4 * c4) = a[g];
  __syncthreads();
  for (int g = rs; g < G; g += RS) {                  // row slot g folds glimpse g (RS >= G)
    f32x4 sum = *reinterpret_cast<const f32x4*>(red + (g * C) + 4 * c4);
    for (int q = 1; q < RS; ++q) sum += *reinterpret_cast<const f32x4*>(red + ((q * G + g) * C) + 4 * c4);
    *reinterpret_cast<f32x4*>(pooled + (long long)n * G * C + g * C + 4 * c4) = sum;
  }
}

template <typename FT>
int glimpse_fwd_launch(const FT* feat, const float* logits, int N, int S, int C, int G, int unit_softmax,
                       float* wts, float* pooled, void* stream) {
  if (!feat || !logits || !pooled || N <= 0 || S <= 0 || C <= 0) return VQF_E_BADARG;
  if (S > MAXS || (G != 1 && G != 2) || N > 65535) return VQF_E_UNSUPPORTED;
  dim3 grid((C + 1023) / 1024, N);
  hipStream_t s = (hipStream_t)stream;
  {
    // narrow tensors with enough rows: row slots instead of idle threads (glimpse_pool_fwd_rows_kernel)
    const int CT = C / 4;
    const bool pow2 = CT > 0 && (CT & (CT - 1)) == 0;
    if ((C % 4) == 0 && pow2 && CT <= 256 && CT >= 16 && S >= 4 * (1024 / CT) && aligned16(feat) && aligned16(pooled)) {
      const int RS = 1024 / CT;
      const size_t lds = sizeof(float) * ((size_t)G * S + (size_t)RS * G * C);
      if (RS >= G && lds <= 64 * 1024) {
        if (G == 2)
          VQF_LAUNCH(KID_GLIMPSE_FWD, (glimpse_pool_fwd_rows_kernel<2, FT>), dim3(N), dim3(1024), lds, s, feat, logits, N, S, C,
                     unit_softmax, wts, pooled);
        else
          VQF_LAUNCH(KID_GLIMPSE_FWD, (glimpse_pool_fwd_rows_kernel<1, FT>), dim3(N), dim3(1024), lds, s, feat, logits, N, S, C,
                     unit_softmax, wts, pooled);
        return vqf_last_error();
      }
    }
  }
  if (G == 2)
    VQF_LAUNCH(KID_GLIMPSE_FWD, (glimpse_pool_fwd_kernel<2, FT>), grid, dim3(256), 0, s, feat, logits, N, S,
               C, unit_softmax, wts, pooled);
  else
    VQF_LAUNCH(KID_GLIMPSE_FWD, (glimpse_pool_fwd_kernel<1, FT>), grid, dim3(256), 0, s, feat, logits, N, S,
               C, unit_softmax, wts, pooled);
  return vqf_last_error();
}

template <typename FT>
int glimpse_bwd_launch(const float* dpooled, const float* dwts_extra, const FT* feat, const float* wts, int N,
                       int S, int C, int G, int unit_softmax, float* dlogits, float* dfeat, void* stream) {
  if (!dpooled || !feat || !wts || !dlogits || N <= 0 || S <= 0 || C <= 0) return VQF_E_BADARG;
  if (S > MAXS || (G != 1 && G != 2)) return VQF_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  // one workgroup per sample.  A wave keeps ceil(C / 256) 16-byte loads per lane in flight (a row of the grid); about 64 such
  // loads per lane and CU stream best: 8 waves per CU at C = 2048 (the headline: 256 threads per sample, two samples per CU; few
  // resident waves, see VQF_GLIMPSE_BWD_THREADS), 32 at C = 512 (HieCoAtten, one sample per CU: 1024 threads -- 256 left a CU four
  // waves with two loads each, 30 -> 64 us)
  int waves = VQF_GLIMPSE_BWD_THREADS / 64;
  {
    const int cus = vqf_cu_count() > 0 ? vqf_cu_count() : 256;
    const int per_row = (C + 255) / 256 > 8 ? 8 : (C + 255) / 256;
    int target = 64 / per_row;
    target = target < 8 ? 8 : (target > 32 ? 32 : target);
    const int want = (target * cus + N - 1) / N;
    if (want > waves) waves = want > 16 ? 16 : want;
  }
  const dim3 block(64 * waves);
  if (G == 2)
    VQF_LAUNCH(KID_GLIMPSE_BWD, (glimpse_pool_bwd_kernel<2, FT>), dim3(N), block, 0, s, dpooled,
               dwts_extra, feat, wts, N, S, C, unit_softmax, dlogits, dfeat);
  else
    VQF_LAUNCH(KID_GLIMPSE_BWD, (glimpse_pool_bwd_kernel<1, FT>), dim3(N), block, 0, s, dpooled,
               dwts_extra, feat, wts, N, S, C, unit_softmax, dlogits, dfeat);
  return vqf_last_error();
}
}  // namespace

extern "C" {

int vqf_glimpse_pool_fwd(const float* feat, const float* logits, int N, int S, int C, int G,
                         int unit_softmax, float* wts, float* pooled, void* stream) {
  return glimpse_fwd_launch<float>(feat, logits, N, S, C, G, unit_softmax, wts, pooled, stream);
}

int vqf_glimpse_pool_fwd_bf16(const void* feat, const float* logits, int N, int S, int C, int G,
                              int unit_softmax, float* wts, float* pooled, void* stream) {
  return glimpse_fwd_launch<__bf16>((const __bf16*)feat, logits, N, S, C, G, unit_softmax, wts, pooled, stream);
}

int vqf_glimpse_pool_bwd(const float* dpooled, const float* dwts_extra, const float* feat,
                         const float* wts, int N, int S, int C, int G, int unit_softmax,
                         float* dlogits, float* dfeat, void* stream) {
  return glimpse_bwd_launch<float>(dpooled, dwts_extra, feat, wts, N, S, C, G, unit_softmax, dlogits, dfeat,
                                   stream);
}

int vqf_glimpse_pool_bwd_bf16(const float* dpooled, const float* dwts_extra, const void* feat,
                              const float* wts, int N, int S, int C, int G, int unit_softmax,
                              float* dlogits, void* stream) {
  return glimpse_bwd_launch<__bf16>(dpooled, dwts_extra, (const __bf16*)feat, wts, N, S, C, G, unit_softmax,
                                    dlogits, nullptr, stream);
}

}  // extern "C"
