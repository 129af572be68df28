// Attention heads of the co-attention ladder (HBM-bound, wave64 reductions).
//
//   question side (S = T tokens):  mfb.py:81-89   / mhb_coAtt.py:83-91
//   image side    (S = 196 regions, C = 2048): mfb.py:114-123 / mhb_coAtt.py:113-121
//
// att_logits_*   : hidden (M,Hh) -> 2 logits per row, and its backward through
//                  the preceding ReLU (one pass over the hidden activations).
// glimpse_pool_* : softmax over the S positions of a sample (or the reference's
//                  singleton-axis softmax == 1, mfb.py:84,118) and the two
//                  glimpse-weighted sums over the (N,S,C) feature tensor.  The
//                  image tensor is streamed exactly once per pass with 16-byte
//                  coalesced loads along C; softmax rows (S <= 1024) live in LDS
//                  and are reduced with wavefront shuffles.
#include "common.h"

namespace {

constexpr int MAXS = 1024;

// ---- logits[m,g] = hid[m,:] . w2[g,:] + b2[g]; one wave per row ------------
__global__ void att_logits_fwd_kernel(const float* __restrict__ hid, const float* __restrict__ w2,
                                      const float* __restrict__ b2, int M, int Hh,
                                      float* __restrict__ logits) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const float* h = hid + (long long)row * Hh;
  float a0 = 0.f, a1 = 0.f;
  if ((Hh & 3) == 0 && aligned16_dev(h) && aligned16_dev(w2)) {
    for (int c = lane * 4; c < Hh; c += 256) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(h + c);
      const f32x4 u = *reinterpret_cast<const f32x4*>(w2 + c);
      const f32x4 v = *reinterpret_cast<const f32x4*>(w2 + Hh + c);
      a0 += x[0] * u[0] + x[1] * u[1] + x[2] * u[2] + x[3] * u[3];
      a1 += x[0] * v[0] + x[1] * v[1] + x[2] * v[2] + x[3] * v[3];
    }
  } else {
    for (int c = lane; c < Hh; c += 64) { a0 += h[c] * w2[c]; a1 += h[c] * w2[Hh + c]; }
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if (lane == 0) {
    logits[2 * (long long)row] = a0 + b2[0];
    logits[2 * (long long)row + 1] = a1 + b2[1];
  }
}

// ---- backward of the 2-logit head through the ReLU -------------------------
// block = 256 threads, thread = 4 consecutive hidden columns (per 1024-column
// chunk); a block folds LB_ROWS rows and writes one partial slab row:
//   part[b][0..Hh)      sum_m dl[m,0] * hid[m,j]
//   part[b][Hh..2Hh)    sum_m dl[m,1] * hid[m,j]
//   part[b][2Hh..3Hh)   sum_m dhid_pre[m,j]
//   part[b][3Hh..3Hh+2) sum_m dl[m,g]
constexpr int LB_ROWS = 128;

__global__ void att_logits_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ hid,
                                      const float* __restrict__ w2, int M, int Hh,
                                      float* __restrict__ dhid_pre, float* __restrict__ part) {
  const int r0 = blockIdx.x * LB_ROWS, r1 = min(M, r0 + LB_ROWS);
  const int pw = 3 * Hh + 4;
  float* prow = part + (long long)blockIdx.x * pw;
  const bool vec = ((Hh & 3) == 0) && aligned16_dev(hid) && aligned16_dev(dhid_pre);
  for (int c = threadIdx.x * 4; c < Hh; c += 1024) {
    const int nc = min(4, Hh - c);
    float u[4], v[4], s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u[j] = j < nc ? w2[c + j] : 0.f;
      v[j] = j < nc ? w2[Hh + c + j] : 0.f;
    }
    if (vec) {
      for (int r = r0; r < r1; ++r) {
        const float d0 = dl[2 * (long long)r], d1 = dl[2 * (long long)r + 1];
        const f32x4 x = *reinterpret_cast<const f32x4*>(hid + (long long)r * Hh + c);
        f32x4 gp;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          gp[j] = x[j] > 0.f ? (d0 * u[j] + d1 * v[j]) : 0.f;
          s0[j] += d0 * x[j];
          s1[j] += d1 * x[j];
          sb[j] += gp[j];
        }
        *reinterpret_cast<f32x4*>(dhid_pre + (long long)r * Hh + c) = gp;
      }
    } else {
      for (int r = r0; r < r1; ++r) {
        const float d0 = dl[2 * (long long)r], d1 = dl[2 * (long long)r + 1];
        const float* h = hid + (long long)r * Hh + c;
        float* o = dhid_pre + (long long)r * Hh + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < nc) {
            const float x = h[j];
            const float gpre = x > 0.f ? (d0 * u[j] + d1 * v[j]) : 0.f;
            o[j] = gpre;
            s0[j] += d0 * x;
            s1[j] += d1 * x;
            sb[j] += gpre;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nc) { prow[c + j] = s0[j]; prow[Hh + c + j] = s1[j]; prow[2 * Hh + c + j] = sb[j]; }
  }
  if (threadIdx.x < 2) {
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += dl[2 * (long long)r + threadIdx.x];
    prow[3 * Hh + threadIdx.x] = a;
  }
}

// ---- softmax over S + two glimpse sums -------------------------------------
// grid (ceil(C/1024), N); thread = 4 consecutive channels
__global__ void glimpse_pool_fwd_kernel(const float* __restrict__ feat,
                                        const float* __restrict__ logits, int N, int S, int C,
                                        int unit, float* __restrict__ wts,
                                        float* __restrict__ pooled) {
  __shared__ float w[2][MAXS];
  const int n = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave < 2) {
    const int g = wave;
    const float* lg = logits + (long long)n * S * 2 + g;
    if (unit) {
      for (int s = lane; s < S; s += 64) w[g][s] = 1.0f;
    } else {
      float mx = -INFINITY;
      for (int s = lane; s < S; s += 64) mx = fmaxf(mx, lg[2 * s]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int s = lane; s < S; s += 64) { const float e = expf(lg[2 * s] - mx); w[g][s] = e; sum += e; }
      sum = wave_sum(sum);
      const float rs = 1.0f / sum;
      for (int s = lane; s < S; s += 64) w[g][s] *= rs;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && wts)
    for (int i = tid; i < 2 * S; i += blockDim.x)
      wts[(long long)n * 2 * S + i] = w[i / S][i % S];

  const int c = (blockIdx.x * blockDim.x + tid) * 4;
  if (c >= C) return;
  const float* f = feat + (long long)n * S * C + c;
  const bool vec = ((C & 3) == 0) && aligned16_dev(feat);
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  if (vec) {
    int s = 0;
    for (; s + 3 < S; s += 4) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(f + (long long)s * C);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(f + (long long)(s + 1) * C);
      const f32x4 x2 = *reinterpret_cast<const f32x4*>(f + (long long)(s + 2) * C);
      const f32x4 x3 = *reinterpret_cast<const f32x4*>(f + (long long)(s + 3) * C);
      a0 += x0 * w[0][s]; a1 += x0 * w[1][s];
      a0 += x1 * w[0][s + 1]; a1 += x1 * w[1][s + 1];
      a0 += x2 * w[0][s + 2]; a1 += x2 * w[1][s + 2];
      a0 += x3 * w[0][s + 3]; a1 += x3 * w[1][s + 3];
    }
    for (; s < S; ++s) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(f + (long long)s * C);
      a0 += x * w[0][s]; a1 += x * w[1][s];
    }
    *reinterpret_cast<f32x4*>(pooled + (long long)n * 2 * C + c) = a0;
    *reinterpret_cast<f32x4*>(pooled + (long long)n * 2 * C + C + c) = a1;
  } else {
    const int nc = min(4, C - c);
    for (int s = 0; s < S; ++s)
      for (int j = 0; j < nc; ++j) {
        const float x = f[(long long)s * C + j];
        a0[j] += x * w[0][s]; a1[j] += x * w[1][s];
      }
    for (int j = 0; j < nc; ++j) {
      pooled[(long long)n * 2 * C + c + j] = a0[j];
      pooled[(long long)n * 2 * C + C + c + j] = a1[j];
    }
  }
}

// block per sample; wave per position s (strided); then the softmax backward
__global__ void glimpse_pool_bwd_kernel(const float* __restrict__ dpooled,
                                        const float* __restrict__ feat,
                                        const float* __restrict__ wts, int N, int S, int C, int unit,
                                        float* __restrict__ dlogits, float* __restrict__ dfeat) {
  __shared__ float dw[2][MAXS];
  __shared__ float ws[2][MAXS];
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  for (int i = tid; i < 2 * S; i += blockDim.x) ws[i / S][i % S] = wts[(long long)n * 2 * S + i];
  __syncthreads();
  const float* dp0 = dpooled + (long long)n * 2 * C;
  const float* dp1 = dp0 + C;
  const bool vec = ((C & 3) == 0) && aligned16_dev(feat) && aligned16_dev(dpooled) &&
                   (dfeat == nullptr || aligned16_dev(dfeat));
  for (int s = wave; s < S; s += nwave) {
    const float* f = feat + ((long long)n * S + s) * C;
    float* df = dfeat ? dfeat + ((long long)n * S + s) * C : nullptr;
    const float w0 = ws[0][s], w1 = ws[1][s];
    float a0 = 0.f, a1 = 0.f;
    if (vec) {
      for (int c = lane * 4; c < C; c += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(f + c);
        const f32x4 p = *reinterpret_cast<const f32x4*>(dp0 + c);
        const f32x4 q = *reinterpret_cast<const f32x4*>(dp1 + c);
        a0 += x[0] * p[0] + x[1] * p[1] + x[2] * p[2] + x[3] * p[3];
        a1 += x[0] * q[0] + x[1] * q[1] + x[2] * q[2] + x[3] * q[3];
        if (df) *reinterpret_cast<f32x4*>(df + c) = p * w0 + q * w1;
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float x = f[c];
        a0 += x * dp0[c]; a1 += x * dp1[c];
        if (df) df[c] = w0 * dp0[c] + w1 * dp1[c];
      }
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (lane == 0) { dw[0][s] = a0; dw[1][s] = a1; }
  }
  __syncthreads();
  if (wave < 2) {
    const int g = wave;
    float dot = 0.f;
    if (!unit) {
      for (int s = lane; s < S; s += 64) dot += ws[g][s] * dw[g][s];
      dot = wave_sum(dot);
    }
    for (int s = lane; s < S; s += 64) {
      // softmax over a singleton axis: y == 1 and dy - sum(dy*y) == 0 exactly
      const float d = unit ? 0.f : ws[g][s] * (dw[g][s] - dot);
      dlogits[((long long)n * S + s) * 2 + g] = d;
    }
  }
}

}  // namespace

extern "C" {

int vqf_att_logits_fwd(const float* hid, const float* w2, const float* b2, int M, int Hh,
                       float* logits, void* stream) {
  if (!hid || !w2 || !b2 || !logits || M <= 0 || Hh <= 0) return VQF_E_BADARG;
  VQF_LAUNCH(KID_ATT_LOGITS_FWD, att_logits_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0,
             (hipStream_t)stream, hid, w2, b2, M, Hh, logits);
  return vqf_last_error();
}

size_t vqf_att_logits_bwd_ws_bytes(int M, int Hh) {
  if (M <= 0 || Hh <= 0) return 0;
  return (size_t)((M + LB_ROWS - 1) / LB_ROWS) * (size_t)(3 * Hh + 4) * sizeof(float) +
         (size_t)(3 * Hh + 4) * sizeof(float);
}

int vqf_att_logits_bwd(const float* dlogits, const float* hid, const float* w2, int M, int Hh,
                       float* dhid_pre, float* dw2, float* db2, float* dbias1, void* ws,
                       size_t ws_bytes, void* stream) {
  if (!dlogits || !hid || !w2 || !dhid_pre || !dw2 || !db2 || M <= 0 || Hh <= 0)
    return VQF_E_BADARG;
  if (!ws || ws_bytes < vqf_att_logits_bwd_ws_bytes(M, Hh)) return VQF_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int nb = (M + LB_ROWS - 1) / LB_ROWS;
  const int pw = 3 * Hh + 4;
  float* part = (float*)ws;
  float* red = part + (size_t)nb * pw;     // reduced row [pw]
  VQF_LAUNCH(KID_ATT_LOGITS_BWD, att_logits_bwd_kernel, dim3(nb), dim3(256), 0, s, dlogits, hid,
             w2, M, Hh, dhid_pre, part);
  int rc = vqf_last_error();
  if (rc) return rc;
  rc = vqf_group_reduce_f32(part, 1, nb, pw, red, stream);
  if (rc) return rc;
  hipError_t e = hipMemcpyAsync(dw2, red, (size_t)2 * Hh * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return (int)e;
  if (dbias1) {
    e = hipMemcpyAsync(dbias1, red + 2 * Hh, (size_t)Hh * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
  }
  e = hipMemcpyAsync(db2, red + 3 * Hh, 2 * sizeof(float), hipMemcpyDeviceToDevice, s);
  return e == hipSuccess ? VQF_OK : (int)e;
}

int vqf_glimpse_pool_fwd(const float* feat, const float* logits, int N, int S, int C,
                         int unit_softmax, float* wts, float* pooled, void* stream) {
  if (!feat || !logits || !pooled || N <= 0 || S <= 0 || C <= 0) return VQF_E_BADARG;
  if (S > MAXS) return VQF_E_UNSUPPORTED;
  dim3 grid((C + 1023) / 1024, N);
  VQF_LAUNCH(KID_GLIMPSE_FWD, glimpse_pool_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream,
             feat, logits, N, S, C, unit_softmax, wts, pooled);
  return vqf_last_error();
}

int vqf_glimpse_pool_bwd(const float* dpooled, const float* feat, const float* wts, int N, int S,
                         int C, int unit_softmax, float* dlogits, float* dfeat, void* stream) {
  if (!dpooled || !feat || !wts || !dlogits || N <= 0 || S <= 0 || C <= 0) return VQF_E_BADARG;
  if (S > MAXS) return VQF_E_UNSUPPORTED;
  VQF_LAUNCH(KID_GLIMPSE_BWD, glimpse_pool_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream,
             dpooled, feat, wts, N, S, C, unit_softmax, dlogits, dfeat);
  return vqf_last_error();
}

}  // extern "C"
