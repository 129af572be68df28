// Element-wise stages of HieCoAtten / AttentionNet (HBM-bound, 16-byte accesses):
//   dropout                      hieCoAtten.py:26,28   networks.py:22,24,55,57   (F.dropout, p=0.5)
//   y = dropout(tanh(a [+ b]))   hieCoAtten.py:32-33,38-39,45-46
//   row softmax over the last axis   modules.py:91-92 (Attention_2)
// Dropout masks are Philox4x32-10(seed, element index / 4), regenerated in the backward, or an
// explicit uint8 keep-mask (parity tests).
#include "common.h"

namespace {

__device__ __forceinline__ void keep4(const uint8_t* __restrict__ keep, uint64_t seed, uint32_t thr,
                                      float inv_keep, long long i4, float (&sc)[4]) {
  if (keep) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(keep + 4 * i4);
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[j] = ((w >> (8 * j)) & 0xFFu) ? inv_keep : 0.f;
  } else if (thr != 0u) {
    const uint4 r = philox4x32_10((uint64_t)i4, seed);
    sc[0] = r.x >= thr ? inv_keep : 0.f;
    sc[1] = r.y >= thr ? inv_keep : 0.f;
    sc[2] = r.z >= thr ? inv_keep : 0.f;
    sc[3] = r.w >= thr ? inv_keep : 0.f;
  } else {
    sc[0] = sc[1] = sc[2] = sc[3] = 1.0f;
  }
}

// MODE 0: y = x * sc                      (dropout forward; also its backward with x = dy)
// MODE 1: y = tanh(a + b) * sc            (b may be null)
// MODE 2: dx = dy * sc * (1 - t^2), t = y / sc   (backward of MODE 1 given its output y)
template <int MODE>
__global__ void ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                          const uint8_t* __restrict__ keep, uint64_t seed, uint32_t thr,
                          float inv_keep, long long n4, float* __restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float sc[4];
    keep4(keep, seed, thr, inv_keep, i, sc);
    f32x4 x = *reinterpret_cast<const f32x4*>(a + 4 * i);
    f32x4 y;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = x[j] * sc[j];
    } else if (MODE == 1) {
      if (b) x += *reinterpret_cast<const f32x4*>(b + 4 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = vqf_tanh_fast(x[j]) * sc[j];
    } else {
      const f32x4 yy = *reinterpret_cast<const f32x4*>(b + 4 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = sc[j] > 0.f ? yy[j] * (1.0f / inv_keep) : 0.f;
        y[j] = x[j] * sc[j] * (1.0f - t * t);
      }
    }
    *reinterpret_cast<f32x4*>(out + 4 * i) = y;
  }
}

// The same three modes over 2-D operands with ROW STRIDES (column blocks of wider buffers: HieCoAtten's [Cv | img_] product
// of the concatenated fc_Wbv / fc_Wv weights and its gradient buffer).  The dropout index of element (r, c) is r * W + c, i.e.
// that of the contiguous logical (R, W) tensor: a strided call gives the bits of the flat one.
template <int MODE>
__global__ void ew2d_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                            const uint8_t* __restrict__ keep, uint64_t seed, uint32_t thr, float inv_keep, int R, int W4,
                            float* __restrict__ out, int ldo) {
  const unsigned n4 = (unsigned)R * (unsigned)W4;
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const unsigned r = i / (unsigned)W4, c4 = i - r * (unsigned)W4;
    float sc[4];
    keep4(keep, seed, thr, inv_keep, (long long)i, sc);
    f32x4 x = *reinterpret_cast<const f32x4*>(a + (long long)r * lda + 4 * c4);
    f32x4 y;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = x[j] * sc[j];
    } else if (MODE == 1) {
      if (b) x += *reinterpret_cast<const f32x4*>(b + (long long)r * ldb + 4 * c4);
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = vqf_tanh_fast(x[j]) * sc[j];
    } else {
      const f32x4 yy = *reinterpret_cast<const f32x4*>(b + (long long)r * ldb + 4 * c4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = sc[j] > 0.f ? yy[j] * (1.0f / inv_keep) : 0.f;
        y[j] = x[j] * sc[j] * (1.0f - t * t);
      }
    }
    *reinterpret_cast<f32x4*>(out + (long long)r * ldo + 4 * c4) = y;
  }
}

// y[b, t, :] = x[b, t, :] * keep / (1 - p) over a (B, T, H) tensor whose first two axes carry free element strides on either
// side (sb_*, st_*; the last axis contiguous): the LSTM-output dropout (mfb.py:70, mhb_coAtt.py:75) together with the
// (T, B, H) -> (B, T, H) re-layout between the recursion's time-major states and the attention head's sample-major rows, and
// its backward (the same call with the strides swapped).  The dropout index of (b, t, h) is (b * T + t) * H + h.
__global__ void dropout_bt_kernel(const float* __restrict__ x, long long sb_in, long long st_in, const uint8_t* __restrict__ keep,
                                  uint64_t seed, uint32_t thr, float inv_keep, int B, int T, int H4, float* __restrict__ y,
                                  long long sb_out, long long st_out) {
  const unsigned rows = (unsigned)B * (unsigned)T, n4 = rows * (unsigned)H4;
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const unsigned r = i / (unsigned)H4, c4 = i - r * (unsigned)H4;
    const unsigned b = r / (unsigned)T, t = r - b * (unsigned)T;
    float sc[4];
    keep4(keep, seed, thr, inv_keep, (long long)i, sc);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + b * sb_in + t * st_in + 4 * c4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] * sc[j];
    *reinterpret_cast<f32x4*>(y + b * sb_out + t * st_out + 4 * c4) = o;
  }
}

// The gate of modules.py:103-109 (Nonlinear_layer): y = tanh(a) * sigmoid(b), and its backward from the inputs:
// da = dy * sigmoid(b) * (1 - tanh(a)^2), db = dy * tanh(a) * sigmoid(b) * (1 - sigmoid(b)).  tanhf / expf of libm accuracy class
// (the modules of networks.py chain six such layers).
template <bool BWD>
__global__ void gate_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ dy, long long n4,
                            float* __restrict__ o1, float* __restrict__ o2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(a + 4 * i), z = *reinterpret_cast<const f32x4*>(b + 4 * i);
    f32x4 r1, r2;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (BWD) g = *reinterpret_cast<const f32x4*>(dy + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = tanhf(x[j]), sg = 1.0f / (1.0f + expf(-z[j]));
      if (BWD) {
        r1[j] = g[j] * sg * (1.0f - t * t);
        r2[j] = g[j] * t * sg * (1.0f - sg);
      } else {
        r1[j] = t * sg;
      }
    }
    *reinterpret_cast<f32x4*>(o1 + 4 * i) = r1;
    if (BWD) *reinterpret_cast<f32x4*>(o2 + 4 * i) = r2;
  }
}

// one wave per row
__global__ void softmax_rows_fwd_kernel(const float* __restrict__ x, int R, int W, float* __restrict__ y) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const float* p = x + (long long)row * W;
  float* q = y + (long long)row * W;
  float mx = -INFINITY;
  for (int c = lane; c < W; c += 64) mx = fmaxf(mx, p[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < W; c += 64) { const float e = expf(p[c] - mx); q[c] = e; sum += e; }
  sum = wave_sum(sum);
  const float rs = 1.0f / sum;
  for (int c = lane; c < W; c += 64) q[c] *= rs;
}

__global__ void softmax_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int R,
                                        int W, float* __restrict__ dx) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const float* g = dy + (long long)row * W;
  const float* p = y + (long long)row * W;
  float dot = 0.f;
  for (int c = lane; c < W; c += 64) dot += g[c] * p[c];
  dot = wave_sum(dot);
  for (int c = lane; c < W; c += 64) dx[(long long)row * W + c] = p[c] * (g[c] - dot);
}

// log_softmax over the last axis, one wave per row: y = (x - max) - log(sum exp(x - max))   (torch's formulation)
__global__ void log_softmax_rows_fwd_kernel(const float* __restrict__ x, int R, int W, float* __restrict__ y) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const float* p = x + (long long)row * W;
  float* q = y + (long long)row * W;
  float mx = -INFINITY;
  for (int c = lane; c < W; c += 64) mx = fmaxf(mx, p[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < W; c += 64) sum += expf(p[c] - mx);
  sum = wave_sum(sum);
  const float ls = logf(sum);
  for (int c = lane; c < W; c += 64) q[c] = (p[c] - mx) - ls;
}

// dx = dy - exp(y) * sum_c dy
__global__ void log_softmax_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int R,
                                            int W, float* __restrict__ dx) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const float* g = dy + (long long)row * W;
  const float* p = y + (long long)row * W;
  float tot = 0.f;
  for (int c = lane; c < W; c += 64) tot += g[c];
  tot = wave_sum(tot);
  for (int c = lane; c < W; c += 64) dx[(long long)row * W + c] = g[c] - expf(p[c]) * tot;
}

int ew_args_ok(const void* a, const void* out, const uint8_t* keep, float p, long long n) {
  if (!a || !out || n <= 0) return VQF_E_BADARG;
  if (n % 4) return VQF_E_UNSUPPORTED;
  if (p < 0.f || p >= 1.f) return VQF_E_BADARG;
  if (!aligned16(a) || !aligned16(out) || (keep && (((uintptr_t)keep) & 3))) return VQF_E_ALIGN;
  return VQF_OK;
}

template <int MODE>
int ew_launch(int kid, const float* a, const float* b, const uint8_t* keep, uint64_t seed, float p,
              long long n, float* out, void* stream) {
  const uint32_t thr = (keep || p == 0.f) ? 0u : drop_threshold_host(p);
  const float inv_keep = (keep || p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
  const long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  VQF_LAUNCH(kid, ew_kernel<MODE>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, keep,
             seed, thr, inv_keep, n4, out);
  return vqf_last_error();
}

template <int MODE>
int ew2d_launch(int kid, const float* a, int lda, const float* b, int ldb, const uint8_t* keep, uint64_t seed, float p, int R,
                int W, float* out, int ldo, void* stream) {
  if (!a || !out || R <= 0 || W <= 0 || p < 0.f || p >= 1.f) return VQF_E_BADARG;
  if ((W % 4) || (lda % 4) || (ldo % 4) || (b && (ldb % 4)) || (long long)R * (W / 4) >= (1LL << 31)) return VQF_E_UNSUPPORTED;
  if (lda < W || ldo < W || (b && ldb < W)) return VQF_E_BADARG;
  if (!aligned16(a) || !aligned16(out) || (b && !aligned16(b)) || (keep && (((uintptr_t)keep) & 3))) return VQF_E_ALIGN;
  const uint32_t thr = (keep || p == 0.f) ? 0u : drop_threshold_host(p);
  const float inv_keep = (keep || p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
  long long blocks = ((long long)R * (W / 4) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  VQF_LAUNCH(kid, ew2d_kernel<MODE>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, keep, seed, thr,
             inv_keep, R, W / 4, out, ldo);
  return vqf_last_error();
}

}  // namespace

extern "C" {

int vqf_dropout_bt(const float* x, long long sb_in, long long st_in, const uint8_t* keep, uint64_t seed, float p_drop, int B,
                   int T, int H, float* y, long long sb_out, long long st_out, void* stream) {
  if (!x || !y || B <= 0 || T <= 0 || H <= 0 || p_drop < 0.f || p_drop >= 1.f) return VQF_E_BADARG;
  if ((H % 4) || (sb_in % 4) || (st_in % 4) || (sb_out % 4) || (st_out % 4) || (long long)B * T * (H / 4) >= (1LL << 31))
    return VQF_E_UNSUPPORTED;
  if (!aligned16(x) || !aligned16(y) || (keep && (((uintptr_t)keep) & 3))) return VQF_E_ALIGN;
  const uint32_t thr = (keep || p_drop == 0.f) ? 0u : drop_threshold_host(p_drop);
  const float inv_keep = (keep || p_drop > 0.f) ? 1.0f / (1.0f - p_drop) : 1.0f;
  long long blocks = ((long long)B * T * (H / 4) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  VQF_LAUNCH(KID_DROPOUT, dropout_bt_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, sb_in, st_in, keep, seed,
             thr, inv_keep, B, T, H / 4, y, sb_out, st_out);
  return vqf_last_error();
}

int vqf_tanh_dropout_fwd2d(const float* a, int lda, const float* b, int ldb, const uint8_t* keep, uint64_t seed, float p_drop,
                           int R, int W, float* y, int ldy, void* stream) {
  return ew2d_launch<1>(KID_TANH_DROP_FWD, a, lda, b, ldb, keep, seed, p_drop, R, W, y, ldy, stream);
}

int vqf_tanh_dropout_bwd2d(const float* dy, int lddy, const float* y, int ldy, const uint8_t* keep, uint64_t seed, float p_drop,
                           int R, int W, float* dx, int lddx, void* stream) {
  if (!y) return VQF_E_BADARG;
  return ew2d_launch<2>(KID_TANH_DROP_BWD, dy, lddy, y, ldy, keep, seed, p_drop, R, W, dx, lddx, stream);
}

int vqf_dropout_f32(const float* x, const uint8_t* keep, uint64_t seed, float p_drop, long long n,
                    float* y, void* stream) {
  int rc = ew_args_ok(x, y, keep, p_drop, n);
  if (rc) return rc;
  return ew_launch<0>(KID_DROPOUT, x, nullptr, keep, seed, p_drop, n, y, stream);
}

int vqf_tanh_dropout_fwd(const float* a, const float* b, const uint8_t* keep, uint64_t seed,
                         float p_drop, long long n, float* y, void* stream) {
  int rc = ew_args_ok(a, y, keep, p_drop, n);
  if (rc) return rc;
  if (b && !aligned16(b)) return VQF_E_ALIGN;
  return ew_launch<1>(KID_TANH_DROP_FWD, a, b, keep, seed, p_drop, n, y, stream);
}

int vqf_tanh_dropout_bwd(const float* dy, const float* y, const uint8_t* keep, uint64_t seed,
                         float p_drop, long long n, float* dx, void* stream) {
  int rc = ew_args_ok(dy, dx, keep, p_drop, n);
  if (rc) return rc;
  if (!y || !aligned16(y)) return VQF_E_BADARG;
  return ew_launch<2>(KID_TANH_DROP_BWD, dy, y, keep, seed, p_drop, n, dx, stream);
}

int vqf_gate_tanh_sigmoid_fwd(const float* a, const float* b, long long n, float* y, void* stream) {
  if (!a || !b || !y || n <= 0) return VQF_E_BADARG;
  if (n % 4) return VQF_E_UNSUPPORTED;
  if (!aligned16(a) || !aligned16(b) || !aligned16(y)) return VQF_E_ALIGN;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  VQF_LAUNCH(KID_TANH_DROP_FWD, gate_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b,
             (const float*)nullptr, n / 4, y, (float*)nullptr);
  return vqf_last_error();
}

int vqf_gate_tanh_sigmoid_bwd(const float* dy, const float* a, const float* b, long long n, float* da, float* db, void* stream) {
  if (!dy || !a || !b || !da || !db || n <= 0) return VQF_E_BADARG;
  if (n % 4) return VQF_E_UNSUPPORTED;
  if (!aligned16(dy) || !aligned16(a) || !aligned16(b) || !aligned16(da) || !aligned16(db)) return VQF_E_ALIGN;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  VQF_LAUNCH(KID_TANH_DROP_BWD, gate_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, dy, n / 4, da,
             db);
  return vqf_last_error();
}

int vqf_softmax_rows_fwd(const float* x, int R, int W, float* y, void* stream) {
  if (!x || !y || R <= 0 || W <= 0) return VQF_E_BADARG;
  VQF_LAUNCH(KID_SOFTMAX_FWD, softmax_rows_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0,
             (hipStream_t)stream, x, R, W, y);
  return vqf_last_error();
}

int vqf_softmax_rows_bwd(const float* dy, const float* y, int R, int W, float* dx, void* stream) {
  if (!dy || !y || !dx || R <= 0 || W <= 0) return VQF_E_BADARG;
  VQF_LAUNCH(KID_SOFTMAX_BWD, softmax_rows_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0,
             (hipStream_t)stream, dy, y, R, W, dx);
  return vqf_last_error();
}

int vqf_log_softmax_rows_fwd(const float* x, int R, int W, float* y, void* stream) {
  if (!x || !y || R <= 0 || W <= 0) return VQF_E_BADARG;
  VQF_LAUNCH(KID_SOFTMAX_FWD, log_softmax_rows_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0,
             (hipStream_t)stream, x, R, W, y);
  return vqf_last_error();
}

int vqf_log_softmax_rows_bwd(const float* dy, const float* y, int R, int W, float* dx, void* stream) {
  if (!dy || !y || !dx || R <= 0 || W <= 0) return VQF_E_BADARG;
  VQF_LAUNCH(KID_SOFTMAX_BWD, log_softmax_rows_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0,
             (hipStream_t)stream, dy, y, R, W, dx);
  return vqf_last_error();
}

}  // extern "C"
