// Training-step tail of the solver: the two criteria of solver.py:25-28 with
// their gradients, and torch.optim.Adam (solver.py:29) as one multi-tensor launch.
//
//  * cross entropy (nn.CrossEntropyLoss(), mean over the non-ignored rows, ignore_index = -100):
//    one block per row computes lse, the row loss and dlogits = (softmax - onehot) / count in the
//    same pass; a single-block kernel adds the row losses in a fixed order.
//  * KL divergence (nn.KLDivLoss(), default reduction: mean over all N*A elements):
//    loss = mean(t * (log t - logp)) with 0 where t == 0; dlogp = -t / (N*A).
//  * Adam: m, v, p updated in place in the order of torch's single-tensor path; HBM-bound
//    (16 B read + 12 B written per element).
#include "common.h"
#include <math.h>

namespace {

constexpr int TR_THREADS = 256;

__device__ __forceinline__ float block_sum(float v, float* sh) {   // sh: >= 4 floats; result on all threads
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// one block per row n
__global__ __launch_bounds__(TR_THREADS) void ce_rows_kernel(const float* __restrict__ logits,
                                                             const long long* __restrict__ target, int N, int A,
                                                             float* __restrict__ rowloss,
                                                             float* __restrict__ dlogits) {
  __shared__ float sh[4];
  const int n = blockIdx.x;
  // rows that count (every block recomputes it: N int64 loads out of L2)
  float cnt = 0.f;
  for (int i = threadIdx.x; i < N; i += TR_THREADS) cnt += (target[i] != -100) ? 1.f : 0.f;
  cnt = block_sum(cnt, sh);
  const float inv_cnt = cnt > 0.f ? 1.f / cnt : 0.f;

  const float* x = logits + (size_t)n * A;
  float mx = -INFINITY;
  for (int a = threadIdx.x; a < A; a += TR_THREADS) mx = fmaxf(mx, x[a]);
  mx = block_max(mx, sh);
  float se = 0.f;
  for (int a = threadIdx.x; a < A; a += TR_THREADS) se += expf(x[a] - mx);
  se = block_sum(se, sh);
  const float lse = mx + logf(se);
  const long long t = target[n];
  const bool live = (t != -100);
  if (threadIdx.x == 0) rowloss[n] = (live && t >= 0 && t < A) ? (lse - x[t]) : (live ? NAN : 0.f);
  if (dlogits) {
    float* d = dlogits + (size_t)n * A;
    const float inv_se = 1.f / se;
    for (int a = threadIdx.x; a < A; a += TR_THREADS) {
      float p = expf(x[a] - mx) * inv_se;
      d[a] = live ? (p - ((long long)a == t ? 1.f : 0.f)) * inv_cnt : 0.f;
    }
  }
}

// loss = sum(part[0..P)) * scale  (scale < 0: divide by the number of targets != -100 instead)
__global__ __launch_bounds__(TR_THREADS) void loss_finish_kernel(const float* __restrict__ part, int P, float scale,
                                                                 const long long* __restrict__ target, int N,
                                                                 float* __restrict__ loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < P; i += TR_THREADS) s += part[i];
  s = block_sum(s, sh);
  if (target) {
    float cnt = 0.f;
    for (int i = threadIdx.x; i < N; i += TR_THREADS) cnt += (target[i] != -100) ? 1.f : 0.f;
    cnt = block_sum(cnt, sh);
    scale = 1.f / cnt;       // 0/0 -> NaN like torch when every row is ignored
  }
  if (threadIdx.x == 0) loss[0] = s * scale;
}

// element-wise part of KLDivLoss; each block reduces a contiguous chunk into part[blockIdx.x]
constexpr int KL_PER_BLOCK = TR_THREADS * 16;
__global__ __launch_bounds__(TR_THREADS) void kldiv_kernel(const float* __restrict__ logp,
                                                           const float* __restrict__ tgt, long long n, float inv_n,
                                                           float* __restrict__ part, float* __restrict__ dlogp) {
  __shared__ float sh[4];
  const long long base = (long long)blockIdx.x * KL_PER_BLOCK;
  float s = 0.f;
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    long long i = base + j * TR_THREADS + threadIdx.x;
    if (i < n) {
      float t = tgt[i], x = logp[i];
      s += t > 0.f ? t * (logf(t) - x) : (t == 0.f ? 0.f : NAN);
      if (dlogp) dlogp[i] = -t * inv_n;
    }
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ---- Adam -------------------------------------------------------------------
constexpr int ADAM_MAX_TENSORS = VQF_ADAM_MAX_TENSORS;
constexpr int ADAM_CHUNK = TR_THREADS * 16;      // elements per block
struct AdamTable {
  float* p[ADAM_MAX_TENSORS];
  const float* g[ADAM_MAX_TENSORS];
  float* m[ADAM_MAX_TENSORS];
  float* v[ADAM_MAX_TENSORS];
  long long n[ADAM_MAX_TENSORS];
  int block0[ADAM_MAX_TENSORS + 1];   // first block of tensor i
  int count;
};
struct AdamHyper { float beta1, beta2, one_m_beta1, one_m_beta2, eps, wd, step_size, bc2_sqrt; };

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamHyper& h) {
  if (h.wd != 0.f) g = g + h.wd * p;
  m = m + (g - m) * h.one_m_beta1;                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * h.beta2 + h.one_m_beta2 * g * g;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
  p = p - h.step_size * (m / denom);               // param.addcdiv_(exp_avg, denom, -step_size)
}

__global__ __launch_bounds__(TR_THREADS) void adam_kernel(const AdamTable tb, const AdamHyper h) {
  // block -> tensor (uniform scan of <= 32 entries held in kernarg SGPRs)
  int ti = 0;
  const int b = blockIdx.x;
  while (ti + 1 < tb.count && b >= tb.block0[ti + 1]) ++ti;
  float* __restrict__ p = tb.p[ti];
  const float* __restrict__ g = tb.g[ti];
  float* __restrict__ m = tb.m[ti];
  float* __restrict__ v = tb.v[ti];
  const long long n = tb.n[ti];
  const long long base = (long long)(b - tb.block0[ti]) * ADAM_CHUNK;
  const bool vec = aligned16_dev(p) && aligned16_dev(g) && aligned16_dev(m) && aligned16_dev(v);
  if (vec && base + ADAM_CHUNK <= n) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long long i = base + (long long)(j * TR_THREADS + threadIdx.x) * 4;
      f32x4 pp = *(const f32x4*)(p + i), gg = *(const f32x4*)(g + i);
      f32x4 mm = *(const f32x4*)(m + i), vv = *(const f32x4*)(v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pp[e], me = mm[e], ve = vv[e];
        adam_one(pe, gg[e], me, ve, h);
        pp[e] = pe; mm[e] = me; vv[e] = ve;
      }
      *(f32x4*)(p + i) = pp; *(f32x4*)(m + i) = mm; *(f32x4*)(v + i) = vv;
    }
  } else {
    for (int j = 0; j < 16; ++j) {
      long long i = base + j * TR_THREADS + threadIdx.x;
      if (i < n) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, g[i], mm, vv, h);
        p[i] = pp; m[i] = mm; v[i] = vv;
      }
    }
  }
}

}  // namespace

extern "C" {

size_t vqf_loss_ws_bytes(int N, int A) {
  long long n = (long long)N * A;
  long long parts = (n + KL_PER_BLOCK - 1) / KL_PER_BLOCK;
  if (parts < N) parts = N;
  return (size_t)(parts > 0 ? parts : 1) * sizeof(float);
}

int vqf_ce_loss(const float* logits, const long long* target, int N, int A, float* loss, float* dlogits,
                void* ws, size_t ws_bytes, void* stream) {
  if (N <= 0 || A <= 0) return VQF_E_BADARG;
  if (!logits || !target || !loss || !ws) return VQF_E_BADARG;
  if (ws_bytes < (size_t)N * sizeof(float)) return VQF_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* rowloss = (float*)ws;
  vqf_prof_dims(N, A, 0);
  VQF_LAUNCH(KID_CE_LOSS, ce_rows_kernel, dim3(N), dim3(TR_THREADS), 0, s, logits, target, N, A, rowloss, dlogits);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(TR_THREADS), 0, s, (const float*)rowloss, N, -1.f, target, N,
                     loss);
  return vqf_last_error();
}

int vqf_kldiv_loss(const float* logp, const float* target, int N, int A, float* loss, float* dlogp, void* ws,
                   size_t ws_bytes, void* stream) {
  if (N <= 0 || A <= 0) return VQF_E_BADARG;
  if (!logp || !target || !loss || !ws) return VQF_E_BADARG;
  const long long n = (long long)N * A;
  const long long parts = (n + KL_PER_BLOCK - 1) / KL_PER_BLOCK;
  if (parts > 0x7fffffffLL) return VQF_E_BADARG;
  if (ws_bytes < (size_t)parts * sizeof(float)) return VQF_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)ws;
  const float inv_n = (float)(1.0 / (double)n);
  vqf_prof_dims(N, A, 0);
  VQF_LAUNCH(KID_KLDIV_LOSS, kldiv_kernel, dim3((unsigned)parts), dim3(TR_THREADS), 0, s, logp, target, n, inv_n, part,
             dlogp);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(TR_THREADS), 0, s, (const float*)part, (int)parts, inv_n,
                     (const long long*)nullptr, 0, loss);
  return vqf_last_error();
}

int vqf_adam_step(const VqfAdamTensor* tensors, int count, double lr, double beta1, double beta2, double eps,
                  double weight_decay, long long step, void* stream) {
  if (count < 0 || step < 1) return VQF_E_BADARG;
  if (count && !tensors) return VQF_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  AdamHyper h;
  // hyper-parameters are combined in double and rounded once, as torch does with its Python scalars
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  h.beta1 = (float)beta1; h.beta2 = (float)beta2;
  h.one_m_beta1 = (float)(1.0 - beta1); h.one_m_beta2 = (float)(1.0 - beta2);
  h.eps = (float)eps; h.wd = (float)weight_decay;
  h.step_size = (float)(lr / bc1);
  h.bc2_sqrt = (float)sqrt(bc2);
  int i = 0;
  while (i < count) {
    AdamTable tb;
    tb.count = 0;
    long long blocks = 0;
    while (i < count && tb.count < ADAM_MAX_TENSORS) {
      const VqfAdamTensor& t = tensors[i];
      if (t.n < 0) return VQF_E_BADARG;
      if (t.n == 0) { ++i; continue; }
      if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) return VQF_E_BADARG;
      long long nb = (t.n + ADAM_CHUNK - 1) / ADAM_CHUNK;
      if (blocks + nb > 0x7fffffffLL) break;
      const int k = tb.count++;
      tb.p[k] = t.param; tb.g[k] = t.grad; tb.m[k] = t.exp_avg; tb.v[k] = t.exp_avg_sq; tb.n[k] = t.n;
      tb.block0[k] = (int)blocks;
      blocks += nb;
      ++i;
    }
    if (tb.count == 0) {
      if (i < count) return VQF_E_UNSUPPORTED;   // a single tensor too large for one grid
      break;
    }
    tb.block0[tb.count] = (int)blocks;
    for (int k = tb.count; k < ADAM_MAX_TENSORS; ++k) {
      tb.p[k] = nullptr; tb.g[k] = nullptr; tb.m[k] = nullptr; tb.v[k] = nullptr; tb.n[k] = 0;
      tb.block0[k + 1] = (int)blocks;
    }
    vqf_prof_dims(tb.count, (int)(blocks > 0x7fffffff ? 0x7fffffff : blocks), 0);
    VQF_LAUNCH(KID_ADAM, adam_kernel, dim3((unsigned)blocks), dim3(TR_THREADS), 0, s, tb, h);
    int rc = vqf_last_error();
    if (rc) return rc;
  }
  return VQF_OK;
}

}  // extern "C"
