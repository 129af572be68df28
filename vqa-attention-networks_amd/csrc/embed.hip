// Question-encoder front end: e = tanh(Embedding(q)) or e = Embedding(q), and the backward.
//
//   mfb.py:68 / mhb_coAtt.py:69   que_embedded = F.tanh(self.word_embedding(questions))
//   hieCoAtten.py:27, networks.py:23,56, mhb_coAtt.py:181   plain lookups (vqf_embed_fwd / _bwd: the same kernels without the tanh)
//
// forward : out[t, :] = tanh(W[ids[t], :])                       one wave per token row, 16-byte accesses when E % 4 == 0
// backward: dW[v, :] = sum_{t: ids[t] == v} dout[t, :] * (1 - out[t, :]^2)        for EVERY vocabulary row v (zeros where unused)
//   One workgroup per 1-16 consecutive vocabulary rows: its four waves scan 2048 token ids per round (512 each, loaded once,
//   one ballot pass per row) and list a row's matching tokens in LDS in increasing order; the workgroup then adds their rows
//   column-parallel in that order.  No atomics, no sort: the sum
//   order is the token order, the same on every run (torch's embedding backward sorts the indices and runs ~20 small kernels,
//   ~0.1 ms of the MFB train step; this is one launch of ~10 us).
// Ids outside [0, V) select no row: the forward writes zeros for them, the backward ignores them (the reference's ids come
// from its own vocabulary, utils.py:185,303-304; torch would raise).
#include "common.h"

namespace {

// Time-major form (swapN = N > 0): ids are stored (N, Tq) as the reference holds them, the rows of out / dout are ordered
// (Tq, N) -- what the batch-major LSTM consumes (mfb.py:68-69), so no transposing copy sits between the lookup and the recursion.
__device__ __forceinline__ int tok_of_row(int t, int T, int swapN) {
  return swapN > 0 ? (t % swapN) * (T / swapN) + t / swapN : t;
}

// keep scale of element e of the flat (T, E) tensor: the draw of the element-wise dropout kernels (elementwise.hip keep4: one Philox
// call per 4 consecutive elements, or an explicit uint8 keep-mask)
__device__ __forceinline__ float keep_scale(const uint8_t* __restrict__ keep, uint64_t seed, uint32_t thr, float inv_keep, long long e) {
  if (keep) return keep[e] ? inv_keep : 0.f;
  if (thr == 0u) return 1.0f;
  const uint4 r = philox4x32_10((uint64_t)(e >> 2), seed);
  const uint32_t w = (e & 3) == 0 ? r.x : (e & 3) == 1 ? r.y : (e & 3) == 2 ? r.z : r.w;
  return w >= thr ? inv_keep : 0.f;
}

// DROP: out = dropout(W[ids]) (hieCoAtten.py:27-28: the lookup and its always-on functional dropout in one pass)
template <bool TANH, bool DROP = false>
__global__ void embed_fwd_kernel(const float* __restrict__ W, const long long* __restrict__ ids, int T, int V, int E,
                                      float* __restrict__ out, int swapN, const uint8_t* __restrict__ keep = nullptr, uint64_t seed = 0,
                                      uint32_t thr = 0u, float inv_keep = 1.0f) {
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= T) return;
  const int lane = threadIdx.x & 63;
  const long long id = ids[tok_of_row(t, T, swapN)];
  const bool ok = id >= 0 && id < V;
  const float* w = W + (ok ? id : 0) * (long long)E;
  float* o = out + (long long)t * E;
  if ((E & 3) == 0 && aligned16_dev(W) && aligned16_dev(out)) {
    for (int c = lane * 4; c < E; c += 256) {
      f32x4 x = ok ? *reinterpret_cast<const f32x4*>(w + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 y = x;
      if (TANH) y = f32x4{tanhf(x[0]), tanhf(x[1]), tanhf(x[2]), tanhf(x[3])};
      if (DROP) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] *= keep_scale(keep, seed, thr, inv_keep, (long long)t * E + c + j);
      }
      *reinterpret_cast<f32x4*>(o + c) = y;
    }
  } else {
    for (int c = lane; c < E; c += 64) {
      float y = ok ? (TANH ? tanhf(w[c]) : w[c]) : 0.f;
      if (DROP) y *= keep_scale(keep, seed, thr, inv_keep, (long long)t * E + c);
      o[c] = y;
    }
  }
}

constexpr int EB_CHUNK = 2048;     // token ids scanned per round: 512 per wave (the match lists live in LDS)
constexpr int EB_ROWS = 16;        // vocabulary rows per workgroup, at most

// One workgroup owns RV <= 16 consecutive vocabulary rows (RV = 1 for the synthetic V = 1000: a workgroup per row as in round 3;
// a dataset vocabulary of ~20 k words: 10 rows per workgroup, 2000 workgroups): the token ids of a round are loaded ONCE per
// workgroup and compared against each of its rows from registers, so the work is O(V / RV * T) id loads + O(V * T / 64) compares,
// not O(V * T) loads (at V = 20 k the one-row form re-read the whole id list 20 000 times).
template <bool TANH, bool DROP = false>
__global__ void __launch_bounds__(256) embed_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                            const long long* __restrict__ ids, int T, int V, int E, int RV,
                                                            float* __restrict__ dW, int swapN, const uint8_t* __restrict__ keep = nullptr,
                                                            uint64_t seed = 0, uint32_t thr = 0u, float inv_keep = 1.0f) {
  __shared__ int list[4][EB_CHUNK / 4];
  __shared__ int count[4];
  const int v0 = blockIdx.x * RV, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // every thread owns columns tid, tid + 256, ... (E <= 1024: at most 4) of each of the workgroup's rows
  float acc[EB_ROWS][4];
#pragma unroll
  for (int r = 0; r < EB_ROWS; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;
  for (int t0 = 0; t0 < T; t0 += EB_CHUNK) {
    const int b0 = t0 + wave * (EB_CHUNK / 4);
    long long idv[EB_CHUNK / 4 / 64];
#pragma unroll
    for (int q = 0; q < EB_CHUNK / 4 / 64; ++q) {        // the 8 id loads of a wave are issued together, once per round
      const int t = b0 + 64 * q + lane;
      idv[q] = t < T ? ids[tok_of_row(t, T, swapN)] : -1;
    }
#pragma unroll
    for (int r = 0; r < EB_ROWS; ++r) {
      if (r >= RV) break;                                // (uniform)
      const long long v = v0 + r;
      {                                                  // wave w lists the matches among its 512 tokens, in order
        int n = 0;
#pragma unroll
        for (int q = 0; q < EB_CHUNK / 4 / 64; ++q) {
          const bool hit = idv[q] == v;
          const unsigned long long m = __ballot(hit);
          if (hit) list[wave][n + __popcll(m & ((1ull << lane) - 1ull))] = b0 + 64 * q + lane;
          n += __popcll(m);
        }
        if (lane == 0) count[wave] = n;
      }
      __syncthreads();
      for (int w = 0; w < 4; ++w) {                      // wave lists in wave order = token order: a fixed summation order
        const int n = count[w];
        for (int i = 0; i < n; ++i) {
          const long long row = (long long)list[w][i] * E;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int c = tid + 256 * k;
            if (c < E) {
              if (TANH) {
                const float y = out[row + c];
                acc[r][k] += dout[row + c] * (1.0f - y * y);
              } else if (DROP) {
                acc[r][k] += dout[row + c] * keep_scale(keep, seed, thr, inv_keep, row + c);
              } else {
                acc[r][k] += dout[row + c];
              }
            }
          }
        }
      }
      __syncthreads();                                   // the lists are rewritten for the next row / round
    }
  }
#pragma unroll
  for (int r = 0; r < EB_ROWS; ++r) {
    if (r >= RV || v0 + r >= V) break;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = tid + 256 * k;
      if (c < E) dW[(long long)(v0 + r) * E + c] = acc[r][k];
    }
  }
}

static int embed_rows_per_wg(int V) {
  int rv = (V + 2047) / 2048;
  return rv < 1 ? 1 : (rv > EB_ROWS ? EB_ROWS : rv);
}

}  // namespace

extern "C" {

int vqf_embed_tanh_fwd(const float* W, const long long* ids, int T, int V, int E, float* out, void* stream) {
  if (!W || !ids || !out || T <= 0 || V <= 0 || E <= 0) return VQF_E_BADARG;
  vqf_prof_dims(T, V, E);
  VQF_LAUNCH(KID_EMBED_FWD, embed_fwd_kernel<true>, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, ids, T, V, E, out, 0);
  return vqf_last_error();
}

int vqf_embed_tanh_bwd(const float* dout, const float* out, const long long* ids, int T, int V, int E, float* dW,
                       void* stream) {
  if (!dout || !out || !ids || !dW || T <= 0 || V <= 0 || E <= 0) return VQF_E_BADARG;
  if (E > 1024) return VQF_E_UNSUPPORTED;
  vqf_prof_dims(T, V, E);
  const int rv = embed_rows_per_wg(V);
  VQF_LAUNCH(KID_EMBED_BWD, embed_bwd_kernel<true>, dim3((V + rv - 1) / rv), dim3(256), 0, (hipStream_t)stream, dout, out, ids, T, V, E,
             rv, dW, 0);
  return vqf_last_error();
}

int vqf_embed_fwd(const float* W, const long long* ids, int T, int V, int E, float* out, void* stream) {
  if (!W || !ids || !out || T <= 0 || V <= 0 || E <= 0) return VQF_E_BADARG;
  vqf_prof_dims(T, V, E);
  VQF_LAUNCH(KID_EMBED_FWD, embed_fwd_kernel<false>, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, ids, T, V, E, out, 0);
  return vqf_last_error();
}

int vqf_embed_bwd(const float* dout, const long long* ids, int T, int V, int E, float* dW, void* stream) {
  if (!dout || !ids || !dW || T <= 0 || V <= 0 || E <= 0) return VQF_E_BADARG;
  if (E > 1024) return VQF_E_UNSUPPORTED;
  vqf_prof_dims(T, V, E);
  const float* none = nullptr;
  const int rv = embed_rows_per_wg(V);
  VQF_LAUNCH(KID_EMBED_BWD, embed_bwd_kernel<false>, dim3((V + rv - 1) / rv), dim3(256), 0, (hipStream_t)stream, dout, none, ids, T, V,
             E, rv, dW, 0);
  return vqf_last_error();
}

// out = dropout(W[ids]) and its weight gradient dW[v] = sum_{t: ids[t] == v} dout[t] * keep / (1 - p): hieCoAtten.py:27-28 in one pass
// each way.  The mask is the one vqf_dropout_f32 draws over the flat (T, E) tensor (same seed -> same bits as lookup + dropout).
int vqf_embed_dropout_fwd(const float* W, const long long* ids, int T, int V, int E, const uint8_t* keep, uint64_t seed, float p_drop,
                          float* out, void* stream) {
  if (!W || !ids || !out || T <= 0 || V <= 0 || E <= 0 || p_drop < 0.f || p_drop >= 1.f) return VQF_E_BADARG;
  if (E % 4) return VQF_E_UNSUPPORTED;
  const uint32_t thr = (keep || p_drop == 0.f) ? 0u : drop_threshold_host(p_drop);
  const float inv_keep = (keep || p_drop > 0.f) ? 1.0f / (1.0f - p_drop) : 1.0f;
  vqf_prof_dims(T, V, E);
  VQF_LAUNCH(KID_EMBED_FWD, (embed_fwd_kernel<false, true>), dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, ids, T, V, E, out, 0,
             keep, seed, thr, inv_keep);
  return vqf_last_error();
}

int vqf_embed_dropout_bwd(const float* dout, const long long* ids, int T, int V, int E, const uint8_t* keep, uint64_t seed, float p_drop,
                          float* dW, void* stream) {
  if (!dout || !ids || !dW || T <= 0 || V <= 0 || E <= 0 || p_drop < 0.f || p_drop >= 1.f) return VQF_E_BADARG;
  if (E > 1024 || (E % 4)) return VQF_E_UNSUPPORTED;
  const uint32_t thr = (keep || p_drop == 0.f) ? 0u : drop_threshold_host(p_drop);
  const float inv_keep = (keep || p_drop > 0.f) ? 1.0f / (1.0f - p_drop) : 1.0f;
  vqf_prof_dims(T, V, E);
  const float* none = nullptr;
  const int rv = embed_rows_per_wg(V);
  VQF_LAUNCH(KID_EMBED_BWD, (embed_bwd_kernel<false, true>), dim3((V + rv - 1) / rv), dim3(256), 0, (hipStream_t)stream, dout, none, ids, T,
             V, E, rv, dW, 0, keep, seed, thr, inv_keep);
  return vqf_last_error();
}

// time-major forms: ids (N, Tq) int64, out / dout rows ordered (Tq, N)
int vqf_embed_tanh_fwd_tm(const float* W, const long long* ids, int N, int Tq, int V, int E, float* out, void* stream) {
  if (!W || !ids || !out || N <= 0 || Tq <= 0 || V <= 0 || E <= 0) return VQF_E_BADARG;
  const int T = N * Tq;
  vqf_prof_dims(T, V, E);
  VQF_LAUNCH(KID_EMBED_FWD, embed_fwd_kernel<true>, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, ids, T, V, E, out, N);
  return vqf_last_error();
}

int vqf_embed_tanh_bwd_tm(const float* dout, const float* out, const long long* ids, int N, int Tq, int V, int E, float* dW,
                          void* stream) {
  if (!dout || !out || !ids || !dW || N <= 0 || Tq <= 0 || V <= 0 || E <= 0) return VQF_E_BADARG;
  if (E > 1024) return VQF_E_UNSUPPORTED;
  const int T = N * Tq;
  vqf_prof_dims(T, V, E);
  const int rv = embed_rows_per_wg(V);
  VQF_LAUNCH(KID_EMBED_BWD, embed_bwd_kernel<true>, dim3((V + rv - 1) / rv), dim3(256), 0, (hipStream_t)stream, dout, out, ids, T, V, E,
             rv, dW, N);
  return vqf_last_error();
}

}  // extern "C"
