// Whole-sequence ("persistent") form of the small-batch LSTM recursion of lstm.hip: ONE launch runs
// all S steps; the H/4 workgroups exchange the recurrent state through HBM inside the launch.
//
// Why: with one launch per step (lstm.hip) a step costs 10.2 us forward / 13.9 us backward
// (rocprofv3, S=512, B=14, H=1024), of which ~1.5 us is the kernel boundary and ~2.5 us the re-read
// of the 16 MB W_hh out of the Infinity Cache; the 0.85 us of MFMA work is a small part.  Here W_hh
// stays in registers for the whole sequence (each lane keeps its 64 operand values) and a step ends
// with a hand-off instead of a kernel boundary.
//
// Hand-off protocol (MI355X guide, Guideline 16 / visibility table, first row): the per-XCD L2s are
// not coherent, so every byte another workgroup will read is stored WRITE-THROUGH (sc1, 16-byte
// buffer stores), every storing wave drains (s_waitcnt vmcnt(0)), the workgroup barriers, ONE lane
// adds to the step's counter with an agent-scope atomic; consumers poll that counter with a relaxed
// agent-scope (sc1) load from ONE lane, barrier, and then read the bytes with sc1 loads only.  One
// counter per step (zeroed by a memset node in front of the launch): nothing is ever reset or reused
// inside the launch.  Every spin is bounded; a timeout raises the abort word, which every other spin
// also watches, so the grid always drains; the outputs are then poisoned with NaN and
// vqf_lstm_persist_status() reports VQF_E_TIMEOUT.
//
// forward  (all-gather):     step s reads all of h_{s-1} (B x H, 56 KB) straight out of the output
//                            tensor hs, each workgroup publishes its B x 4 slice of h_s.
// backward (reduce-scatter): workgroup q owns gate rows J_q of W_hh (16 x H, in registers) and its
//                            own dG_s[:, J_q]; it publishes the partial product dG_s[:, J_q] W_hh[J_q, :]
//                            (B x H) cut into H/4 pieces, and sums the H/4 pieces addressed to its own
//                            4 hidden units in a fixed order (deterministic, no atomics on data).
//                            dG never crosses workgroups.  MFMA orientation D[n][b] = W^T dG^T makes a
//                            lane's 4 accumulator registers 4 consecutive n: one 16-byte store each.
// Constraints: B <= 16, H in {256, 512, 768, 1024}; grid = H/4 workgroups of 256 threads, all
// co-resident by construction (<= 256 small workgroups; several fit on one CU).
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int UPB = 4;
constexpr int SYNC_HDR = 16;              // sync[0] = abort word; counters start at sync[16]
constexpr unsigned SPIN_LIMIT = 1u << 22; // ~ seconds; only a lost workgroup can reach it

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16));   // aux 16 = sc1
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 16);
}

// ONE lane: wait until *c >= target.  Bounded; raises / follows the abort word.
__device__ __forceinline__ void wait_count(gu32* c, unsigned target, gu32* abortw) {
  unsigned spins = 0;
  while (__hip_atomic_load(c, RLX_AGENT) < target) {
    if ((++spins & 31u) == 0u) {
      if (spins > SPIN_LIMIT || __hip_atomic_load(abortw, RLX_AGENT) != 0u) {
        __hip_atomic_store(abortw, 1u, RLX_AGENT);
        break;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// ------------------------------------------------------------------------------------------------
template <int KI>
__global__ void __launch_bounds__(256, 1)
lstm_fwd_persist_kernel(const float* __restrict__ xw, const float* __restrict__ w_hh, int S, int B,
                        float* hs, float* __restrict__ cs, float* __restrict__ gates_out, unsigned* sync) {
  constexpr int H = 256 * KI, KW = H / 4, NC = KW / 16;
  __shared__ float part[4][16][16];          // [wave][b][n = gate*4 + u]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * UPB;
  const unsigned NB = gridDim.x;
  gu32* abortw = (gu32*)sync;
  gu32* cnt = (gu32*)(sync + SYNC_HDR);

  // B operand of every step: rows n = gate*4 + u of W_hh, this wave's K range, 64 values per lane
  f32x4 wv[NC];
  {
    const float* wr = w_hh + (long long)((r >> 2) * H + u0 + (r & 3)) * H + wave * KW + 4 * g;
#pragma unroll
    for (int c = 0; c < NC; ++c) wv[c] = *reinterpret_cast<const f32x4*>(wr + 16 * c);
  }
  const bool own = tid < B * UPB;            // (b, u) owners: all inside wave 0 (B <= 16)
  const int ob = tid / UPB, ou = tid % UPB, col = u0 + ou;
  const float keep = r < B ? 1.f : 0.f;
  const unsigned hoff = (unsigned)(((r < B ? r : 0) * H + wave * KW + 4 * g) * 4);
  const unsigned step_bytes = (unsigned)(B * H * 4);
  float c_prev = 0.f;

  for (int s = 0; s < S; ++s) {
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (own) {                               // issued before the wait: its latency hides behind the poll
      const float* xp = xw + ((long long)s * B + ob) * 4 * H + col;
      x[0] = xp[0]; x[1] = xp[H]; x[2] = xp[2 * H]; x[3] = xp[3 * H];
    }
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      if (tid == 0) wait_count(cnt + (s - 1), NB, abortw);
      __syncthreads();
      const __amdgpu_buffer_rsrc_t rh = make_rsrc(hs + (long long)(s - 1) * B * H, step_bytes);
      f32x4 hv[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) hv[c] = ld16_sc1(rh, hoff + 64u * c);
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
#pragma unroll
      for (int q = 0; q < 4; ++q) part[wave][4 * g + q][r] = acc0[q] + acc1[q];   // D[b = 4g+q][n = r]
      __syncthreads();
      if (own) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int k = 0; k < 4; ++k) pre[k] += part[w][ob][4 * k + ou];
      }
    }
    float h = 0.f;
    if (own) {
      const float gi = sigmoidf_(pre[0] + x[0]);
      const float gf = sigmoidf_(pre[1] + x[1]);
      const float gg = tanhf(pre[2] + x[2]);
      const float go = sigmoidf_(pre[3] + x[3]);
      const float c = gf * c_prev + gi * gg;
      h = go * tanhf(c);
      c_prev = c;
      cs[((long long)s * B + ob) * H + col] = c;
      float* gt = gates_out + ((long long)s * B + ob) * 4 * H + col;
      gt[0] = gi; gt[H] = gf; gt[2 * H] = gg; gt[3 * H] = go;
    }
    if (wave == 0) {                          // the 4 units of a row -> one write-through 16-byte store
      const float h1 = __shfl(h, lane + 1, 64), h2 = __shfl(h, lane + 2, 64), h3 = __shfl(h, lane + 3, 64);
      if (own && ou == 0) {
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(hs + (long long)s * B * H, step_bytes);
        st16_sc1(ro, (unsigned)((ob * H + u0) * 4), f32x4{h, h1, h2, h3});
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the only storing wave drains ...
      if (tid == 0) __hip_atomic_fetch_add(cnt + s, 1u, RLX_AGENT);   // ... then ONE lane signals
    }
  }
  if (__hip_atomic_load(abortw, RLX_AGENT) != 0u && own)   // make a timeout visible in the results
    hs[((long long)(S - 1) * B + ob) * H + col] = __builtin_nanf("");
}

// ------------------------------------------------------------------------------------------------
template <int KI>
__global__ void __launch_bounds__(256, 1)
lstm_bwd_persist_kernel(const float* __restrict__ dhs, const float* __restrict__ gates,
                        const float* __restrict__ cs, const float* __restrict__ w_hh, int S, int B,
                        float* __restrict__ dgates, float* pbuf, unsigned* sync) {
  constexpr int H = 256 * KI, H4 = 4 * H, NT = H / 64;     // N tiles (16 wide) per wave
  constexpr int NBLK = H / UPB, NLD = NBLK / 16;           // source workgroups; loads per thread in the reduce
  __shared__ float dgl[16][16];                            // own dG_s: [b][kappa = gate*4 + u]
  __shared__ __attribute__((aligned(16))) float red[16][16][4];   // [source lane][b][u]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int blk = blockIdx.x, u0 = blk * UPB;
  gu32* abortw = (gu32*)sync;
  gu32* cnt = (gu32*)(sync + SYNC_HDR);

  // A operand: W_hh[j(kappa)][n] for kappa = 4kk + g, n = n0(t) + r
  float wreg[NT][4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int kap = 4 * kk + g;
    const float* wr = w_hh + (long long)((kap >> 2) * H + u0 + (kap & 3)) * H + wave * (NT * 16) + r;
#pragma unroll
    for (int t = 0; t < NT; ++t) wreg[t][kk] = wr[16 * t];
  }
  const bool own = tid < B * UPB;
  const int ob = tid / UPB, ou = tid % UPB, col = u0 + ou;
  const int pb = tid & 15, pl = tid >> 4;                  // reduce: row b, source lane
  const unsigned half_bytes = (unsigned)NBLK * NBLK * B * 16u;     // one parity of pbuf
  dgl[tid >> 4][tid & 15] = 0.f;                           // rows >= B stay zero
  float dc_carry = 0.f;
  __syncthreads();

  for (int s = S - 1; s >= 0; --s) {
    float dh_in = 0.f, gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, c_s = 0.f, cp = 0.f;
    if (own) {                                             // issued before the wait
      const long long bh = ((long long)s * B + ob) * H + col;
      dh_in = dhs[bh];
      const float* gt = gates + ((long long)s * B + ob) * H4 + col;
      gi = gt[0]; gf = gt[H]; gg = gt[2 * H]; go = gt[3 * H];
      c_s = cs[bh];
      cp = s > 0 ? cs[bh - (long long)B * H] : 0.f;
    }
    float dhc = 0.f;
    if (s < S - 1) {
      if (tid == 0) wait_count(cnt + (s + 1), NBLK, abortw);
      __syncthreads();
      // pieces addressed to this workgroup: pbuf[parity(s+1)][blk][q = 0..NBLK-1][b][4]
      const __amdgpu_buffer_rsrc_t rp =
          make_rsrc(pbuf + (size_t)((s + 1) & 1) * (half_bytes / 4) + (size_t)blk * NBLK * B * 4, (unsigned)NBLK * B * 16u);
      f32x4 pv[NLD];
#pragma unroll
      for (int m = 0; m < NLD; ++m) pv[m] = ld16_sc1(rp, (unsigned)(((pl + 16 * m) * B + (pb < B ? pb : 0)) * 16));
      f32x4 a = {0, 0, 0, 0};
#pragma unroll
      for (int m = 0; m < NLD; ++m) a += pv[m];
      *reinterpret_cast<f32x4*>(&red[pl][pb][0]) = a;
      __syncthreads();
      if (own) {
#pragma unroll
        for (int l = 0; l < 16; ++l) dhc += red[l][ob][ou];
      }
    }
    if (own) {
      const float dh = dh_in + dhc;
      const float tc = tanhf(c_s);
      const float dc = dc_carry + dh * go * (1.0f - tc * tc);
      const float d0 = dc * gg * gi * (1.0f - gi);
      const float d1 = dc * cp * gf * (1.0f - gf);
      const float d2 = dc * gi * (1.0f - gg * gg);
      const float d3 = dh * tc * go * (1.0f - go);
      float* d = dgates + ((long long)s * B + ob) * H4 + col;
      d[0] = d0; d[H] = d1; d[2 * H] = d2; d[3 * H] = d3;
      dgl[ob][ou] = d0; dgl[ob][4 + ou] = d1; dgl[ob][8 + ou] = d2; dgl[ob][12 + ou] = d3;
      dc_carry = dc * gf;
    }
    if (s > 0) {
      __syncthreads();                                     // dgl complete
      float bf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) bf[kk] = dgl[r][4 * kk + g];
      const __amdgpu_buffer_rsrc_t rw = make_rsrc(pbuf + (size_t)(s & 1) * (half_bytes / 4), half_bytes);
#pragma unroll
      for (int t0 = 0; t0 < NT; t0 += 4) {
        f32x4v acc[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = f32x4v{0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[t0 + tt][kk], bf[kk], acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {                   // D[n = n0 + 4g + q][b = r]: 4 consecutive n per lane
          const int ug = (wave * NT + t0 + tt) * 4 + g;    // destination workgroup (owner of those 4 units)
          if (r < B)
            st16_sc1(rw, (unsigned)(((ug * NBLK + blk) * B + r) * 16),
                     f32x4{acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]});
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains ...
      __syncthreads();                                     // ... all of them have ...
      if (tid == 0) __hip_atomic_fetch_add(cnt + s, 1u, RLX_AGENT);   // ... then ONE lane signals
    }
  }
  if (__hip_atomic_load(abortw, RLX_AGENT) != 0u && own)
    dgates[(long long)ob * H4 + col] = __builtin_nanf("");
}

size_t sync_bytes(int S) { return (((size_t)(SYNC_HDR + S) * 4 + 255) / 256) * 256; }
size_t pbuf_bytes(int B, int H) { return (size_t)2 * (H / UPB) * (H / UPB) * B * 16; }

}  // namespace

extern "C" {

int vqf_lstm_persist_supported(int B, int H) {
  return (B >= 1 && B <= 16 && (H == 256 || H == 512 || H == 768 || H == 1024)) ? 1 : 0;
}

size_t vqf_lstm_persist_ws_bytes(int S, int B, int H) {
  if (S <= 0 || !vqf_lstm_persist_supported(B, H)) return 0;
  return sync_bytes(S) + pbuf_bytes(B, H);
}

int vqf_lstm_seq_fwd_persist(const float* xw, const float* w_hh, int S, int B, int H, float* hs, float* cs,
                             float* gates, void* ws, size_t ws_bytes, void* stream) {
  if (!xw || !w_hh || !hs || !cs || !gates || !ws || S <= 0) return VQF_E_BADARG;
  if (!vqf_lstm_persist_supported(B, H)) return VQF_E_UNSUPPORTED;
  if (!aligned16(xw) || !aligned16(w_hh) || !aligned16(hs) || !aligned16(ws)) return VQF_E_ALIGN;
  if (ws_bytes < sync_bytes(S)) return VQF_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(ws, 0, sync_bytes(S), s);
  if (e != hipSuccess) return (int)e;
  unsigned* sync = (unsigned*)ws;
  vqf_prof_dims(S, B, H);
  dim3 grid(H / UPB), block(256);
  switch (H / 256) {
    case 1: VQF_LAUNCH(KID_LSTM_FWD, lstm_fwd_persist_kernel<1>, grid, block, 0, s, xw, w_hh, S, B, hs, cs, gates, sync); break;
    case 2: VQF_LAUNCH(KID_LSTM_FWD, lstm_fwd_persist_kernel<2>, grid, block, 0, s, xw, w_hh, S, B, hs, cs, gates, sync); break;
    case 3: VQF_LAUNCH(KID_LSTM_FWD, lstm_fwd_persist_kernel<3>, grid, block, 0, s, xw, w_hh, S, B, hs, cs, gates, sync); break;
    default: VQF_LAUNCH(KID_LSTM_FWD, lstm_fwd_persist_kernel<4>, grid, block, 0, s, xw, w_hh, S, B, hs, cs, gates, sync); break;
  }
  return vqf_last_error();
}

int vqf_lstm_seq_bwd_persist(const float* dhs, const float* gates, const float* cs, const float* w_hh, int S, int B,
                             int H, float* dgates, void* ws, size_t ws_bytes, void* stream) {
  if (!dhs || !gates || !cs || !w_hh || !dgates || !ws || S <= 0) return VQF_E_BADARG;
  if (!vqf_lstm_persist_supported(B, H)) return VQF_E_UNSUPPORTED;
  if (!aligned16(ws)) return VQF_E_ALIGN;
  if (ws_bytes < sync_bytes(S) + pbuf_bytes(B, H)) return VQF_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(ws, 0, sync_bytes(S), s);
  if (e != hipSuccess) return (int)e;
  unsigned* sync = (unsigned*)ws;
  float* pbuf = (float*)((char*)ws + sync_bytes(S));
  vqf_prof_dims(S, B, H);
  dim3 grid(H / UPB), block(256);
  switch (H / 256) {
    case 1: VQF_LAUNCH(KID_LSTM_BWD, lstm_bwd_persist_kernel<1>, grid, block, 0, s, dhs, gates, cs, w_hh, S, B, dgates, pbuf, sync); break;
    case 2: VQF_LAUNCH(KID_LSTM_BWD, lstm_bwd_persist_kernel<2>, grid, block, 0, s, dhs, gates, cs, w_hh, S, B, dgates, pbuf, sync); break;
    case 3: VQF_LAUNCH(KID_LSTM_BWD, lstm_bwd_persist_kernel<3>, grid, block, 0, s, dhs, gates, cs, w_hh, S, B, dgates, pbuf, sync); break;
    default: VQF_LAUNCH(KID_LSTM_BWD, lstm_bwd_persist_kernel<4>, grid, block, 0, s, dhs, gates, cs, w_hh, S, B, dgates, pbuf, sync); break;
  }
  return vqf_last_error();
}

int vqf_lstm_persist_status(const void* ws, void* stream) {
  if (!ws) return VQF_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  unsigned word = 0;
  hipError_t e = hipMemcpyAsync(&word, ws, sizeof(word), hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return (int)e;
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) return (int)e;
  return word ? VQF_E_TIMEOUT : VQF_OK;
}

}  // extern "C"
