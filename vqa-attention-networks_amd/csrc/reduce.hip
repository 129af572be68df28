// Small HBM-bound helpers: column sums (bias gradients), slab reducers, ReLU
// backward, row scaling, row dot products and the F.normalize coefficients.
#include "common.h"

namespace {

constexpr int CS_ROWS = 256;   // rows folded by one colsum block ...
// ... of the scalar kernel; the 16-byte kernels fold fewer rows per workgroup on tall tensors, so that a (50176, 512) tensor is
// 784 workgroups, not 196 (one column tile wide: fewer workgroups than CUs), and fewer still on short ones: a (512, 5000) bias
// gradient was 10 workgroups (15-24 us for 10 MB, five times per MFB step); ~512 workgroups when the tensor allows, >= 16 rows each
static inline int cs_rows_vec(int M, int N) {
  const long long tiles_c = (N / 4 + 255) / 256;
  long long rpb = ((M * tiles_c / 512 + 7) / 8) * 8;
  const int cap = M > 16384 ? 64 : CS_ROWS;
  return (int)(rpb < 16 ? 16 : (rpb > cap ? cap : rpb));
}

// partial[b, c] = sum over rows [b*CS_ROWS, ...) of in[r, c];  thread = column
__global__ void colsum_partial_kernel(const float* __restrict__ in, int M, int N, int ld,
                                      float* __restrict__ partial) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += in[(long long)r * ld + c];
    a1 += in[(long long)(r + 1) * ld + c];
    a2 += in[(long long)(r + 2) * ld + c];
    a3 += in[(long long)(r + 3) * ld + c];
  }
  for (; r < r1; ++r) a0 += in[(long long)r * ld + c];
  partial[(long long)blockIdx.y * N + c] = (a0 + a1) + (a2 + a3);
}

// The same with 16-byte loads: thread = 4 consecutive columns x one of RS row slots (RS = 256 / (columns / 4) per column tile of
// <= 1024 columns), eight rows in flight per thread, the slots folded through LDS in a fixed order.  N % 4 == 0, ld % 4 == 0.
__global__ void __launch_bounds__(256) colsum_partial_vec_kernel(const float* __restrict__ in, int M, int N, int ld, int rpb,
                                                                 float* __restrict__ partial) {
  __shared__ f32x4 red[256];
  const int tile_c4 = min(256, (N >> 2) - blockIdx.x * 256);      // float4 columns of this column tile
  int CT = 1;
  while (CT < tile_c4) CT <<= 1;                                  // threads per row: a power of two >= the tile's float4 columns
  const int RS = 256 / CT;
  const int tid = threadIdx.x, c4 = tid % CT, rs = tid / CT;
  const bool live = c4 < tile_c4;
  const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
  const float* p = in + 4ll * (blockIdx.x * 256 + c4);
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  if (live) {
    int r = r0 + rs;
    for (; r + 7 * RS < r1; r += 8 * RS) {
      f32x4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(p + (long long)(r + q * RS) * ld);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q & 3] += v[q];
    }
    for (; r < r1; r += RS) acc[0] += *reinterpret_cast<const f32x4*>(p + (long long)r * ld);
  }
  red[tid] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (rs == 0 && live) {
    f32x4 sum = red[c4];
    for (int q = 1; q < RS; ++q) sum += red[q * CT + c4];
    *reinterpret_cast<f32x4*>(partial + (long long)blockIdx.y * N + 4ll * (blockIdx.x * 256 + c4)) = sum;
  }
}

// out[g, c] = sum_{j<J} in[(g*J + j), c]
__global__ void group_reduce_kernel(const float* __restrict__ in, int G, int J, int W,
                                    float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (c >= W) return;
  const float* p = in + (long long)g * J * W + c;
  float a0 = 0.f, a1 = 0.f;
  int j = 0;
  for (; j + 1 < J; j += 2) { a0 += p[(long long)j * W]; a1 += p[(long long)(j + 1) * W]; }
  if (j < J) a0 += p[(long long)j * W];
  out[(long long)g * W + c] = a0 + a1;
}

// out[c] = sum_{j<J} in[j, c] in ONE launch for a few hundred partial rows (the tail of every bias-gradient reduction: two launches
// before round 5): a workgroup = 16 columns x 16 row slots, slot q folds rows q, q + 16, ... (eight loads in flight), the slots are
// added through LDS in slot order.  Deterministic; W / 16 workgroups.
__global__ void __launch_bounds__(256) colreduce_one_kernel(const float* __restrict__ in, int J, int W, float* __restrict__ out) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < W) {
    int j = q;
    for (; j + 7 * 16 < J; j += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = in[(long long)(j + 16 * u) * W + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u & 3] += v[u];
    }
    for (; j < J; j += 16) a[0] += in[(long long)j * W + c];
  }
  red[q][cl] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (q == 0 && c < W) {
    float sum = red[0][cl];
#pragma unroll
    for (int u = 1; u < 16; ++u) sum += red[u][cl];
    out[c] = sum;
  }
}

// stage 1 of the two-stage column reduction: block (x, y) folds rows [y*chunk, (y+1)*chunk)
__global__ void group_reduce_stage1_kernel(const float* __restrict__ in, int J, int W, int chunk,
                                           float* __restrict__ part) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= W) return;
  const int j0 = blockIdx.y * chunk, j1 = min(J, j0 + chunk);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int j = j0;
  for (; j + 3 < j1; j += 4) {
    a0 += in[(long long)j * W + c];
    a1 += in[(long long)(j + 1) * W + c];
    a2 += in[(long long)(j + 2) * W + c];
    a3 += in[(long long)(j + 3) * W + c];
  }
  for (; j < j1; ++j) a0 += in[(long long)j * W + c];
  part[(long long)blockIdx.y * W + c] = (a0 + a1) + (a2 + a3);
}

__global__ void relu_bwd_kernel(const float* __restrict__ dX, const float* __restrict__ Y,
                                long long n, float* __restrict__ dXpre) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dXpre[i] = Y[i] > 0.f ? dX[i] : 0.f;
}

// dXpre[m, c] = (dX[m, c] + wts[m] * dpooled[m / L, c]) * (Y[m, c] > 0 ? scale : 0): backward of Y = dropout(relu(pre)) whose
// output is ALSO pooled by an attention head (the rank-1 term is that head's gradient: never materialised); in place allowed.
// Workgroup = CS_ROWS rows x a column tile of <= 1024 columns, thread = 4 columns x a row slot (as colsum_partial_vec_kernel);
// partial != nullptr: the column sums of the workgroup's rows go to partial[blockIdx.y][:] (the bias gradient, no second pass).
__global__ void __launch_bounds__(256) relu_bwd_rank1_kernel(const float* __restrict__ dX, const float* __restrict__ Y,
                                                             const float* __restrict__ wts, const float* __restrict__ dpooled,
                                                             int L, float scale, int M, int C, int rpb,
                                                             float* __restrict__ dXpre, float* __restrict__ partial) {
  __shared__ f32x4 red[256];
  const int tile_c4 = min(256, (C >> 2) - blockIdx.x * 256);
  int CT = 1;
  while (CT < tile_c4) CT <<= 1;
  const int RS = 256 / CT;
  const int tid = threadIdx.x, c4 = tid % CT, rs = tid / CT;
  const bool live = c4 < tile_c4;
  const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
  const long long col = 4ll * (blockIdx.x * 256 + c4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    for (int r = r0 + rs; r < r1; r += 2 * RS) {
      const int rb = min(r + RS, r1 - 1);                  // second row of the trip (a tail trip re-reads row r: not stored)
      const bool hasB = r + RS < r1;
      f32x4 dA = *reinterpret_cast<const f32x4*>(dX + (long long)r * C + col);
      f32x4 dB = *reinterpret_cast<const f32x4*>(dX + (long long)rb * C + col);
      const f32x4 yA = *reinterpret_cast<const f32x4*>(Y + (long long)r * C + col);
      const f32x4 yB = *reinterpret_cast<const f32x4*>(Y + (long long)rb * C + col);
      if (wts) {
        dA += *reinterpret_cast<const f32x4*>(dpooled + (long long)(r / L) * C + col) * wts[r];
        dB += *reinterpret_cast<const f32x4*>(dpooled + (long long)(rb / L) * C + col) * wts[rb];
      }
      f32x4 oA, oB;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        oA[j] = yA[j] > 0.f ? dA[j] * scale : 0.f;
        oB[j] = yB[j] > 0.f ? dB[j] * scale : 0.f;
      }
      *reinterpret_cast<f32x4*>(dXpre + (long long)r * C + col) = oA;
      acc += oA;
      if (hasB) {
        *reinterpret_cast<f32x4*>(dXpre + (long long)rb * C + col) = oB;
        acc += oB;
      }
    }
  }
  if (!partial) return;
  red[tid] = acc;
  __syncthreads();
  if (rs == 0 && live) {
    f32x4 sum = red[c4];
    for (int q = 1; q < RS; ++q) sum += red[q * CT + c4];
    *reinterpret_cast<f32x4*>(partial + (long long)blockIdx.y * C + col) = sum;
  }
}

// out[i] = a[i] + b[i] for up to 4 (a, b, out, n) segments in one launch (gradient halves of concatenated weights)
struct AddSegs { const float* a[4]; const float* b[4]; float* out[4]; long long n[4]; int count; };
__global__ void multi_add_kernel(const AddSegs sg) {
  const int k = blockIdx.y;
  if (k >= sg.count) return;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < sg.n[k]; i += stride) sg.out[k][i] = sg.a[k][i] + sg.b[k][i];
}

// dst[k][0 .. n[k]) = src[k][0 .. n[k]) for up to 8 segments in one launch (packing weights of layers that share an input)
struct CopySegs { const float* src[8]; float* dst[8]; long long n[8]; int count; };
__global__ void multi_copy_kernel(const CopySegs sg) {
  const int k = blockIdx.y;
  if (k >= sg.count) return;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < sg.n[k]; i += stride) sg.dst[k][i] = sg.src[k][i];
}

// one wave per row: Y[m,:] = R[m,:] * inv[m / L]
__global__ void scale_rows_kernel(const float* __restrict__ R, const float* __restrict__ inv,
                                  int M, int L, int W, float* __restrict__ Y) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const float s = inv[row / L];
  const float* r = R + (long long)row * W;
  float* y = Y + (long long)row * W;
  if ((W & 3) == 0 && aligned16_dev(r) && aligned16_dev(y)) {
    for (int c = lane * 4; c < W; c += 256) {
      f32x4 v = *reinterpret_cast<const f32x4*>(r + c);
      v *= s;
      *reinterpret_cast<f32x4*>(y + c) = v;
    }
  } else {
    for (int c = lane; c < W; c += 64) y[c] = r[c] * s;
  }
}

__global__ void rowdot_kernel(const float* __restrict__ Y, const float* __restrict__ dY, int M,
                              int W, float* __restrict__ out) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const float* a = Y + (long long)row * W;
  const float* b = dY + (long long)row * W;
  float acc = 0.f;
  if ((W & 3) == 0 && aligned16_dev(a) && aligned16_dev(b)) {
    for (int c = lane * 4; c < W; c += 256) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(a + c);
      const f32x4 y = *reinterpret_cast<const f32x4*>(b + c);
      acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
  } else {
    for (int c = lane; c < W; c += 64) acc += a[c] * b[c];
  }
  acc = wave_sum(acc);
  if (lane == 0) out[row] = acc;
}

// one wave per sample
__global__ void l2_group_norm_kernel(const float* __restrict__ rowssq, int N, int L,
                                     float* __restrict__ norm, float* __restrict__ inv) {
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  float a = 0.f;
  for (int l = lane; l < L; l += 64) a += rowssq[(long long)n * L + l];
  a = wave_sum(a);
  if (lane == 0) {
    const float nr = sqrtf(a);
    norm[n] = nr;
    inv[n] = 1.0f / fmaxf(nr, 1e-12f);      // F.normalize eps (mfb.py:105,135)
  }
}

__global__ void l2_norm_bwd_coef_lin_kernel(const float* __restrict__ dl, const float* __restrict__ lin, int G,
                                            const float* __restrict__ norm, const float* __restrict__ inv, int N, int L,
                                            float* __restrict__ coefA, float* __restrict__ coefB,
                                            float* __restrict__ unit) {
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  float a = 0.f;
  const long long base = (long long)n * L * G;
  for (int i = lane; i < L * G; i += 64) a += dl[base + i] * lin[base + i];
  a = wave_sum(a);
  if (lane == 0) {
    coefA[n] = 1.0f;
    unit[n] = 1.0f;
    coefB[n] = norm[n] > 1e-12f ? inv[n] * inv[n] * a : 0.f;   // clamp_min branch has no projection term
  }
}

__global__ void l2_norm_bwd_coef_kernel(const float* __restrict__ rowdot,
                                        const float* __restrict__ norm,
                                        const float* __restrict__ inv, int N, int L,
                                        float* __restrict__ coefA, float* __restrict__ coefB) {
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  float a = 0.f;
  for (int l = lane; l < L; l += 64) a += rowdot[(long long)n * L + l];
  a = wave_sum(a);
  if (lane == 0) {
    coefA[n] = inv[n];
    coefB[n] = norm[n] > 1e-12f ? inv[n] * a : 0.f;   // clamp_min branch has no projection term
  }
}

}  // namespace

// out[c] = sum_{j<J} in[j, c] with enough workgroups to be bandwidth-bound: rows are folded in
// VQF_REDUCE_SPLITS chunks (grid.y) into `scratch` (VQF_REDUCE_SPLITS x W floats), then once more.
int vqf_colreduce_2stage(const float* in, int J, int W, float* out, float* scratch, hipStream_t s) {
  if (J <= 2 * VQF_REDUCE_SPLITS)
    return vqf_group_reduce_f32(in, 1, J, W, out, (void*)s);
  if (J <= 4096) {       // a few hundred partial rows: one launch (16 row slots per column) instead of two
    VQF_LAUNCH(KID_GROUP_REDUCE, colreduce_one_kernel, dim3((W + 15) / 16), dim3(256), 0, s, in, J, W, out);
    return vqf_last_error();
  }
  const int chunk = (J + VQF_REDUCE_SPLITS - 1) / VQF_REDUCE_SPLITS;
  const int ny = (J + chunk - 1) / chunk;
  dim3 grid((W + 255) / 256, ny);
  VQF_LAUNCH(KID_GROUP_REDUCE, group_reduce_stage1_kernel, grid, dim3(256), 0, s, in, J, W, chunk,
             scratch);
  int rc = vqf_last_error();
  if (rc) return rc;
  return vqf_group_reduce_f32(scratch, 1, ny, W, out, (void*)s);
}

extern "C" {

size_t vqf_colsum_ws_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  const int rpb = cs_rows_vec(M, N);
  return (size_t)((M + rpb - 1) / rpb + VQF_REDUCE_SPLITS) * (size_t)N * sizeof(float);
}

int vqf_colsum_f32(const float* dY, int M, int N, int ldy, float* db, void* ws, size_t ws_bytes,
                   void* stream) {
  if (!dY || !db || M <= 0 || N <= 0 || ldy < N) return VQF_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int nb = (M + CS_ROWS - 1) / CS_ROWS;
  dim3 grid((N + 255) / 256, nb);
  // (one block of rows: the column-per-thread kernel straight into db -- unless the 16-byte form can spread a tall-enough tensor
  //  over more workgroups: a (256, 1000) logit gradient was 4 workgroups walking 256 rows each, 20 us)
  const bool vec_ok = (N % 4) == 0 && (ldy % 4) == 0 && aligned16(dY) && ws && aligned16(ws) && ws_bytes >= vqf_colsum_ws_bytes(M, N);
  if (nb == 1 && !(vec_ok && M >= 64)) {
    VQF_LAUNCH(KID_COLSUM, colsum_partial_kernel, grid, dim3(256), 0, s, dY, M, N, ldy, db);
    return vqf_last_error();
  }
  if (!ws || ws_bytes < vqf_colsum_ws_bytes(M, N)) return VQF_E_WORKSPACE;
  int nbv = nb;
  if ((N % 4) == 0 && (ldy % 4) == 0 && aligned16(dY) && aligned16(ws)) {      // 16-byte loads, eight rows in flight per thread
    const int rpb = cs_rows_vec(M, N);
    nbv = (M + rpb - 1) / rpb;
    VQF_LAUNCH(KID_COLSUM, colsum_partial_vec_kernel, dim3((N / 4 + 255) / 256, nbv), dim3(256), 0, s, dY, M, N, ldy, rpb,
               (float*)ws);
  } else {
    VQF_LAUNCH(KID_COLSUM, colsum_partial_kernel, grid, dim3(256), 0, s, dY, M, N, ldy, (float*)ws);
  }
  int rc = vqf_last_error();
  if (rc) return rc;
  return vqf_colreduce_2stage((const float*)ws, nbv, N, db, (float*)ws + (size_t)nbv * N, s);
}

int vqf_group_reduce_f32(const float* in, int G, int J, int W, float* out, void* stream) {
  if (!in || !out || G <= 0 || J <= 0 || W <= 0) return VQF_E_BADARG;
  dim3 grid((W + 255) / 256, G);
  VQF_LAUNCH(KID_GROUP_REDUCE, group_reduce_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, G,
             J, W, out);
  return vqf_last_error();
}

int vqf_relu_bwd_f32(const float* dX, const float* Y, int M, int C, float* dXpre, float* dbias,
                     void* ws, size_t ws_bytes, void* stream) {
  if (!dX || !Y || !dXpre || M <= 0 || C <= 0) return VQF_E_BADARG;
  const long long n = (long long)M * C;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  VQF_LAUNCH(KID_RELU_BWD, relu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dX, Y,
             n, dXpre);
  int rc = vqf_last_error();
  if (rc || !dbias) return rc;
  return vqf_colsum_f32(dXpre, M, C, C, dbias, ws, ws_bytes, stream);
}

int vqf_relu_bwd_rank1_f32(const float* dX, const float* Y, const float* wts, const float* dpooled, int L, float scale, int M,
                           int C, float* dXpre, float* dbias, void* ws, size_t ws_bytes, void* stream) {
  if (!dX || !Y || !dXpre || M <= 0 || C <= 0 || (wts && (!dpooled || L <= 0))) return VQF_E_BADARG;
  if (C % 4) return VQF_E_UNSUPPORTED;
  if (!aligned16(dX) || !aligned16(Y) || !aligned16(dXpre) || (dpooled && !aligned16(dpooled))) return VQF_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int rpb = cs_rows_vec(M, C);
  const int nb = (M + rpb - 1) / rpb;
  float* partial = nullptr;
  if (dbias) {
    if (nb == 1) partial = dbias;
    else {
      if (!ws || !aligned16(ws) || ws_bytes < vqf_colsum_ws_bytes(M, C)) return VQF_E_WORKSPACE;
      partial = (float*)ws;
    }
  }
  VQF_LAUNCH(KID_RELU_BWD, relu_bwd_rank1_kernel, dim3((C / 4 + 255) / 256, nb), dim3(256), 0, s, dX, Y, wts, dpooled,
             L > 0 ? L : 1, scale, M, C, rpb, dXpre, partial);
  int rc = vqf_last_error();
  if (rc || !dbias || nb == 1) return rc;
  return vqf_colreduce_2stage(partial, nb, C, dbias, partial + (size_t)nb * C, s);
}

int vqf_multi_add_f32(const float* const* a, const float* const* b, float* const* out, const long long* n, int count, void* stream) {
  if (!a || !b || !out || !n || count <= 0 || count > 4) return VQF_E_BADARG;
  AddSegs sg;
  long long nmax = 0;
  for (int k = 0; k < count; ++k) {
    if (!a[k] || !b[k] || !out[k] || n[k] <= 0) return VQF_E_BADARG;
    sg.a[k] = a[k]; sg.b[k] = b[k]; sg.out[k] = out[k]; sg.n[k] = n[k];
    nmax = n[k] > nmax ? n[k] : nmax;
  }
  sg.count = count;
  long long blocks = (nmax + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  VQF_LAUNCH(KID_MULTI_ADD, multi_add_kernel, dim3((unsigned)blocks, count), dim3(256), 0, (hipStream_t)stream, sg);
  return vqf_last_error();
}

int vqf_multi_copy_f32(const float* const* src, float* const* dst, const long long* n, int count, void* stream) {
  if (!src || !dst || !n || count <= 0 || count > 8) return VQF_E_BADARG;
  CopySegs sg;
  long long nmax = 0;
  for (int k = 0; k < count; ++k) {
    if (!src[k] || !dst[k] || n[k] <= 0) return VQF_E_BADARG;
    sg.src[k] = src[k]; sg.dst[k] = dst[k]; sg.n[k] = n[k];
    nmax = n[k] > nmax ? n[k] : nmax;
  }
  sg.count = count;
  long long blocks = (nmax + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  VQF_LAUNCH(KID_MULTI_COPY, multi_copy_kernel, dim3((unsigned)blocks, count), dim3(256), 0, (hipStream_t)stream, sg);
  return vqf_last_error();
}

int vqf_scale_rows(const float* R, const float* inv, int M, int L, int W, float* Y, void* stream) {
  if (!R || !inv || !Y || M <= 0 || L <= 0 || W <= 0) return VQF_E_BADARG;
  VQF_LAUNCH(KID_SCALE_ROWS, scale_rows_kernel, dim3((M + 3) / 4), dim3(256), 0,
             (hipStream_t)stream, R, inv, M, L, W, Y);
  return vqf_last_error();
}

int vqf_rowdot(const float* Y, const float* dY, int M, int W, float* rowdot, void* stream) {
  if (!Y || !dY || !rowdot || M <= 0 || W <= 0) return VQF_E_BADARG;
  VQF_LAUNCH(KID_ROWDOT, rowdot_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, Y,
             dY, M, W, rowdot);
  return vqf_last_error();
}

int vqf_l2_group_norm(const float* rowssq, int N, int L, float* norm, float* inv, void* stream) {
  if (!rowssq || !norm || !inv || N <= 0 || L <= 0) return VQF_E_BADARG;
  VQF_LAUNCH(KID_L2_GROUP_NORM, l2_group_norm_kernel, dim3((N + 3) / 4), dim3(256), 0,
             (hipStream_t)stream, rowssq, N, L, norm, inv);
  return vqf_last_error();
}

int vqf_l2_norm_bwd_coef(const float* rowdot, const float* norm, const float* inv, int N, int L,
                         float* coefA, float* coefB, void* stream) {
  if (!rowdot || !norm || !inv || !coefA || !coefB || N <= 0 || L <= 0) return VQF_E_BADARG;
  VQF_LAUNCH(KID_L2_BWD_COEF, l2_norm_bwd_coef_kernel, dim3((N + 3) / 4), dim3(256), 0,
             (hipStream_t)stream, rowdot, norm, inv, N, L, coefA, coefB);
  return vqf_last_error();
}

// The same coefficients for the UN-NORMALISED formulation (the fusion output R is handed on without the scale pass; the
// consumer folds 1/norm into its GEMM epilogue and returns dYs = dY / norm):  dR = dYs - coefB * R with
// coefB = inv^2 * sum(R * dYs), and sum(R * dYs) over a sample = sum_{l,g} dlogits * lin (attention.hip, LIN).
// coefA = 1 and unit = 1 (the "inv" the fusion backward multiplies its 0.5 / |Y| by).
int vqf_l2_norm_bwd_coef_lin(const float* dlogits, const float* lin, int G, const float* norm, const float* inv, int N,
                             int L, float* coefA, float* coefB, float* unit, void* stream) {
  if (!dlogits || !lin || !norm || !inv || !coefA || !coefB || !unit || N <= 0 || L <= 0 || G <= 0) return VQF_E_BADARG;
  VQF_LAUNCH(KID_L2_BWD_COEF, l2_norm_bwd_coef_lin_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, dlogits,
             lin, G, norm, inv, N, L, coefA, coefB, unit);
  return vqf_last_error();
}

}  // extern "C"
