/*
 * vqa_fusion.h -- C ABI of libvqa_fusion.so (gfx950 / MI355X only).
 *
 * Drop-in boundary for the attention-fusion hot path of
 * klory/vqa-attention-networks.  The reference has NO native/FFI interface
 * (SURVEY.md section 8b): its boundary is the Python nn.Module forward()
 * signatures.  Each entry point below therefore names the *reference Python
 * lines* whose arithmetic it replaces; the Python host layer
 * (the host/ directory of vqa-attention-networks_amd) keeps the reference class names,
 * constructors, forward() signatures and state_dict keys and calls these
 * functions through ctypes with raw device pointers.  INTEGRATION.md shows
 * the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise;
 *     row-major, innermost dimension contiguous;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream);
 *   - no function allocates, frees or synchronises; scratch memory is passed
 *     in by the caller, so every call is hipGraph-capturable;
 *   - return value: 0 on success, a hipError_t (>0) from the launch, or a
 *     negative VQF_E_* code for argument errors;
 *   - re-entrant: no mutable global state except the opt-in profiler.
 *
 * Shapes use the reference's symbols: N batch, T tokens, L image regions
 * (196), D image channels (2048), H LSTM width, O = 1000 pooled outputs,
 * k = 5 pooling window (mfb.py:42-43,100), so k*O = 5000.
 */
#ifndef VQA_FUSION_H
#define VQA_FUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQF_OK 0
#define VQF_E_BADARG (-1)      /* null pointer / non-positive size           */
#define VQF_E_ALIGN (-2)       /* pointer or leading dimension not 16-B aligned where required */
#define VQF_E_UNSUPPORTED (-3) /* shape outside what the kernel supports     */
#define VQF_E_WORKSPACE (-4)   /* caller's scratch buffer too small          */
#define VQF_E_TIMEOUT (-5)     /* reserved: no entry point of the product build returns it (the whole-sequence LSTM whose in-launch
                                  hand-off could time out left the library in ABI 5: tools/variants/README.md) */

#define VQF_POOL_K 5           /* mfb.py:100  .view(N, L, 1000, 5)           */

/* GEMM epilogue flags */
#define VQF_GEMM_RELU 1        /* C = max(C, 0)                              */
#define VQF_GEMM_ACCUM 2       /* C += result (dgrad accumulation)           */
#define VQF_GEMM_OUT_BF16 4    /* vqf_gemm_bf16 only: C points to bf16 storage (ldc in elements), result rounded
                                  to nearest even; VQF_E_UNSUPPORTED unless the 256x256-tile kernel applies */

/* ABI version (2: LSTM entry points take a workspace and flags, large-tile GEMMs, solver tail, staging; 4: library
 * options, row-scaled GEMM; 5: the whole-sequence LSTM entry points left the library, HBM yardsticks, fused epilogues of
 * the HieCoAtten path; 6: vqf_gate_tanh_sigmoid_*, per-sample-tile GEMM entry points, stated tanh accuracy contract;
 * 7: vqf_hie_affinity, column sums and a partial-row pitch in the vqf_hie_* passes) and
 * build information ("libvqa_fusion gfx950 fp32-mfma(...) tiles ...") */
int vqf_abi_version(void);
const char* vqf_build_info(void);

/* --------------------------------------------------------------------------
 * Library options: process-wide launch policy, an int per option, read by the launchers as a plain load (no
 * getenv on the launch path).  -1 = the library's default.  The initial value of each option is taken ONCE, when
 * the library is loaded, from the environment variable of the same name -- VQF_OPT_X reads VQF_X, see
 * vqf_option_env_name (VQF_GEMM_F32_PERSIST=0 python bench.py still works for command-line A/Bs; the r02 spellings
 * VQF_GEMM_F32_PP / VQF_GEMM_BF16_PP of the two *_LOOP options are still accepted); after that only vqf_set_option changes it.
 * vqf_set_option stores `value` (negative = back to the default) and, if `previous` != NULL, hands back the value
 * it replaced so that a caller can restore it; both return VQF_OK or VQF_E_BADARG (unknown id / NULL out pointer).
 */
#define VQF_OPT_GEMM_F32_PERSIST 0   /* large-tile fp32 GEMM: 1 = persistent workgroups, one per CU (default); 0 = one per tile */
#define VQF_OPT_GEMM_BF16_PERSIST 1  /* the same for the large-tile bf16 GEMM */
#define VQF_OPT_GEMM_F32_LOOP 2      /* fp32 large-tile loop form: 0 lockstep, 1 ping-pong, 2 staggered halves (default) */
#define VQF_OPT_GEMM_BF16_LOOP 3     /* bf16 large-tile loop form: 0 lockstep (r01), 1 ping-pong (default), 3 = 32x32x16 also for (0,0) */
#define VQF_OPT_GEMM_F32_BIG 4       /* 0 = never use the 256x256-tile fp32 kernel */
#define VQF_OPT_GEMM_BF16_BIG 5      /* 0 = never use the 256x256-tile bf16 kernel */
#define VQF_OPT_GEMM_F32_WAVE 6      /* small-M per-wave-tile kernel: 0 = never, 1 = every wave streams its own B slab (r02), 2 = B staged once per workgroup (default) */
#define VQF_OPT_FUSE_COAL 7          /* MFB fusion kernels: 0 = direct (strided) P / dP access everywhere, 1 = LDS-transposed with the forward's register prefetch, default = LDS-transposed, forward without prefetch */
#define VQF_OPT_FUSE_LS 8            /* MFB fusion forward: row splits per sample (>= 1), tuning probe */
#define VQF_OPT_FUSE_LS_BWD 9        /* MFB fusion backward: row splits per sample (1..16), tuning probe */
#define VQF_OPT_GEMM_CU_LIMIT 10     /* persistent large-tile GEMMs use at most this many CUs (multiple of 8; leaves the rest
                                        of the chip to kernels of other streams); <= 0 or -1 = all */
#define VQF_OPT_GEMM_F32_EDGE 11      /* 0 = the large-tile fp32 GEMM treats a short last column tile like a full one (A/B) */
#define VQF_OPT_GEMM_F32_ROUNDS 12    /* 0 = mid-size fp32 GEMMs are NOT split at a whole number of rounds of the large-tile kernel (A/B; see vqf_gemm_f32_big_rows) */
#define VQF_OPT_GEMM_SPLITK_FUSED 13  /* split-K combined INSIDE the GEMM launch (the last-arriving workgroup of an output tile sums the slabs in
                                        split order; same bits as the slabs + vqf_splitk_reduce form): default = in the 256x256-tile kernels
                                        with >= 64 output tiles only; 1 = in the 128x128-tile kernels too (measured slower there); 0 = never */
#define VQF_OPT_GEMM_F32_STREAMK 14   /* 1 = stream-K tail in the large-tile fp32 GEMM: the K slabs of a mid-size product's partial last round are
                                        shared out evenly over the CUs, 2-3 part images per tail tile combined in the launch (opt-in:
                                        measured a wash against the default, the whole-rounds row split; csrc/gemm_f32_big.hip) */
#define VQF_OPT_GEMM_SPLITK_ORDER 15  /* 1 = split-K launches of the large-tile kernels remap work items to XCDs over all (split, tile) items
                                        jointly: the 32 CUs of an XCD take 32 / tiles_n row tiles x ALL column tiles of one split at a time.
                                        Same bits.  Measured (profiles/r05_splitk_order_pmc.txt): the image projection's fp32 weight
                                        gradient moves 6.86 instead of 10.34 GB beyond L2 (bf16: 4.35 / 4.86) and takes 1.4 % LONGER
                                        (14.68 vs 14.48 ms; bf16 2.29 vs 2.27): opt-in, for a fabric that has something else to carry */
#define VQF_OPT_GEMM_F32_SAMPLE 16    /* 0 = vqf_gemm_f32_sample reports every shape unsupported, i.e. HieCoAtten's per-sample products run on the
                                        256x256 / 128x128 kernels as in round 4 (A/B); 2 = it takes every shape it can run, also batches whose
                                        NS * N / 256 work items fill less than half of the CUs (default: those stay on vqf_gemm_f32) */
#define VQF_OPT_GEMM_F32_N80 17       /* 0 = never use the one-round 16x16x4-tile kernels of csrc/gemm_f32_n80.hip: the M = 512 forward
                                        projections run on the 128x128 kernel with split-K + slab reduce and the fused LSTM step on
                                        the per-wave kernel, as before round 5 (A/B) */
#define VQF_OPT_COUNT 18
int vqf_set_option(int option, int value, int* previous);
int vqf_get_option(int option, int* value);
/* the environment variable read for `option` at load time: "VQF_" + the name of its VQF_OPT_* constant ("" if unknown) */
const char* vqf_option_env_name(int option);

/* Launch counters since the library was loaded (which GEMM kernel family a call was routed to: tests and tools use them
 * to prove that a shape reached the kernel they mean to check; the fp32 families give bit-identical results, so the
 * output cannot tell).  VQF_OK or VQF_E_BADARG. */
#define VQF_STAT_GEMM_F32_TILE128 0  /* csrc/gemm_f32.hip, 128x128 tiles (incl. the batched form)  */
#define VQF_STAT_GEMM_F32_BIG 1      /* csrc/gemm_f32_big.hip, 256x256 tiles, LDS-DMA              */
#define VQF_STAT_GEMM_F32_WAVE 2     /* csrc/gemm_f32_wave.hip, one tile per wave (small M)        */
#define VQF_STAT_GEMM_BF16_TILE128 3 /* csrc/gemm_bf16.hip                                         */
#define VQF_STAT_GEMM_BF16_BIG 4     /* csrc/gemm_bf16_big.hip                                     */
#define VQF_STAT_GEMM_F32_SAMPLE 5   /* csrc/gemm_f32_sample.hip, one sample's rows x 256 columns per workgroup */
#define VQF_STAT_GEMM_F32_N80 6      /* csrc/gemm_f32_n80.hip, 128 x 80 tiles, one round, no split-K */
#define VQF_STAT_COUNT 7
int vqf_stat_get(int stat, long long* value);

/* --------------------------------------------------------------------------
 * Dense projections on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32).
 *
 *   C[m,n] (+)= sum_k Aop[m,k] * Bop[n,k]  (+ bias[n])  (relu)
 *   ta = 0: Aop[m,k] = A[m*lda + k]      ta = 1: Aop[m,k] = A[k*lda + m]
 *   tb = 0: Bop[n,k] = B[n*ldb + k]      tb = 1: Bop[n,k] = B[k*ldb + n]
 *
 * Replaces every nn.Linear / 1x1 nn.Conv2d of the path and their autograd:
 *   forward  (ta=0,tb=0): mfb.py:76,79,81,92,96,109,112,114,126-127,137;
 *                         mhb_coAtt.py:81,83,94,98,111,113,124-125,136-137,148;
 *                         hieCoAtten.py:25,30-31,35-36,54
 *   dgrad    (ta=0,tb=1): dX = dY * W
 *   wgrad    (ta=1,tb=1): dW = dY^T * X
 * `ws`/`ws_bytes`: optional split-K scratch (may be NULL/0: no split).
 * Vector (16-byte) loads are used when pointers and leading dimensions allow
 * it; any shape/alignment is accepted.
 * Routing (all deterministic; which kernel ran: vqf_stat_get): the 256x256-tile LDS-DMA kernel for large shapes
 * (vqf_gemm_f32_big_rows), the per-wave-tile kernel for the recurrent small-M products, the one-round 128x80-tile kernel
 * (csrc/gemm_f32_n80.hip, round 5) for forward products whose 128x80 tiles fill 80..100 % of the CUs once -- M <= 1024,
 * K % 128 == 0, K >= 512, no VQF_GEMM_ACCUM: the 512 x 5000 projections of mfb.py:76,92,126,127 at batch 512, with or
 * without `ws` -- and the 128x128-tile kernel (split-K over `ws` + slab reduce) for everything else.  The 128x80 kernel adds
 * k in a different order (v_mfma_f32_16x16x4_f32) than the others: same value to fp32 rounding, not the same bits.
 */
/* Scratch vqf_gemm_f32 can use for this shape (deterministic split-K slabs); with less it picks fewer
 * splits.  Never more than 16 * M * N * 4 bytes. */
size_t vqf_gemm_f32_ws_bytes(int ta, int tb, int M, int N, int K);
int vqf_gemm_f32(int ta, int tb, int M, int N, int K,
                 const float* A, int lda, const float* B, int ldb,
                 float* C, int ldc, const float* bias, int flags,
                 void* ws, size_t ws_bytes, void* stream);

/* The same product with a per-row-group scale in the epilogue:
 *   C[m,n] = relu?( rowscale[m / rows_per_scale] * sum_k Aop[m,k] Bop[n,k] + bias[n] )
 * -- the co-attention conv applied to the UN-NORMALISED fusion output R of a sample (mfb.py:105-109: F.normalize, then
 * co_att_conv1): rowscale = 1 / max(||R_n||, eps), rows_per_scale = L, so the normalised tensor is never written.
 * No split-K, no VQF_GEMM_ACCUM; routed like vqf_gemm_f32 (see vqf_gemm_f32_big_rows). */
int vqf_gemm_f32_rowscale(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, const float* bias, int flags, const float* rowscale, int rows_per_scale,
                          void* stream);

/* How vqf_gemm_f32 / vqf_gemm_f32_rowscale route a product (for tools and bench.py's per-launch lookups): the number of
 * leading rows that run on the 256x256-tile kernel -- M (all), 0 (none: 128x128 / per-wave kernels), or, for a mid-size
 * shape whose tile count is not a whole number of rounds of the chip's CUs, the largest row block that is (a multiple
 * of 256 rows); the remaining M - rows rows are a second launch on the other kernels.  Depends on the library options
 * gemm_f32_big / gemm_f32_rounds and the device's CU count only. */
int vqf_gemm_f32_big_rows(int ta, int tb, int M, int N, int K);
/* Per-sample-tile product (csrc/gemm_f32_sample.hip; hieCoAtten.py:25,30,35 and their input gradients at BASELINE config 4):
 * C (NS*L, N) = A (NS*L, K) * Bop^T (+ bias) (VQF_GEMM_RELU), A row-major with the rows of sample n at n*L .. n*L + L - 1,
 * B (N, K) (tb = 0) or (K, N) (tb = 1).  One workgroup owns a sample's L rows x 256 columns: NS * N / 256 work items, a whole
 * number of rounds of the CUs for NS = 256 where 256x256 tiles leave 1.53; L = 192 + 4e rows = six 32-row MFMA tiles + e <= 1
 * four-row groups on v_mfma_f32_4x4x1_16B_f32 (no padded rows).  Same bits as vqf_gemm_f32 on the same operands.
 * Supported (vqf_gemm_f32_sample_supported): L = 192 or 196, N % 256 == 0, K % 16 == 0, K >= 64; only the
 * VQF_GEMM_RELU flag, and (unless option gemm_f32_sample = 2) NS * N / 256 >= half the CU count; else VQF_E_UNSUPPORTED (the
 * caller uses vqf_gemm_f32). */
int vqf_gemm_f32_sample_supported(int NS, int L, int N, int K);
int vqf_gemm_f32_sample(int tb, int NS, int L, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                        const float* bias, int flags, void* stream);

/* Batched form: for b < batch, C_b = Aop_b * Bop_b^T with A_b = A + b*strideA etc.
 * (element strides).  No bias, no split-K.  Per-sample products of
 * hieCoAtten.py:32,38,41,45,48 and modules.py:65,91,94. */
int vqf_gemm_f32_batched(int ta, int tb, int batch, int M, int N, int K,
                         const float* A, int lda, long long strideA,
                         const float* B, int ldb, long long strideB,
                         float* C, int ldc, long long strideC, int flags, void* stream);

/* bf16 x bf16 -> fp32 form (v_mfma_f32_32x32x16_bf16, fp32 accumulate; BASELINE config 3: bf16
 * storage of the image tensor / activations / weights of the large projections).  A and B point
 * to bf16 (uint16) data, same ta/tb meaning as vqf_gemm_f32; C, bias fp32.  Returns
 * VQF_E_UNSUPPORTED unless: bases 16-byte aligned, lda/ldb % 8 == 0, K % 8 == 0 for a
 * K-contiguous operand, row extent % 8 == 0 (and >= 8) for a K-major operand. */
/* Scratch vqf_gemm_bf16 can use for this shape (deterministic split-K slabs of its 256x256-tile path,
 * csrc/gemm_bf16_big.hip); with less (or none) it picks fewer splits or the 128x128 kernel. */
size_t vqf_gemm_bf16_ws_bytes(int ta, int tb, int M, int N, int K);
int vqf_gemm_bf16(int ta, int tb, int M, int N, int K,
                  const void* A, int lda, const void* B, int ldb,
                  float* C, int ldc, const float* bias, int flags,
                  void* ws, size_t ws_bytes, void* stream);
/* the bf16 form of vqf_gemm_f32_rowscale: C = relu?( rowscale[m / rows_per_scale] * sum_k Aop Bop + bias ), fp32 C; no
 * split-K, no VQF_GEMM_ACCUM / VQF_GEMM_OUT_BF16 (bf16 mode of the co-attention conv on the un-normalised fusion output) */
int vqf_gemm_bf16_rowscale(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                           float* C, int ldc, const float* bias, int flags, const float* rowscale, int rows_per_scale,
                           void* stream);

/* y (R x ldy, bf16) = round-to-nearest-even(x (R x C, fp32)), columns C..ldy-1 zero-filled
 * (padding K up to a multiple of 8/32 for vqf_gemm_bf16).  ldy % 8 == 0. */
int vqf_cast_f32_bf16(const float* x, int R, int C, int ldx, void* y, int ldy, void* stream);

/* db[n] = sum_m dY[m,n]   (bias gradients of the Linear/Conv layers).
 * ws: scratch of at least vqf_colsum_ws_bytes(M,N) bytes. */
size_t vqf_colsum_ws_bytes(int M, int N);
int vqf_colsum_f32(const float* dY, int M, int N, int ldy, float* db,
                   void* ws, size_t ws_bytes, void* stream);

/* out[g,c] = sum_{j<J} in[(g*J + j), c]   (partial-slab reducer) */
int vqf_group_reduce_f32(const float* in, int G, int J, int W, float* out, void* stream);

/* dXpre = dX * (Y > 0); optional dbias[c] = sum_m dXpre[m,c].
 * ReLU backward of the "multilayer" attention MLPs (mfb.py:78-80,111-113).
 * ws as for vqf_colsum_f32 (only needed when dbias != NULL). */
int vqf_relu_bwd_f32(const float* dX, const float* Y, int M, int C, float* dXpre,
                     float* dbias, void* ws, size_t ws_bytes, void* stream);

/* Backward of Y = dropout(relu(pre)) (hieCoAtten.py:25-26) whose output is also pooled by an attention head (:41): the head's
 * gradient into Y is the rank-1 term wts[m] * dpooled[m / L, :], added here instead of being materialised:
 *   dXpre[m,c] = (dX[m,c] + wts[m] * dpooled[m / L, c]) * (Y[m,c] > 0 ? scale : 0),  scale = 1 / (1 - p)
 * (Y > 0 <=> pre > 0 and the element was kept); wts == NULL: no rank-1 term; dbias / ws as vqf_relu_bwd_f32; in place
 * (dXpre == dX) allowed; C % 4 == 0. */
int vqf_relu_bwd_rank1_f32(const float* dX, const float* Y, const float* wts, const float* dpooled, int L, float scale, int M,
                           int C, float* dXpre, float* dbias, void* ws, size_t ws_bytes, void* stream);
/* out[k][i] = a[k][i] + b[k][i], i < n[k], for count <= 4 segments in ONE launch; dst[k][i] = src[k][i] for count <= 8
 * segments in one launch (HOST arrays of device pointers): packing the weights of layers that share an input into one GEMM
 * operand and summing the gradient halves back. */
int vqf_multi_add_f32(const float* const* a, const float* const* b, float* const* out, const long long* n, int count, void* stream);
int vqf_multi_copy_f32(const float* const* src, float* const* dst, const long long* n, int count, void* stream);

/* --------------------------------------------------------------------------
 * Attention heads: hidden -> 2 glimpse logits, softmax, glimpse-weighted sum.
 * Question side (S = T):  mfb.py:81-89   / mhb_coAtt.py:83-91
 * Image side    (S = L):  mfb.py:114-123 / mhb_coAtt.py:113-121
 */

/* logits[m,g] = hid[m,:] . w2[g,:] + b2[g],  g < G (G = 2 glimpses: mfb.py / mhb_coAtt.py;
 * G = 1: hieCoAtten.py:40,47 fc_Whv/fc_Whq, modules.py:60 Attention_1.fc); hid is (M,Hh),
 * w2 (G,Hh), logits (M,G). */
int vqf_att_logits_fwd(const float* hid, const float* w2, const float* b2,
                       int M, int Hh, int G, float* logits, void* stream);

/* vqf_att_logits_fwd for a hidden layer hid = relu(pre + b1) whose `pre` is LINEAR in the layer's input: additionally
 * lin[m,g] = sum_{j: hid[m,j] > 0} w2[g,j] * (hid[m,j] - b1[j]), the part of the logit linear in that input.  With it
 * sum(Y * dY) of F.normalize's backward is sum_g dlogits[m,g] * lin[m,g] (vqf_l2_norm_bwd_coef_lin): no pass over Y, dY. */
int vqf_att_logits_fwd_lin(const float* hid, const float* w2, const float* b2, const float* b1, int M, int Hh, int G,
                           float* logits, float* lin, void* stream);

/* Backward of the G-logit head; relu_mask != 0: THROUGH the ReLU that produced hid:
 *   dhid_pre[m,j] = (sum_g dl[m,g] w2[g,j]) * (hid[m,j] > 0  or 1)
 *   dw2[g,j] = sum_m dl[m,g] hid[m,j];  db2[g] = sum_m dl[m,g]
 *   dbias1[j] = sum_m dhid_pre[m,j]          (bias grad of the layer before)
 * ws: at least vqf_att_logits_bwd_ws_bytes(M,Hh). */
size_t vqf_att_logits_bwd_ws_bytes(int M, int Hh);
int vqf_att_logits_bwd(const float* dlogits, const float* hid, const float* w2,
                       int M, int Hh, int G, int relu_mask, float* dhid_pre, float* dw2,
                       float* db2, float* dbias1, void* ws, size_t ws_bytes, void* stream);
/* Same, with the STORED dhid_pre rows multiplied by rowscale[m / rows_per_scale] (NULL: 1); dbias1 sums the unscaled
 * values.  For a layer fed by vqf_gemm_f32_rowscale: dW1 = dhid_pre^T R and dR = dhid_pre W1 then need no scaling. */
int vqf_att_logits_bwd_rowscale(const float* dlogits, const float* hid, const float* w2, const float* rowscale,
                                int rows_per_scale, int M, int Hh, int G, int relu_mask, float* dhid_pre, float* dw2,
                                float* db2, float* dbias1, void* ws, size_t ws_bytes, void* stream);
/* The same with the stored rows in bf16 (round-to-nearest-even; dhid_pre_bf16: (M, Hh) bf16, 8-byte aligned, Hh % 4 == 0),
 * through the ReLU, G = 2: the A operand of the layer's bf16 weight- and input-gradient products (BASELINE config 3,
 * mhb_coAtt.py:97-98 co_att_conv1) without the fp32 round trip + vqf_cast_f32_bf16 launch (ABI 7).  Sums stay fp32. */
int vqf_att_logits_bwd_rowscale_obf16(const float* dlogits, const float* hid, const float* w2, const float* rowscale,
                                      int rows_per_scale, int M, int Hh, int G, void* dhid_pre_bf16, float* dw2, float* db2,
                                      float* dbias1, void* ws, size_t ws_bytes, void* stream);

/* wts[n,g,:] = softmax_s(logits[n,:,g])   (unit_softmax != 0: wts == 1, the
 * mfb.py:84,118 singleton-axis softmax);  pooled[n, g*C + c] = sum_s wts[n,g,s] feat[n,s,c].
 * feat (N,S,C), logits (N*S,G), wts (N,G,S), pooled (N,G*C).  S <= 1024, G in {1,2}. */
int vqf_glimpse_pool_fwd(const float* feat, const float* logits, int N, int S, int C, int G,
                         int unit_softmax, float* wts, float* pooled, void* stream);

/* Given dpooled (N,G*C) and, optionally, dwts_extra (N,G,S) = gradient arriving through the
 * returned attention weights (hieCoAtten.py:55 returns av/aq; networks.py:64-66 feeds them to
 * fc): dlogits (N*S,G) through the softmax (all zeros when unit_softmax), and, if dfeat != NULL,
 * dfeat[n,s,c] = sum_g wts[n,g,s] dpooled[n,gC+c] (overwrites; the image is data). */
int vqf_glimpse_pool_bwd(const float* dpooled, const float* dwts_extra, const float* feat,
                         const float* wts, int N, int S, int C, int G, int unit_softmax,
                         float* dlogits, float* dfeat, void* stream);

/* The same two stages over a bf16 feature tensor (bf16 storage of the image grid, BASELINE config 3;
 * produced by vqf_feat_transpose(out_bf16=1) or vqf_cast_f32_bf16): products and sums in fp32.
 * The feature tensor is data, so there is no dfeat.  C % 4 == 0 for the vector path. */
int vqf_glimpse_pool_fwd_bf16(const void* feat, const float* logits, int N, int S, int C, int G,
                              int unit_softmax, float* wts, float* pooled, void* stream);
int vqf_glimpse_pool_bwd_bf16(const float* dpooled, const float* dwts_extra, const void* feat,
                              const float* wts, int N, int S, int C, int G, int unit_softmax,
                              float* dlogits, void* stream);

/* --------------------------------------------------------------------------
 * MFB fusion: product, dropout, k=5 sum-pool, signed sqrt, L2 normalise.
 *   mfb.py:98-106 (L = 196 regions) and mfb.py:128-135 (L = 1, final block);
 *   mhb_coAtt.py:100-108,126-145.
 *
 *  P    (N*L, 5*O)  projected image features
 *  pbias (5*O) or NULL: bias of that projection, added on load (P then comes WITHOUT bias; lets the
 *                   projection GEMM run on its own stream and keeps the bias gradient in this stage)
 *  q    (N,   5*O)  projected question, broadcast over the L rows of a sample
 *  cascade (N*L, 5*O) or NULL: third factor of MHB's high-order block
 *                   (mhb_coAtt.py:205, the dropped-out first-order product)
 *  keep (N*L, 5*O)  uint8 keep-mask or NULL.  NULL and p_drop > 0: the mask is
 *                   generated in-kernel from Philox4x32-10(seed, element index / 8), one 16-bit draw
 *                   per element (keep iff draw >= p * 65536), identically in forward and backward.
 *  R    (N*L, O)    signed sqrt of the pooled sums (un-normalised)
 *  rowssq (N*L*4)   per-row sum of R^2 (= sum |pooled|) as FOUR partial sums per row (one per wave of the workgroup: no barrier
 *                   in the row loop); vqf_l2_group_norm(rowssq, N, 4 * L, ...) adds them up
 *  zdrop (N*L,5*O) or NULL: the dropped-out product itself (only MHB needs it)
 *  O % 4 == 0 and O <= 1024 (mfb.py:42-43 hard-codes O = 1000); else VQF_E_UNSUPPORTED.
 */
int vqf_mfb_fuse_fwd(const float* P, const float* pbias, const float* q, const float* cascade,
                     const uint8_t* keep, uint64_t seed, float p_drop,
                     int N, int L, int O, float* R, float* rowssq, float* zdrop,
                     void* stream);

/* norm[n] = sqrt(sum_l rowssq[n*L+l]);  inv[n] = 1 / max(norm[n], 1e-12)  (F.normalize eps) */
int vqf_l2_group_norm(const float* rowssq, int N, int L, float* norm, float* inv, void* stream);

/* Y[m,:] = R[m,:] * inv[m / L]   (in place allowed: Y == R) */
int vqf_scale_rows(const float* R, const float* inv, int M, int L, int W, float* Y, void* stream);

/* rowdot[m] = sum_o Y[m,o] * dY[m,o] */
int vqf_rowdot(const float* Y, const float* dY, int M, int W, float* rowdot, void* stream);

/* Coefficients of the F.normalize backward per sample:
 *   dR = coefA[n] * dY - coefB[n] * Y,
 *   coefA = inv, coefB = inv * sum_l rowdot   (coefB = 0 when norm <= eps: clamped branch) */
int vqf_l2_norm_bwd_coef(const float* rowdot, const float* norm, const float* inv,
                         int N, int L, float* coefA, float* coefB, void* stream);
/* The un-normalised formulation (R handed on without vqf_scale_rows; the consumer scales in its GEMM epilogue and returns
 * dYs = dY / norm):  dR = dYs - coefB R,  coefB[n] = inv[n]^2 * sum_{l,g} dlogits lin  (0 in the clamped branch),
 * coefA[n] = unit[n] = 1: call vqf_mfb_fuse_bwd with dY := dYs, Y := R, inv := unit. */
int vqf_l2_norm_bwd_coef_lin(const float* dlogits, const float* lin, int G, const float* norm, const float* inv, int N,
                             int L, float* coefA, float* coefB, float* unit, void* stream);

/* Backward of vqf_mfb_fuse_fwd given dY (N*L,O) w.r.t. the NORMALISED output Y
 * and, optionally, dzdrop (N*L,5*O) w.r.t. the zdrop output (NULL: none):
 *   dS = (coefA dY - coefB Y) * 0.5 * inv / |Y|     (0 where Y == 0: relu'(0) = 0)
 *   dz[c] = (dS[c/5] + dzdrop[c]) * keep[c] / (1-p)
 *   dP[m,c] = dz * q[n,c] (* cascade),  dq[n,c] = sum_l dz * P[m,c] (* cascade)
 *   dcascade[m,c] = dz * P * q          (if cascade != NULL)
 *   dbias_P[c] = sum_m dP[m,c]          (if dbiasP != NULL; this is d pbias as well)
 * P is taken as P + pbias wherever it appears when pbias != NULL.
 * ws: at least vqf_mfb_fuse_bwd_ws_bytes(N,L,O). */
size_t vqf_mfb_fuse_bwd_ws_bytes(int N, int L, int O);
int vqf_mfb_fuse_bwd(const float* dY, const float* dzdrop, const float* Y, const float* inv,
                     const float* coefA, const float* coefB,
                     const float* P, const float* pbias, const float* q, const float* cascade,
                     const uint8_t* keep, uint64_t seed, float p_drop,
                     int N, int L, int O,
                     float* dP, float* dq, float* dcascade, float* dbiasP,
                     void* ws, size_t ws_bytes, void* stream);
/* Same, for the image fusion in bf16 mode (no cascade, no dzdrop): dP is written as bf16
 * (round-to-nearest-even), ready to be the A operand of the weight-gradient vqf_gemm_bf16 -- saves
 * the 2 GB fp32 round trip and the cast pass.  dq / dbiasP stay fp32 and are accumulated from the
 * unrounded values. */
int vqf_mfb_fuse_bwd_bf16dp(const float* dY, const float* Y, const float* inv, const float* coefA,
                            const float* coefB, const float* P, const float* pbias, const float* q,
                            const uint8_t* keep, uint64_t seed, float p_drop, int N, int L, int O,
                            void* dP_bf16, float* dq, float* dbiasP, void* ws, size_t ws_bytes, void* stream);
/* The image fusion with the projection itself stored in bf16 (P written by vqf_gemm_bf16 with
 * VQF_GEMM_OUT_BF16: half the bytes of the largest tensor of the step in all three passes over it). */
int vqf_mfb_fuse_fwd_pbf16(const void* P_bf16, const float* pbias, const float* q, const uint8_t* keep, uint64_t seed,
                           float p_drop, int N, int L, int O, float* R, float* rowssq, void* stream);
/* The same, leaving ALSO a bf16 copy of R (round-to-nearest-even of the fp32 values; row pitch ldrb >= O elements, ldrb % 4 == 0,
 * ldrb <= 1024, 8-byte aligned; columns O .. ldrb-1 zero): the K-padded A operand of the co-attention conv's bf16 GEMM
 * (mhb_coAtt.py:97-98) without a vqf_cast_f32_bf16 pass over R; the fp32 R stays for the backward (ABI 7). */
int vqf_mfb_fuse_fwd_pbf16_rb(const void* P_bf16, const float* pbias, const float* q, const uint8_t* keep, uint64_t seed,
                              float p_drop, int N, int L, int O, float* R, void* R_bf16, int ldrb, float* rowssq, void* stream);
int vqf_mfb_fuse_bwd_pbf16(const float* dY, const float* Y, const float* inv, const float* coefA,
                           const float* coefB, const void* P_bf16, const float* pbias, const float* q,
                           const uint8_t* keep, uint64_t seed, float p_drop, int N, int L, int O, void* dP_bf16,
                           float* dq, float* dbiasP, void* ws, size_t ws_bytes, void* stream);



/* --------------------------------------------------------------------------
 * Element-wise stages of HieCoAtten / AttentionNet.  n % 4 == 0, 16-byte aligned pointers.
 * keep: uint8 keep-mask or NULL (then Philox4x32-10(seed, index/4), one 32-bit draw per element).
 */
/* y = x * keep / (1-p)      F.dropout, hieCoAtten.py:26,28; networks.py:22,24,55,57.
 * Its own backward: call it with x = dy and the same keep/seed. */
int vqf_dropout_f32(const float* x, const uint8_t* keep, uint64_t seed, float p_drop,
                    long long n, float* y, void* stream);
/* The LSTM-output dropout (mfb.py:70, mhb_coAtt.py:75) over a (B, T, H) tensor whose first two axes carry free ELEMENT strides on
 * the input and on the output side (last axis contiguous): y[b,t,:] = x[b,t,:] * keep / (1-p).  One pass also re-lays the
 * recursion's time-major states (T, B, H) out as the sample-major rows (B, T, H) the attention head reads; its backward is the
 * same call with the strides swapped.  Dropout index of (b, t, h): (b*T + t)*H + h (one Philox call per 4 elements as above;
 * keep: (B*T, H) uint8).  H and every stride % 4 == 0. */
int vqf_dropout_bt(const float* x, long long sb_in, long long st_in, const uint8_t* keep, uint64_t seed, float p_drop, int B,
                   int T, int H, float* y, long long sb_out, long long st_out, void* stream);
/* The gate of modules.py:103-109 (Nonlinear_layer.forward): y = tanh(a) * sigmoid(b) over n elements (n % 4 == 0, 16-byte aligned
 * pointers), and its backward from the saved inputs: da = dy * sigmoid(b) * (1 - tanh(a)^2), db = dy * tanh(a) * sigmoid(b) *
 * (1 - sigmoid(b)).  libm tanhf / expf (1-2 ulp). */
int vqf_gate_tanh_sigmoid_fwd(const float* a, const float* b, long long n, float* y, void* stream);
int vqf_gate_tanh_sigmoid_bwd(const float* dy, const float* a, const float* b, long long n, float* da, float* db, void* stream);
/* y = dropout(tanh(a + b))  (b may be NULL)   hieCoAtten.py:32-33,38-39,45-46.
 * Accuracy contract of the tanh in THIS entry point, in vqf_tanh_dropout_fwd2d and in vqf_hie_hv_fwd: (e^2x - 1) / (e^2x + 1) with
 * the hardware exponential and reciprocal -- ABSOLUTE error <= 2e-7 everywhere, i.e. relative accuracy is lost for |x| < 1e-3
 * (6 % at x = 1e-6).  Their consumers are bounded activations in front of a softmax (hieCoAtten.py:32-46); the entry points a
 * chain of steps feeds on (vqf_lstm_*, vqf_embed_tanh_*, vqf_gate_tanh_sigmoid_*) use libm's tanhf (1 ulp relative). */
int vqf_tanh_dropout_fwd(const float* a, const float* b, const uint8_t* keep, uint64_t seed,
                         float p_drop, long long n, float* y, void* stream);
/* dx = dy * keep/(1-p) * (1 - tanh^2), tanh recovered from the saved output y */
int vqf_tanh_dropout_bwd(const float* dy, const float* y, const uint8_t* keep, uint64_t seed,
                         float p_drop, long long n, float* dx, void* stream);
/* The tanh stages over 2-D operands with row strides lda / ldb / ldy (column blocks of wider buffers: HieCoAtten applies fc_Wbv
 * and fc_Wv to the same input, hieCoAtten.py:30,35, as ONE product with the concatenated weights; its halves are consumed in
 * place).  The dropout index of element (r, c) is r * W + c -- the bits of the flat call on the contiguous (R, W) tensor.
 * W and every stride % 4 == 0. */
int vqf_tanh_dropout_fwd2d(const float* a, int lda, const float* b, int ldb, const uint8_t* keep, uint64_t seed, float p_drop,
                           int R, int W, float* y, int ldy, void* stream);
int vqf_tanh_dropout_bwd2d(const float* dy, int lddy, const float* y, int ldy, const uint8_t* keep, uint64_t seed, float p_drop,
                           int R, int W, float* dx, int lddx, void* stream);
/* --------------------------------------------------------------------------
 * HieCoAtten's co-attention ladder (hieCoAtten.py:32-49): the stages with a tiny inner / outer extent T (words per question)
 * as single streaming passes over the rows of the (N*L, E) tensors (csrc/hie.hip), neighbouring element-wise stage fused in.
 * Rows m = n*L + l of a / z / out may be strided (lda / ldz / ldo; column blocks of the concatenated-weight products);
 * C, U: (N, T, L) contiguous; V: rows n*T + t, stride ldv.  part: the T-row sums of a sample.  S = vqf_hie_chunks(N, L) row
 * chunks per sample (1 once N reaches the CU count: one 1024-thread workgroup per sample): with S > 1 `part` is (S, N*T, E)
 * contiguous partial slabs (ldp == E, padd == NULL) to be added up by vqf_hie_slab_sum (fixed order, no atomics); with S == 1
 * the sums are FINAL and go to rows n*T + t of pitch ldp, on top of padd's rows (pitch ldpa) when padd != NULL -- no slab-sum
 * launch (ABI 6).  Supported (vqf_hie_stream_supported):
 * T <= 16, E % 4 == 0, E / 4 divides 256, T * E small enough for LDS; else VQF_E_UNSUPPORTED (the caller uses the batched GEMMs).
 *   vqf_hie_hv_fwd     out = dropout(tanh(a + C^T V)) (:38-39, Hv; a = img_, V = que_);  part[t] += C[t,l] a[l]  (:45, ti = C img_)
 *   vqf_hie_head_bwd   out = dl[l] w sc (1 - (hv/sc)^2): gradient of :38-40 w.r.t. img_ + tq from the logit gradient dl (N*L) of
 *                      fc_Whv (weight w (E)), dHv never materialised;  part[t] += C[t,l] out[l]  (-> dque_);  wpart: S*N
 *                      partial rows of pitch ldw >= E+4 (ABI 7; a column block of a wider buffer), each
 *                      [sum_l dl[l] hv[l,:] | sum_l dl[l] | 0 0 0]  (column sums -> d fc_Whv.weight, d fc_Whv.bias)
 *   vqf_hie_rank_add   out = a + U^T V                 (dimg_ += C^T dti; in place allowed)
 *   vqf_hie_rank_left  out = U^T V;  part[t] += U[t,l] z[l]      (dCv = daff^T Cq;  dCq = daff Cv)
 *                      both (ABI 7): colpart != NULL -> row s*N + n of colpart (pitch ldcp) = the column sums of the rows of
 *                      `out` this workgroup wrote: summed over the S*N rows they are the bias gradients of fc_Wbv / fc_Wv
 *                      (autograd of hieCoAtten.py:30,35) without a column-sum pass over the (N*L, 2E) gradient buffer
 *   vqf_hie_slab_sum   out[r,:] = (add ? add[r,:] : 0) + sum_s part[s][r][:],  r < R, W columns */
int vqf_hie_stream_supported(int N, int L, int E, int T);
int vqf_hie_chunks(int N, int L);
int vqf_hie_hv_fwd(const float* a, int lda, const float* C, const float* V, int ldv, const uint8_t* keep, uint64_t seed,
                   float p_drop, int N, int L, int E, int T, float* out, int ldo, float* part, int ldp, void* stream);
int vqf_hie_head_bwd(const float* hv, int ldh, const float* dl, const float* w, const float* C, const uint8_t* keep,
                     uint64_t seed, float p_drop, int N, int L, int E, int T, float* out, int ldo, float* part, int ldp,
                     const float* padd, int ldpa, float* wpart, int ldw, void* stream);
int vqf_hie_rank_add(const float* a, int lda, const float* U, const float* V, int ldv, int N, int L, int E, int T, float* out,
                     int ldo, float* colpart, int ldcp, void* stream);
int vqf_hie_rank_left(const float* U, const float* V, int ldv, const float* z, int ldz, int N, int L, int E, int T, float* out,
                      int ldo, float* part, int ldp, float* colpart, int ldcp, void* stream);
int vqf_hie_slab_sum(const float* part, int S, int R, int W, const float* add, int lda, float* out, int ldo, void* stream);
/* The ladder's inner-product stage (ABI 7): out[n,t,l] = epi( sum_e x1[n*T+t, e] y1[n*L+l, e]  [+ sum_e x2[..] y2[..]] ), out
 * (N, T, L) contiguous; x*, y*: rows with strides ldx*, ldy* (column blocks of the concatenated-weight products), 16-byte
 * aligned.  Replaces the batched 14-row GEMMs of hieCoAtten.py:32 (C = tanh(Cq Cv^T), functions.HieCoreFn) and of its gradient
 * (dC = dti img_^T + que_ dtq^T: the second pair) with one pass over the y rows on v_mfma_f32_16x16x4_f32 (fp32 exact
 * products, one k-ordered accumulation per output; the order differs from vqf_gemm_f32_batched's: same value to fp32 rounding).
 * epi 0: the sums; 1: dropout(tanh(.)) -- mask and tanh exactly those of vqf_tanh_dropout_fwd on the contiguous (N*T, L)
 * tensor (keep / seed / p_drop as there); 2: its backward, out = sums * sc * (1 - (yprev / sc)^2) with yprev the forward's
 * output (vqf_tanh_dropout_bwd).  x2 == y2 == NULL: one pair.  Supported: T <= 16, E % 32 == 0, pairs * 16 * (E + 4) floats
 * of LDS <= 160 KB; any N <= 65535, any L. */
int vqf_hie_affinity_supported(int N, int L, int E, int T, int pairs);
int vqf_hie_affinity(const float* x1, int ldx1, const float* y1, int ldy1, const float* x2, int ldx2, const float* y2, int ldy2,
                     int epi, const float* yprev, const uint8_t* keep, uint64_t seed, float p_drop, int N, int L, int E, int T,
                     float* out, void* stream);

/* softmax over the last axis of (R,W) and its backward   modules.py:91-92 */
int vqf_softmax_rows_fwd(const float* x, int R, int W, float* y, void* stream);
int vqf_softmax_rows_bwd(const float* dy, const float* y, int R, int W, float* dx, void* stream);

/* log_softmax over the last axis of a (R, W) tensor and its gradient: the classifier tail of MHBCoAtt / MHB
 * (mhb_coAtt.py:149-151, :215-217; implicit dim = 1 on the 2-D logits).  y = (x - max) - log(sum exp(x - max));
 * dx = dy - exp(y) * sum_c dy. */
int vqf_log_softmax_rows_fwd(const float* x, int R, int W, float* y, void* stream);
int vqf_log_softmax_rows_bwd(const float* dy, const float* y, int R, int W, float* dx, void* stream);

/* --------------------------------------------------------------------------
 * LSTM recursion of the question encoder for small per-step batches:
 * mhb_coAtt.py:27-36,72-74 recurs over the minibatch axis (S = N steps of a T-row batch).
 * Single layer, zero initial state, PyTorch gate order i,f,g,o.  One kernel launch per step.
 *   xw     (S,B,4H)  x W_ih^T + b_ih + b_hh   (computed with vqf_gemm_f32)
 *   w_hh   (4H,H)    lstm.weight_hh_l0 (both directions take it as stored; each call packs the
 *                    operand image it needs into ws)
 *   hs, cs (S,B,H)   hidden / cell states;  gates (S,B,4H) ACTIVATED i,f,g,o (saved for backward)
 *   dhs    (S,B,H)   dL/dh_s from the consumers;  dgates (S,B,4H) dL/d(pre-activation): then
 *                    dW_hh = dgates[1:]^T hs[:-1], dW_ih = dgates^T x, db = colsum(dgates), dx = dgates W_ih
 *   dc_carry (B,H)   scratch;  ws: vqf_lstm_seq_ws_bytes(B,H) bytes of scratch (16-byte aligned)
 * Supported: B <= 32, H in {256,512,768,1024} (vqf_lstm_seq_supported); else VQF_E_UNSUPPORTED. */
int vqf_lstm_seq_supported(int B, int H);
size_t vqf_lstm_seq_ws_bytes(int B, int H);
#define VQF_LSTM_BF16 1   /* flags: W_hh and the recurrent operand (h / dG) enter the MFMA as bf16 (bf16 mode);
                             accumulation, gates, cell state and all stored tensors stay fp32 */
int vqf_lstm_seq_fwd(const float* xw, const float* w_hh, int S, int B, int H,
                     float* hs, float* cs, float* gates, int flags, void* ws, size_t ws_bytes, void* stream);
int vqf_lstm_seq_bwd(const float* dhs, const float* gates, const float* cs, const float* w_hh,
                     int S, int B, int H, float* dgates, float* dc_carry, int flags, void* ws, size_t ws_bytes,
                     void* stream);

/* Point-wise cell stages for LARGE per-step batches (the question encoder in its regular orientation,
 * mfb.py:68-70: T = 14 steps of an N-row batch; MHB, mhb_coAtt.py:182-183): the recurrent product is
 * a vqf_gemm_f32 with VQF_GEMM_ACCUM into the input projection, these do the rest of a step in one pass.
 *   fwd: gates (B,4H) pre-activations in -> ACTIVATED i,f,g,o out (in place);  c_out = f c_prev + i g;
 *        h_out = o tanh(c_out).  c_prev = NULL at t = 0.
 *   bwd: dh = dhs_t (+ dh_carry);  dG (B,4H) = pre-activation gradients;  dc_carry updated in place
 *        (first != 0 at the last time step: the carry is not read).  c_prev = NULL at t = 0. */
int vqf_lstm_cell_fwd(float* gates, const float* c_prev, int B, int H, float* c_out, float* h_out, void* stream);
/* One whole forward step in ONE launch: gates (B,4H) += h_prev (B,H) W_hh^T, then the cell stage above in the product's epilogue
 * (the wave tile of csrc/gemm_f32_wave.hip holds 16 hidden units x 4 gates: W_hh's rows are gathered gate-interleaved by the
 * LDS-DMA source addresses).  Two forms.  Where B / 128 x H / 16 tiles fill 80..100 % of the CUs once and H % 128 == 0, H >= 512
 * (the question encoder at batch 512, H = 1024: 256 tiles) one workgroup per CU multiplies 128 rows x 16 units on
 * v_mfma_f32_16x16x4_f32 with the four column tiles = the four gates of a unit, so the cell is a per-lane epilogue
 * (csrc/gemm_f32_n80.hip, round 5: 45-47 us against 60-63; same cell arithmetic, the product's k order is that kernel's);
 * otherwise, or with option gemm_f32_n80 = 0, the per-wave form: bit-identical to vqf_gemm_f32(VQF_GEMM_ACCUM) +
 * vqf_lstm_cell_fwd.  Supported (vqf_lstm_step_supported): B % 128 == 0, H % 16 == 0, H >= 256; else VQF_E_UNSUPPORTED. */
int vqf_lstm_step_supported(int B, int H);
int vqf_lstm_step_fwd(const float* h_prev, const float* w_hh, float* gates, const float* c_prev, int B, int H,
                      float* c_out, float* h_out, void* stream);
int vqf_lstm_cell_bwd(const float* dhs_t, const float* dh_carry, const float* gates, const float* c_t,
                      const float* c_prev, int first, int B, int H, float* dc_carry, float* dG, void* stream);

/* --------------------------------------------------------------------------
 * Question-encoder front end: e = tanh(Embedding(q))   (mfb.py:68, mhb_coAtt.py:69).
 *   W (V,E) word_embedding.weight, ids (T) int64 token ids (the (N,T) question tensor, flattened), out (T,E).
 *   bwd: dW (V,E) = sum over the tokens of each id of dout * (1 - out^2), EVERY row written (zeros where an id does not
 *   occur), summed in token order (deterministic, no atomics).  E <= 1024.  Ids outside [0,V) select no row. */
int vqf_embed_tanh_fwd(const float* W, const long long* ids, int T, int V, int E, float* out, void* stream);
int vqf_embed_tanh_bwd(const float* dout, const float* out, const long long* ids, int T, int V, int E, float* dW,
                       void* stream);
/* the plain lookup e = Embedding(q) and its weight gradient (hieCoAtten.py:27, networks.py:23,56, mhb_coAtt.py:181) */
int vqf_embed_fwd(const float* W, const long long* ids, int T, int V, int E, float* out, void* stream);
int vqf_embed_bwd(const float* dout, const long long* ids, int T, int V, int E, float* dW, void* stream);
/* out = dropout(W[ids]) (hieCoAtten.py:27-28: the lookup and its always-on functional dropout) and its weight gradient, one launch each
 * way; the mask is the one vqf_dropout_f32 draws over the flat (T, E) tensor (keep: (T, E) uint8 or NULL + seed / p_drop).  E % 4 == 0. */
int vqf_embed_dropout_fwd(const float* W, const long long* ids, int T, int V, int E, const uint8_t* keep, uint64_t seed, float p_drop,
                          float* out, void* stream);
int vqf_embed_dropout_bwd(const float* dout, const long long* ids, int T, int V, int E, const uint8_t* keep, uint64_t seed, float p_drop,
                          float* dW, void* stream);
/* time-major forms: ids (N, Tq) as the reference holds them (mfb.py:68), out / dout rows ordered (Tq, N) -- the layout the
 * batch-major LSTM consumes (mfb.py:69 with batch_first=True == T steps of the N-row batch): no transposing copy in between */
int vqf_embed_tanh_fwd_tm(const float* W, const long long* ids, int N, int Tq, int V, int E, float* out, void* stream);
int vqf_embed_tanh_bwd_tm(const float* dout, const float* out, const long long* ids, int N, int Tq, int V, int E, float* dW,
                          void* stream);

/* --------------------------------------------------------------------------
 * Input staging.  data_loader.py:30-32 loads one [2048,14,14] .npy per image
 * and makes it (196,2048) on the CPU (np.transpose(x,(1,2,0)).reshape(-1,2048)).  Here the raw
 * batch src (N, D, L) fp32 (channels outermost, as stored) is transposed on the device into the
 * layout every kernel above consumes, dst (N, L, D), as fp32 (out_bf16 = 0, bit-exact copy) or
 * bf16 (out_bf16 = 1, round-to-nearest-even).  src and dst must not overlap. */
int vqf_feat_transpose(const float* src, int N, int D, int L, int out_bf16, void* dst, void* stream);

/* --------------------------------------------------------------------------
 * Training-step tail (outside the modules, called by the solver loop).
 *
 * vqf_ce_loss     nn.CrossEntropyLoss() of solver.py:28,91: logits (N,A), int64 targets (N);
 *                 loss[0] = mean over rows whose target != -100 of (logsumexp - logit[target]);
 *                 dlogits (N,A) = d loss / d logits (may be NULL).  Same pass for both.
 * vqf_kldiv_loss  nn.KLDivLoss() of solver.py:26,91 (default reduction: mean over all N*A
 *                 elements): loss[0] = mean(t * (log t - logp)), 0 where t == 0;
 *                 dlogp = -t / (N*A) (may be NULL).
 * ws: vqf_loss_ws_bytes(N, A) bytes of scratch (fixed-order partial sums; no atomics).
 */
size_t vqf_loss_ws_bytes(int N, int A);
int vqf_ce_loss(const float* logits, const long long* target, int N, int A, float* loss, float* dlogits,
                void* ws, size_t ws_bytes, void* stream);
int vqf_kldiv_loss(const float* logp, const float* target, int N, int A, float* loss, float* dlogp,
                   void* ws, size_t ws_bytes, void* stream);

/* torch.optim.Adam(model.parameters(), lr) of solver.py:29,93 (amsgrad off, maximize off), all
 * tensors of a step in as few launches as possible (VQF_ADAM_MAX_TENSORS per launch):
 *   g += wd*p;  m += (g-m)(1-b1);  v = v*b2 + (1-b2) g g;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)          (step counts from 1)
 * `tensors` is a HOST array of `count` descriptors holding device pointers. */
#define VQF_ADAM_MAX_TENSORS 32
typedef struct VqfAdamTensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  long long n;
} VqfAdamTensor;
int vqf_adam_step(const VqfAdamTensor* tensors, int count, double lr, double beta1, double beta2, double eps,
                  double weight_decay, long long step, void* stream);

/* --------------------------------------------------------------------------
 * HBM yardsticks (measurement only; bench.py's `hbm_yardsticks`): what a plain streaming kernel of this library reaches,
 * in the units the HBM-bound stages above are priced in.  16 bytes per lane, grid-stride, the launch shape that streams 2 GB
 * buffers fastest (tools/hbm_probe.hip: 2 - 4 workgroups of 256 threads per CU; csrc/yardstick.hip); nt != 0: non-temporal
 * loads / stores.  nbytes % 16 == 0, 16-byte aligned.
 *   vqf_hbm_copy        dst[0..nbytes) = src[0..nbytes)            (read : write = 1 : 1)
 *   vqf_hbm_read_sweep  block_sums[b] = sum of the fp32 values workgroup b read (b < vqf_hbm_read_sweep_blocks(nbytes)):
 *                       a pure read stream whose loads cannot be dropped. */
int vqf_hbm_copy(const void* src, void* dst, long long nbytes, int nt, void* stream);
int vqf_hbm_read_sweep_blocks(long long nbytes);
int vqf_hbm_read_sweep(const void* src, long long nbytes, int nt, float* block_sums, void* stream);

/* --------------------------------------------------------------------------
 * Opt-in profiler: hipEvent pairs around every kernel launch, on the stream
 * the kernel is launched on.  Off by default (zero overhead).  An event pair costs the stream
 * ~6-10 us between two kernels (rocprofv3 kernel trace of bench.py: 10.4 us gaps between bracketed
 * launches, none between unbracketed ones), so a timed region should bracket only what it reports:
 * vqf_prof_filter(min_mnk) > 0 restricts the brackets to GEMM launches with M * N * K >= min_mnk.
 */
void vqf_prof_enable(int on);
void vqf_prof_filter(long long min_mnk);
void vqf_prof_reset(void);
int vqf_prof_num_kernels(void);
const char* vqf_prof_kernel_name(int id);
/* synchronises the recorded events; returns launches and total milliseconds */
int vqf_prof_get(int id, long long* launches, double* total_ms);
/* same, restricted to launches whose shape tag matches (GEMMs tag M,N,K; -1 = any) */
int vqf_prof_get_shape(int id, int d0, int d1, int d2, long long* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* VQA_FUSION_H */
