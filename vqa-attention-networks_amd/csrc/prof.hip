// Opt-in hipEvent profiler, library options, ABI/version entry points.
#include "common.h"
#include <mutex>
#include <stdlib.h>
#include <vector>

int g_vqf_prof_on = 0;
int g_vqf_opt[VQF_OPT_COUNT];
long long g_vqf_stat[VQF_STAT_COUNT] = {};

namespace {
// environment variable of option i = "VQF_" + the name of its VQF_OPT_* constant (include/vqa_fusion.h says so;
// tests/test_abi_cpu.py checks this table against the header); kOptEnvOld: the r02 spellings of two of them, still read
const char* const kOptEnv[VQF_OPT_COUNT] = {
    "VQF_GEMM_F32_PERSIST", "VQF_GEMM_BF16_PERSIST", "VQF_GEMM_F32_LOOP", "VQF_GEMM_BF16_LOOP", "VQF_GEMM_F32_BIG",
    "VQF_GEMM_BF16_BIG", "VQF_GEMM_F32_WAVE", "VQF_FUSE_COAL", "VQF_FUSE_LS", "VQF_FUSE_LS_BWD", "VQF_GEMM_CU_LIMIT",
    "VQF_GEMM_F32_EDGE", "VQF_GEMM_F32_ROUNDS", "VQF_GEMM_SPLITK_FUSED", "VQF_GEMM_F32_STREAMK", "VQF_GEMM_SPLITK_ORDER", "VQF_GEMM_F32_SAMPLE", "VQF_GEMM_F32_N80"};
const char* const kOptEnvOld[VQF_OPT_COUNT] = {
    nullptr, nullptr, "VQF_GEMM_F32_PP", "VQF_GEMM_BF16_PP", nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
// the environment is read ONCE, when the library is loaded (command-line A/Bs); never on a launch path
struct OptInit {
  OptInit() {
    for (int i = 0; i < VQF_OPT_COUNT; ++i) {
      const char* e = getenv(kOptEnv[i]);
      if (!e && kOptEnvOld[i]) e = getenv(kOptEnvOld[i]);
      g_vqf_opt[i] = (e && ((e[0] >= '0' && e[0] <= '9') || e[0] == '-')) ? atoi(e) : -1;
    }
  }
} g_opt_init;
}  // namespace

namespace {
struct Pair { hipEvent_t a, b; int id; int d[3]; };
std::mutex g_mu;
std::vector<Pair> g_pairs;
std::vector<hipEvent_t> g_free;
thread_local hipEvent_t t_start = nullptr;
thread_local int t_dims[3] = {0, 0, 0};
long long g_min_mnk = 0;     // vqf_prof_filter: bracket only GEMM launches whose M * N * K reaches this (0 = every kernel)

const char* const kNames[KID_COUNT] = {
    "gemm_f32_a0b0(fwd)", "gemm_f32_a0b1(dgrad)", "gemm_f32_a1b0", "gemm_f32_a1b1(wgrad)",
    "splitk_reduce", "colsum", "group_reduce", "relu_bwd",
    "att_logits_fwd", "att_logits_bwd", "glimpse_pool_fwd", "glimpse_pool_bwd",
    "mfb_fuse_fwd", "l2_group_norm", "scale_rows", "rowdot", "l2_norm_bwd_coef", "mfb_fuse_bwd",
    "dropout", "tanh_dropout_fwd", "tanh_dropout_bwd", "softmax_rows_fwd", "softmax_rows_bwd",
    "gemm_bf16", "cast_f32_bf16",
    "lstm_seq_fwd(all steps)", "lstm_seq_bwd(all steps)",
    "ce_loss", "kldiv_loss", "adam_step", "feat_transpose",
    "lstm_cell_fwd", "lstm_cell_bwd", "embed_tanh_fwd", "embed_tanh_bwd", "hbm_copy", "hbm_read_sweep", "multi_add", "multi_copy", "hie_hv_fwd", "hie_head_bwd", "hie_rank_add", "hie_rank_left", "hie_slab_sum", "hie_affinity"};

hipEvent_t get_event() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

bool vqf_prof_begin(int id, hipStream_t s) {
  if (g_min_mnk > 0) {       // an event pair costs the stream ~6-10 us: a timed region brackets its dominant launches only
    const bool gemm = (id >= KID_GEMM_A0B0 && id <= KID_GEMM_A1B1) || id == KID_GEMM_BF16;   // these tag M, N, K right before the launch
    if (!gemm || (long long)t_dims[0] * t_dims[1] * t_dims[2] < g_min_mnk) return false;
  }
  t_start = get_event();
  (void)hipEventRecord(t_start, s);
  return true;
}
void vqf_prof_end(int id, hipStream_t s) {
  hipEvent_t b = get_event();
  (void)hipEventRecord(b, s);
  std::lock_guard<std::mutex> lk(g_mu);
  g_pairs.push_back({t_start, b, id, {t_dims[0], t_dims[1], t_dims[2]}});
}
void vqf_prof_dims(int d0, int d1, int d2) { t_dims[0] = d0; t_dims[1] = d1; t_dims[2] = d2; }

extern "C" {
int vqf_abi_version(void) { return 7; }
int vqf_set_option(int option, int value, int* previous) {
  if (option < 0 || option >= VQF_OPT_COUNT) return VQF_E_BADARG;
  if (previous) *previous = g_vqf_opt[option];
  g_vqf_opt[option] = value < 0 ? -1 : value;
  return VQF_OK;
}
int vqf_stat_get(int stat, long long* value) {
  if (stat < 0 || stat >= VQF_STAT_COUNT || !value) return VQF_E_BADARG;
  *value = __atomic_load_n(&g_vqf_stat[stat], __ATOMIC_RELAXED);
  return VQF_OK;
}
int vqf_get_option(int option, int* value) {
  if (option < 0 || option >= VQF_OPT_COUNT || !value) return VQF_E_BADARG;
  *value = g_vqf_opt[option];
  return VQF_OK;
}
const char* vqf_option_env_name(int option) { return (option >= 0 && option < VQF_OPT_COUNT) ? kOptEnv[option] : ""; }
const char* vqf_build_info(void) {
  return "libvqa_fusion gfx950 fp32-mfma(v_mfma_f32_32x32x2_f32) tiles 128x128x16 + 256x256x16(lds-dma, staggered) + "
         "32x64-per-wave(small M) bf16-mfma(v_mfma_f32_16x16x32_bf16 / 32x32x16) tiles 128x128x32 + "
         "256x256x32(lds-dma, ping-pong) wave64 philox4x32-10";
}
void vqf_prof_enable(int on) { g_vqf_prof_on = on ? 1 : 0; }
void vqf_prof_filter(long long min_mnk) { g_min_mnk = min_mnk > 0 ? min_mnk : 0; }
void vqf_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& p : g_pairs) { g_free.push_back(p.a); g_free.push_back(p.b); }
  g_pairs.clear();
}
int vqf_prof_num_kernels(void) { return KID_COUNT; }
const char* vqf_prof_kernel_name(int id) { return (id >= 0 && id < KID_COUNT) ? kNames[id] : ""; }
int vqf_prof_get(int id, long long* launches, double* total_ms) {
  return vqf_prof_get_shape(id, -1, -1, -1, launches, total_ms);
}
int vqf_prof_get_shape(int id, int d0, int d1, int d2, long long* launches, double* total_ms) {
  if (id < 0 || id >= KID_COUNT || !launches || !total_ms) return VQF_E_BADARG;
  std::lock_guard<std::mutex> lk(g_mu);
  long long n = 0;
  double ms = 0.0;
  for (auto& p : g_pairs) {
    if (p.id != id) continue;
    if ((d0 >= 0 && p.d[0] != d0) || (d1 >= 0 && p.d[1] != d1) || (d2 >= 0 && p.d[2] != d2)) continue;
    hipError_t e = hipEventSynchronize(p.b);
    if (e != hipSuccess) return (int)e;
    float t = 0.f;
    e = hipEventElapsedTime(&t, p.a, p.b);
    if (e != hipSuccess) return (int)e;
    ms += t;
    ++n;
  }
  *launches = n;
  *total_ms = ms;
  return VQF_OK;
}
}
