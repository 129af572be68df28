"""ctypes binding of libvqa_fusion.so (C ABI declared in include/vqa_fusion.h).

The product path has NO CPU fallback: if the shared library is missing or an
op is handed a non-GPU tensor, it raises.  `build()` compiles the library
in-tree with hipcc for gfx950 (cross-compiles without a GPU).
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(_HERE)
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libvqa_fusion.so")
if os.environ.get("VQF_LIB"):          # A/B builds (tools/build_variant.sh); never a different implementation
    LIB_PATH = os.path.abspath(os.environ["VQF_LIB"])
HEADER_PATH = os.path.join(os.path.dirname(PKG_DIR), "include", "vqa_fusion.h")

# The one place the expected ABI number lives (csrc/prof.hip returns it from vqf_abi_version()).
ABI_VERSION = 7

_lock = threading.Lock()
_lib = None

c_f = ctypes.c_void_p          # device float* (raw pointer)
c_i = ctypes.c_int
c_sz = ctypes.c_size_t
c_u64 = ctypes.c_uint64
c_p = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol of include/vqa_fusion.h
SIGNATURES = {
    "vqf_abi_version": (c_i, []),
    "vqf_build_info": (ctypes.c_char_p, []),
    "vqf_set_option": (c_i, [c_i, c_i, ctypes.POINTER(c_i)]),
    "vqf_get_option": (c_i, [c_i, ctypes.POINTER(c_i)]),
    "vqf_option_env_name": (ctypes.c_char_p, [c_i]),
    "vqf_stat_get": (c_i, [c_i, ctypes.POINTER(ctypes.c_longlong)]),
    "vqf_gemm_f32_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "vqf_gemm_f32": (c_i, [c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_p, c_sz, c_p]),
    "vqf_gemm_f32_big_rows": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "vqf_gemm_f32_sample_supported": (c_i, [c_i, c_i, c_i, c_i]),
    "vqf_gemm_f32_sample": (c_i, [c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_p]),
    "vqf_gemm_f32_rowscale": (c_i, [c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_p]),
    "vqf_gemm_f32_batched": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, ctypes.c_longlong,
                                   c_f, c_i, ctypes.c_longlong, c_f, c_i, ctypes.c_longlong, c_i, c_p]),
    "vqf_gemm_bf16_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "vqf_gemm_bf16": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_f, c_i, c_f, c_i, c_p, c_sz, c_p]),
    "vqf_gemm_bf16_rowscale": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_p]),
    "vqf_cast_f32_bf16": (c_i, [c_f, c_i, c_i, c_i, c_p, c_i, c_p]),
    "vqf_colsum_ws_bytes": (c_sz, [c_i, c_i]),
    "vqf_colsum_f32": (c_i, [c_f, c_i, c_i, c_i, c_f, c_p, c_sz, c_p]),
    "vqf_group_reduce_f32": (c_i, [c_f, c_i, c_i, c_i, c_f, c_p]),
    "vqf_relu_bwd_f32": (c_i, [c_f, c_f, c_i, c_i, c_f, c_f, c_p, c_sz, c_p]),
    "vqf_att_logits_fwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_p]),
    "vqf_att_logits_fwd_lin": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_p]),
    "vqf_att_logits_bwd_ws_bytes": (c_sz, [c_i, c_i]),
    "vqf_att_logits_bwd_rowscale": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_sz, c_p]),
    "vqf_att_logits_bwd_rowscale_obf16": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p, c_f, c_f, c_f, c_p, c_sz, c_p]),
    "vqf_att_logits_bwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_sz, c_p]),
    "vqf_glimpse_pool_fwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_p]),
    "vqf_glimpse_pool_bwd": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_p]),
    "vqf_glimpse_pool_fwd_bf16": (c_i, [c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_p]),
    "vqf_glimpse_pool_bwd_bf16": (c_i, [c_f, c_f, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    "vqf_dropout_f32": (c_i, [c_f, c_p, c_u64, ctypes.c_float, ctypes.c_longlong, c_f, c_p]),
    "vqf_dropout_bt": (c_i, [c_f, ctypes.c_longlong, ctypes.c_longlong, c_p, c_u64, ctypes.c_float, c_i, c_i, c_i, c_f,
                             ctypes.c_longlong, ctypes.c_longlong, c_p]),
    "vqf_gate_tanh_sigmoid_fwd": (c_i, [c_f, c_f, ctypes.c_longlong, c_f, c_p]),
    "vqf_gate_tanh_sigmoid_bwd": (c_i, [c_f, c_f, c_f, ctypes.c_longlong, c_f, c_f, c_p]),
    "vqf_tanh_dropout_fwd": (c_i, [c_f, c_f, c_p, c_u64, ctypes.c_float, ctypes.c_longlong, c_f, c_p]),
    "vqf_tanh_dropout_bwd": (c_i, [c_f, c_f, c_p, c_u64, ctypes.c_float, ctypes.c_longlong, c_f, c_p]),
    "vqf_tanh_dropout_fwd2d": (c_i, [c_f, c_i, c_f, c_i, c_p, c_u64, ctypes.c_float, c_i, c_i, c_f, c_i, c_p]),
    "vqf_tanh_dropout_bwd2d": (c_i, [c_f, c_i, c_f, c_i, c_p, c_u64, ctypes.c_float, c_i, c_i, c_f, c_i, c_p]),
    "vqf_relu_bwd_rank1_f32": (c_i, [c_f, c_f, c_f, c_f, c_i, ctypes.c_float, c_i, c_i, c_f, c_f, c_p, c_sz, c_p]),
    "vqf_multi_add_f32": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p]),
    "vqf_multi_copy_f32": (c_i, [c_p, c_p, c_p, c_i, c_p]),
    "vqf_hie_stream_supported": (c_i, [c_i, c_i, c_i, c_i]),
    "vqf_hie_chunks": (c_i, [c_i, c_i]),
    "vqf_hie_hv_fwd": (c_i, [c_f, c_i, c_f, c_f, c_i, c_p, c_u64, ctypes.c_float, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_p]),
    "vqf_hie_head_bwd": (c_i, [c_f, c_i, c_f, c_f, c_f, c_p, c_u64, ctypes.c_float, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_f, c_i,
                               c_f, c_i, c_p]),
    "vqf_hie_rank_add": (c_i, [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_p]),
    "vqf_hie_rank_left": (c_i, [c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_p]),
    "vqf_hie_slab_sum": (c_i, [c_f, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_p]),
    "vqf_hie_affinity_supported": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "vqf_hie_affinity": (c_i, [c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_i, c_f, c_p, c_u64, ctypes.c_float, c_i, c_i, c_i, c_i, c_f, c_p]),
    "vqf_softmax_rows_fwd": (c_i, [c_f, c_i, c_i, c_f, c_p]),
    "vqf_softmax_rows_bwd": (c_i, [c_f, c_f, c_i, c_i, c_f, c_p]),
    "vqf_log_softmax_rows_fwd": (c_i, [c_f, c_i, c_i, c_f, c_p]),
    "vqf_log_softmax_rows_bwd": (c_i, [c_f, c_f, c_i, c_i, c_f, c_p]),
    "vqf_mfb_fuse_fwd": (c_i, [c_f, c_f, c_f, c_f, c_p, c_u64, ctypes.c_float, c_i, c_i, c_i, c_f, c_f, c_f, c_p]),
    "vqf_l2_group_norm": (c_i, [c_f, c_i, c_i, c_f, c_f, c_p]),
    "vqf_scale_rows": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_p]),
    "vqf_rowdot": (c_i, [c_f, c_f, c_i, c_i, c_f, c_p]),
    "vqf_l2_norm_bwd_coef": (c_i, [c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_p]),
    "vqf_l2_norm_bwd_coef_lin": (c_i, [c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_p]),
    "vqf_mfb_fuse_bwd_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "vqf_mfb_fuse_bwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_u64, ctypes.c_float,
                               c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_sz, c_p]),
    "vqf_mfb_fuse_fwd_pbf16": (c_i, [c_p, c_f, c_f, c_p, c_u64, ctypes.c_float, c_i, c_i, c_i, c_f, c_f, c_p]),
    "vqf_mfb_fuse_fwd_pbf16_rb": (c_i, [c_p, c_f, c_f, c_p, c_u64, ctypes.c_float, c_i, c_i, c_i, c_f, c_p, c_i, c_f, c_p]),
    "vqf_mfb_fuse_bwd_pbf16": (c_i, [c_f, c_f, c_f, c_f, c_f, c_p, c_f, c_f, c_p, c_u64, ctypes.c_float,
                                     c_i, c_i, c_i, c_p, c_f, c_f, c_p, c_sz, c_p]),
    "vqf_mfb_fuse_bwd_bf16dp": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_u64, ctypes.c_float,
                                      c_i, c_i, c_i, c_p, c_f, c_f, c_p, c_sz, c_p]),
    "vqf_lstm_seq_supported": (c_i, [c_i, c_i]),
    "vqf_lstm_seq_ws_bytes": (c_sz, [c_i, c_i]),
    "vqf_lstm_seq_fwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_p, c_sz, c_p]),
    "vqf_lstm_seq_bwd": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_i, c_p, c_sz, c_p]),
    "vqf_lstm_cell_fwd": (c_i, [c_f, c_f, c_i, c_i, c_f, c_f, c_p]),
    "vqf_lstm_step_supported": (c_i, [c_i, c_i]),
    "vqf_lstm_step_fwd": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_p]),
    "vqf_lstm_cell_bwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_p]),
    "vqf_embed_tanh_fwd": (c_i, [c_f, c_p, c_i, c_i, c_i, c_f, c_p]),
    "vqf_embed_tanh_bwd": (c_i, [c_f, c_f, c_p, c_i, c_i, c_i, c_f, c_p]),
    "vqf_embed_fwd": (c_i, [c_f, c_p, c_i, c_i, c_i, c_f, c_p]),
    "vqf_embed_bwd": (c_i, [c_f, c_p, c_i, c_i, c_i, c_f, c_p]),
    "vqf_embed_dropout_fwd": (c_i, [c_f, c_p, c_i, c_i, c_i, c_p, c_u64, ctypes.c_float, c_f, c_p]),
    "vqf_embed_dropout_bwd": (c_i, [c_f, c_p, c_i, c_i, c_i, c_p, c_u64, ctypes.c_float, c_f, c_p]),
    "vqf_embed_tanh_fwd_tm": (c_i, [c_f, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "vqf_embed_tanh_bwd_tm": (c_i, [c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "vqf_feat_transpose": (c_i, [c_f, c_i, c_i, c_i, c_i, c_p, c_p]),
    "vqf_loss_ws_bytes": (c_sz, [c_i, c_i]),
    "vqf_ce_loss": (c_i, [c_f, c_p, c_i, c_i, c_f, c_f, c_p, c_sz, c_p]),
    "vqf_kldiv_loss": (c_i, [c_f, c_f, c_i, c_i, c_f, c_f, c_p, c_sz, c_p]),
    "vqf_adam_step": (c_i, [c_p, c_i, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                            ctypes.c_double, ctypes.c_longlong, c_p]),
    "vqf_hbm_copy": (c_i, [c_p, c_p, ctypes.c_longlong, c_i, c_p]),
    "vqf_hbm_read_sweep_blocks": (c_i, [ctypes.c_longlong]),
    "vqf_hbm_read_sweep": (c_i, [c_p, ctypes.c_longlong, c_i, c_f, c_p]),
    "vqf_prof_enable": (None, [c_i]),
    "vqf_prof_filter": (None, [ctypes.c_longlong]),
    "vqf_prof_reset": (None, []),
    "vqf_prof_num_kernels": (c_i, []),
    "vqf_prof_kernel_name": (ctypes.c_char_p, [c_i]),
    "vqf_prof_get": (c_i, [c_i, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double)]),
    "vqf_prof_get_shape": (c_i, [c_i, c_i, c_i, c_i, ctypes.POINTER(ctypes.c_longlong),
                                 ctypes.POINTER(ctypes.c_double)]),
}

class AdamTensor(ctypes.Structure):
    """VqfAdamTensor of include/vqa_fusion.h."""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_longlong)]


_ERR = {-1: "VQF_E_BADARG", -2: "VQF_E_ALIGN", -3: "VQF_E_UNSUPPORTED", -4: "VQF_E_WORKSPACE", -5: "VQF_E_TIMEOUT"}


class VqfError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile csrc/*.hip into libvqa_fusion.so for gfx950 (make -C csrc)."""
    srcs = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith((".hip", ".h"))]
    srcs.append(HEADER_PATH)
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if not stale:
        return LIB_PATH
    cmd = ["make", "-C", CSRC_DIR, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout)
        print(r.stderr)
    if r.returncode:
        raise VqfError("hipcc build of libvqa_fusion.so failed (see output above)")
    return LIB_PATH


def load():
    """Return the loaded library; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise VqfError(
                "libvqa_fusion.so is missing (%s).  The HIP extension is the product path and "
                "there is no CPU fallback: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C %s`." % (LIB_PATH, CSRC_DIR))
        # torch first: PyTorch-ROCm bundles its own libamdhip64 (same SONAME); loading ours
        # before it would put two HIP runtimes in the process (kernels would then be launched
        # on a runtime that has not opened torch's device/streams -> hipErrorNoDevice).
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)        # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.vqf_abi_version() != ABI_VERSION:
            raise VqfError("libvqa_fusion.so ABI version %d, host expects %d: rebuild with `make -C %s`"
                           % (lib.vqf_abi_version(), ABI_VERSION, CSRC_DIR))
        _lib = lib
        return _lib


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise VqfError("%s: %s" % (what, _ERR.get(rc, "error %d" % rc)))
    raise VqfError("%s: HIP error %d" % (what, rc))
