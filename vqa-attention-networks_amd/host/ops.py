"""Tensor-level wrappers over the C ABI (raw device pointers + current stream).

PyTorch is plumbing here: it owns device memory (caching allocator) and the
HIP stream; all arithmetic happens in libvqa_fusion.so.  There is no CPU
fallback -- a CPU tensor raises.
"""
import ctypes
import torch

from . import lib as _l

GEMM_RELU = 1
GEMM_ACCUM = 2
POOL_K = 5


def _lib():
    return _l.load()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _l.VqfError("vqa fusion ops need GPU tensors (HIP extension is the only path; "
                              "no CPU fallback)")
        if t.dtype != torch.float32:
            raise _l.VqfError("fp32 tensor expected, got %s" % t.dtype)
        if not t.is_contiguous():
            raise _l.VqfError("contiguous tensor expected")


class _Workspace:
    """Grow-only scratch buffer per (device, stream): kernels on different streams may run
    concurrently (the image projection runs on a side stream), so they must not share slabs."""

    def __init__(self):
        self.bufs = {}

    def get(self, device, nbytes):
        nbytes = max(int(nbytes), 256)
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        b = self.bufs.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
            self.bufs[key] = b
        return b


_ws = _Workspace()
SPLITK_WS_BYTES = 192 << 20


def workspace(device, nbytes):
    return _ws.get(device, nbytes)


# ---------------------------------------------------------------------------
# library options (include/vqa_fusion.h VQF_OPT_*): process-wide launch policy, cached in the library
OPTIONS = {"gemm_f32_persist": 0, "gemm_bf16_persist": 1, "gemm_f32_loop": 2, "gemm_bf16_loop": 3, "gemm_f32_big": 4,
           "gemm_bf16_big": 5, "gemm_f32_wave": 6, "fuse_coal": 7, "fuse_ls": 8, "fuse_ls_bwd": 9, "gemm_cu_limit": 10,
           "gemm_f32_edge": 11, "gemm_f32_rounds": 12, "gemm_splitk_fused": 13, "gemm_f32_streamk": 14,
           "gemm_splitk_order": 15, "gemm_f32_sample": 16, "gemm_f32_n80": 17}


def set_option(name, value):
    """Set a library option (None / negative = the library's default); returns the value it replaced (-1 = default)."""
    prev = ctypes.c_int(0)
    _l.check(_lib().vqf_set_option(OPTIONS[name], -1 if value is None else int(value), ctypes.byref(prev)),
             "vqf_set_option(%s)" % name)
    return prev.value


def get_option(name):
    v = ctypes.c_int(0)
    _l.check(_lib().vqf_get_option(OPTIONS[name], ctypes.byref(v)), "vqf_get_option(%s)" % name)
    return v.value


STATS = {"gemm_f32_tile128": 0, "gemm_f32_big": 1, "gemm_f32_wave": 2, "gemm_bf16_tile128": 3, "gemm_bf16_big": 4,
         "gemm_f32_sample": 5, "gemm_f32_n80": 6}


def stat(name):
    """Launches routed to a GEMM kernel family since the library was loaded (include/vqa_fusion.h VQF_STAT_*)."""
    v = ctypes.c_longlong(0)
    _l.check(_lib().vqf_stat_get(STATS[name], ctypes.byref(v)), "vqf_stat_get(%s)" % name)
    return v.value


class options:
    """with ops.options(gemm_f32_persist=0): ...   sets the options for the block and restores what was there."""

    def __init__(self, **kw):
        self.kw, self.prev = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.prev[k] = set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_option(k, v)
        return False


def gemm(a, b, ta=False, tb=False, bias=None, relu=False, out=None, accumulate=False,
         M=None, N=None, K=None, splitk=True):
    """C[M,N] = Aop @ Bop^T (+bias) ; a/b are 2-D contiguous.

    ta=False: a is (M,K); ta=True: a is (K,M).   tb=False: b is (N,K); tb=True: b is (K,N).
    """
    _chk(bias)
    for t in (a, b, out):        # 2-D operands may be row-strided views (the ABI takes lda/ldb/ldc)
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise _l.VqfError("gemm: 2-D fp32 GPU tensors with contiguous rows expected")
    if M is None:
        M = a.shape[1] if ta else a.shape[0]
    if K is None:
        K = a.shape[0] if ta else a.shape[1]
        kb = b.shape[0] if tb else b.shape[1]
        if kb != K:
            raise _l.VqfError("gemm: inner dimensions differ (%d vs %d)" % (K, kb))
    if N is None:
        N = b.shape[1] if tb else b.shape[0]
    if out is None:
        if accumulate:
            raise _l.VqfError("gemm: accumulate needs out")
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    flags = (GEMM_RELU if relu else 0) | (GEMM_ACCUM if accumulate else 0)
    ws = None
    if splitk:
        ws = workspace(a.device, max(SPLITK_WS_BYTES, int(_lib().vqf_gemm_f32_ws_bytes(int(ta), int(tb), M, N, K))))
    rc = _lib().vqf_gemm_f32(int(ta), int(tb), M, N, K, _ptr(a), a.stride(0), _ptr(b), b.stride(0),
                             _ptr(out), out.stride(0), _ptr(bias), flags,
                             _ptr(ws), ws.numel() if ws is not None else 0, _stream())
    _l.check(rc, "vqf_gemm_f32")
    return out


def gemm_rows(a, b, L, tb=False, bias=None, relu=False, out=None):
    """C = a @ bop^T (+ bias) (relu) for a (NS*L, K) whose rows come in samples of L rows (the image regions of a sample): the
    per-sample-tile kernel (include/vqa_fusion.h vqf_gemm_f32_sample) where it takes the shape, vqf_gemm_f32 otherwise.
    b: (N, K), or (K, N) with tb; rows of a / b / out may be strided."""
    M, K = a.shape
    N = b.shape[1] if tb else b.shape[0]
    ok = (L > 0 and M % L == 0 and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2
          and a.stride(1) == 1 and b.stride(1) == 1 and (b.shape[0] if tb else b.shape[1]) == K
          and _lib().vqf_gemm_f32_sample_supported(M // L, L, N, K)
          and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0)
    if out is not None:
        ok = ok and out.dtype == torch.float32 and out.dim() == 2 and out.stride(1) == 1 and out.stride(0) % 2 == 0 and out.data_ptr() % 8 == 0
    if not ok:
        return gemm(a, b, tb=tb, bias=bias, relu=relu, out=out)
    _chk(bias)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _l.check(_lib().vqf_gemm_f32_sample(int(bool(tb)), M // L, int(L), N, K, _ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out),
                                        out.stride(0), _ptr(bias), GEMM_RELU if relu else 0, _stream()), "vqf_gemm_f32_sample")
    return out


def gemm_rows_supported(NS, L, N, K):
    """whether gemm_rows runs (NS * L, K) x (N, K) products on the per-sample-tile kernel (include/vqa_fusion.h)"""
    return bool(_lib().vqf_gemm_f32_sample_supported(int(NS), int(L), int(N), int(K)))


def gemm_big_rows(ta, tb, M, N, K):
    """leading rows of an fp32 (ta, tb, M, N, K) product that run on the 256x256-tile kernel (M, 0, or the whole-rounds block of
    a mid-size shape: the other M - rows rows are a second launch): include/vqa_fusion.h vqf_gemm_f32_big_rows"""
    return int(_lib().vqf_gemm_f32_big_rows(int(bool(ta)), int(bool(tb)), int(M), int(N), int(K)))


def gemm_rowscale(a, b, rowscale, rows_per_scale, bias=None, relu=False):
    """C = relu?(rowscale[m // rows_per_scale] * (a @ b^T) + bias): a (M,K), b (N,K) fp32; the per-sample scale sits in the
    GEMM epilogue (the co-attention conv on the un-normalised fusion output)."""
    _chk(a, b, rowscale, bias)
    M, K = a.shape
    N = b.shape[0]
    if b.shape[1] != K or rowscale.numel() * rows_per_scale < M:
        raise _l.VqfError("gemm_rowscale: shape mismatch")
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _l.check(_lib().vqf_gemm_f32_rowscale(0, 0, M, N, K, _ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), N,
                                          _ptr(bias), GEMM_RELU if relu else 0, _ptr(rowscale), int(rows_per_scale),
                                          _stream()), "vqf_gemm_f32_rowscale")
    return out


def _chk_bf16(*ts):
    for t in ts:
        if not t.is_cuda or t.dtype != torch.bfloat16 or t.stride(-1) != 1:
            raise _l.VqfError("bf16 GPU tensor with contiguous rows expected")


def cast_bf16(x, pad_to=8):
    """fp32 (R,C) -> bf16 (R, C rounded up to a multiple of pad_to), zero-padded columns."""
    _chk(x)
    R, C = x.shape
    Cp = (C + pad_to - 1) // pad_to * pad_to
    y = torch.empty((R, Cp), dtype=torch.bfloat16, device=x.device)
    _l.check(_lib().vqf_cast_f32_bf16(_ptr(x), R, C, x.stride(0), _ptr(y), Cp, _stream()), "vqf_cast_f32_bf16")
    return y


GEMM_OUT_BF16 = 4


def gemm_bf16(a, b, ta=False, tb=False, bias=None, relu=False, M=None, N=None, K=None, out=None,
              accumulate=False, out_bf16=False):
    """C fp32 = Aop @ Bop^T with bf16 operands (row strides may exceed the logical widths).
    out_bf16: store C as bf16 (large-tile kernel only); returns None when that kernel does not apply."""
    _chk_bf16(a, b)
    _chk(bias)
    if out is not None:          # 2-D, rows may be strided (the ABI takes ldc)
        want = torch.bfloat16 if out_bf16 else torch.float32
        if not out.is_cuda or out.dtype != want or out.dim() != 2 or out.stride(1) != 1:
            raise _l.VqfError("gemm_bf16: out must be a 2-D %s GPU tensor with contiguous rows" % want)
    if M is None:
        M = a.shape[1] if ta else a.shape[0]
    if K is None:
        K = a.shape[0] if ta else a.shape[1]
    if N is None:
        N = b.shape[1] if tb else b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=a.device)
    flags = (GEMM_RELU if relu else 0) | (GEMM_ACCUM if accumulate else 0) | (GEMM_OUT_BF16 if out_bf16 else 0)
    ws = workspace(a.device, max(SPLITK_WS_BYTES, int(_lib().vqf_gemm_bf16_ws_bytes(int(ta), int(tb), M, N, K))))
    rc = _lib().vqf_gemm_bf16(int(ta), int(tb), M, N, K, _ptr(a), a.stride(0), _ptr(b), b.stride(0),
                              _ptr(out), out.stride(0), _ptr(bias), flags, _ptr(ws), ws.numel(), _stream())
    if out_bf16 and rc == -3:          # VQF_E_UNSUPPORTED: shape outside the large-tile kernel -> caller uses fp32 output
        return None
    _l.check(rc, "vqf_gemm_bf16")
    return out


def gemm_bf16_rowscale(a, b, rowscale, rows_per_scale, bias=None, relu=False, K=None):
    """gemm_rowscale with bf16 operands: C fp32 = relu?(rowscale[m // rows_per_scale] * (a @ b^T) + bias), a (M, >=K), b (N, >=K)"""
    _chk_bf16(a, b)
    _chk(rowscale, bias)
    M, N = a.shape[0], b.shape[0]
    if K is None:
        K = a.shape[1]
    if b.shape[1] < K or a.shape[1] < K or rowscale.numel() * rows_per_scale < M:
        raise _l.VqfError("gemm_bf16_rowscale: shape mismatch")
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _l.check(_lib().vqf_gemm_bf16_rowscale(0, 0, M, N, K, _ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), N, _ptr(bias),
                                           GEMM_RELU if relu else 0, _ptr(rowscale), int(rows_per_scale), _stream()),
             "vqf_gemm_bf16_rowscale")
    return out


def _chk3(*ts):
    """3-D fp32 GPU operands of the batched GEMM: innermost dimension contiguous, row and batch strides free (column blocks
    of wider 2-D buffers viewed as (B, rows, cols))"""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 3 or t.stride(2) != 1:
            raise _l.VqfError("bgemm: 3-D fp32 GPU tensors with a contiguous innermost dimension expected")


def bgemm(a, b, ta=False, tb=False, out=None, accumulate=False):
    """Batched: a (B,M,K)|(B,K,M), b (B,N,K)|(B,K,N) -> (B,M,N)."""
    _chk3(a, b, out)
    Bn = a.shape[0]
    M = a.shape[2] if ta else a.shape[1]
    K = a.shape[1] if ta else a.shape[2]
    N = b.shape[2] if tb else b.shape[1]
    kb = b.shape[1] if tb else b.shape[2]
    if kb != K or b.shape[0] != Bn:
        raise _l.VqfError("bgemm: shape mismatch")
    if out is None:
        out = torch.empty((Bn, M, N), dtype=torch.float32, device=a.device)
    rc = _lib().vqf_gemm_f32_batched(int(ta), int(tb), Bn, M, N, K, _ptr(a), a.stride(1), a.stride(0),
                                     _ptr(b), b.stride(1), b.stride(0), _ptr(out), out.stride(1),
                                     out.stride(0), GEMM_ACCUM if accumulate else 0, _stream())
    _l.check(rc, "vqf_gemm_f32_batched")
    return out


def colsum(x):
    _chk(x)
    M, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    need = _lib().vqf_colsum_ws_bytes(M, N)
    ws = workspace(x.device, need)
    _l.check(_lib().vqf_colsum_f32(_ptr(x), M, N, x.stride(0), _ptr(out), _ptr(ws), ws.numel(), _stream()),
             "vqf_colsum_f32")
    return out


def relu_bwd(dx, y, want_bias=True):
    _chk(dx, y)
    M, C = y.shape
    dpre = torch.empty_like(y)
    db = torch.empty(C, dtype=torch.float32, device=y.device) if want_bias else None
    ws = workspace(y.device, _lib().vqf_colsum_ws_bytes(M, C))
    _l.check(_lib().vqf_relu_bwd_f32(_ptr(dx), _ptr(y), M, C, _ptr(dpre), _ptr(db), _ptr(ws), ws.numel(),
                                     _stream()), "vqf_relu_bwd_f32")
    return dpre, db


def att_logits_fwd(hid, w2, b2):
    """logits (M,G) = hid (M,Hh) @ w2 (G,Hh)^T + b2; G in {1,2}."""
    _chk(hid, w2, b2)
    M, Hh = hid.shape
    G = w2.shape[0]
    out = torch.empty((M, G), dtype=torch.float32, device=hid.device)
    _l.check(_lib().vqf_att_logits_fwd(_ptr(hid), _ptr(w2), _ptr(b2), M, Hh, G, _ptr(out), _stream()),
             "vqf_att_logits_fwd")
    return out


def att_logits_fwd_lin(hid, w2, b2, b1):
    """-> (logits (M,G), lin (M,G)): lin = the part of the logit that is linear in the input of the ReLU layer in front
    (include/vqa_fusion.h vqf_att_logits_fwd_lin)."""
    _chk(hid, w2, b2, b1)
    M, Hh = hid.shape
    G = w2.shape[0]
    out = torch.empty((M, G), dtype=torch.float32, device=hid.device)
    lin = torch.empty((M, G), dtype=torch.float32, device=hid.device)
    _l.check(_lib().vqf_att_logits_fwd_lin(_ptr(hid), _ptr(w2), _ptr(b2), _ptr(b1), M, Hh, G, _ptr(out), _ptr(lin),
                                           _stream()), "vqf_att_logits_fwd_lin")
    return out, lin


def att_logits_bwd(dlogits, hid, w2, relu_mask=True, rowscale=None, rows_per_scale=1, out_bf16=False):
    """rowscale (per row group of rows_per_scale rows): the stored dhid_pre is scaled by it, the bias sums are not.
    out_bf16 (two glimpses, through the ReLU, Hh % 8 == 0): dhid_pre is stored as bf16 (the operand of the bf16 gradient GEMMs)."""
    _chk(dlogits, hid, w2, rowscale)
    M, Hh = hid.shape
    G = w2.shape[0]
    dw2 = torch.empty((G, Hh), dtype=torch.float32, device=hid.device)
    db2 = torch.empty(G, dtype=torch.float32, device=hid.device)
    db1 = torch.empty(Hh, dtype=torch.float32, device=hid.device)
    ws = workspace(hid.device, _lib().vqf_att_logits_bwd_ws_bytes(M, Hh))
    if out_bf16:
        if not relu_mask or G != 2 or Hh % 8:
            raise _l.VqfError("att_logits_bwd: bf16 output needs the two-glimpse head through its ReLU and Hh % 8 == 0")
        dpre = torch.empty((M, Hh), dtype=torch.bfloat16, device=hid.device)
        _l.check(_lib().vqf_att_logits_bwd_rowscale_obf16(_ptr(dlogits), _ptr(hid), _ptr(w2), _ptr(rowscale), int(rows_per_scale),
                                                          M, Hh, G, ctypes.c_void_p(dpre.data_ptr()), _ptr(dw2), _ptr(db2), _ptr(db1),
                                                          _ptr(ws), ws.numel(), _stream()), "vqf_att_logits_bwd_rowscale_obf16")
        return dpre, dw2, db2, db1
    dpre = torch.empty_like(hid)
    _l.check(_lib().vqf_att_logits_bwd_rowscale(_ptr(dlogits), _ptr(hid), _ptr(w2), _ptr(rowscale), int(rows_per_scale),
                                                M, Hh, G, int(bool(relu_mask)), _ptr(dpre), _ptr(dw2), _ptr(db2), _ptr(db1),
                                                _ptr(ws), ws.numel(), _stream()), "vqf_att_logits_bwd")
    return dpre, dw2, db2, db1


def glimpse_pool_fwd(feat, logits, unit_softmax, pooled_out=None):
    """feat (N,S,C) fp32 | bf16, logits (N*S,G) -> wts (N,G,S), pooled (N,G*C) (fp32; pooled_out: written there)."""
    bf = feat.dtype == torch.bfloat16
    (_chk_bf16 if bf else _chk)(feat)
    if not feat.is_contiguous():
        raise _l.VqfError("contiguous feature tensor expected")
    _chk(logits)
    N, S, C = feat.shape
    G = logits.shape[1]
    wts = torch.empty((N, G, S), dtype=torch.float32, device=feat.device)
    if pooled_out is not None:
        _chk(pooled_out)
        if tuple(pooled_out.shape) != (N, G * C):
            raise _l.VqfError("glimpse_pool_fwd: pooled_out must be (N, G*C)")
        pooled = pooled_out
    else:
        pooled = torch.empty((N, G * C), dtype=torch.float32, device=feat.device)
    fn = _lib().vqf_glimpse_pool_fwd_bf16 if bf else _lib().vqf_glimpse_pool_fwd
    _l.check(fn(_ptr(feat), _ptr(logits), N, S, C, G, int(bool(unit_softmax)), _ptr(wts), _ptr(pooled), _stream()),
             "vqf_glimpse_pool_fwd")
    return wts, pooled


def glimpse_pool_bwd(dpooled, feat, wts, unit_softmax, want_dfeat, dwts=None):
    _chk(dpooled, wts, dwts)
    N, S, C = feat.shape
    G = wts.shape[1]
    dlogits = torch.empty((N * S, G), dtype=torch.float32, device=feat.device)
    if feat.dtype == torch.bfloat16:             # bf16 feature storage: the tensor is data
        _chk_bf16(feat)
        if want_dfeat:
            raise _l.VqfError("glimpse_pool_bwd: a bf16 feature tensor cannot receive a gradient")
        _l.check(_lib().vqf_glimpse_pool_bwd_bf16(_ptr(dpooled), _ptr(dwts), _ptr(feat), _ptr(wts), N, S, C, G,
                                                  int(bool(unit_softmax)), _ptr(dlogits), _stream()),
                 "vqf_glimpse_pool_bwd_bf16")
        return dlogits, None
    _chk(feat)
    dfeat = torch.empty_like(feat) if want_dfeat else None
    _l.check(_lib().vqf_glimpse_pool_bwd(_ptr(dpooled), _ptr(dwts), _ptr(feat), _ptr(wts), N, S, C, G,
                                         int(bool(unit_softmax)), _ptr(dlogits), _ptr(dfeat), _stream()),
             "vqf_glimpse_pool_bwd")
    return dlogits, dfeat


def dropout(x, keep=None, seed=0, p_drop=0.5, out=None):
    _chk(x, out)
    y = torch.empty_like(x) if out is None else out          # in place (out is x) allowed
    _l.check(_lib().vqf_dropout_f32(_ptr(x), _keep_ptr(keep), int(seed), float(p_drop), x.numel(), _ptr(y),
                                    _stream()), "vqf_dropout_f32")
    return y


def dropout_bt(x, out, keep=None, seed=0, p_drop=0.0):
    """out[b,t,:] = x[b,t,:] * keep / (1 - p) for 3-D fp32 tensors of the same (B, T, H) shape with ANY strides on the first two
    axes (a transposed view in, a contiguous tensor out, or the other way round); the mask is indexed by (b, t, h)."""
    for t_ in (x, out):
        if not t_.is_cuda or t_.dtype != torch.float32 or t_.dim() != 3 or t_.stride(2) != 1:
            raise _l.VqfError("dropout_bt: 3-D fp32 GPU tensors with a contiguous last axis expected")
    if x.shape != out.shape:
        raise _l.VqfError("dropout_bt: shapes differ")
    B, T, H = x.shape
    _l.check(_lib().vqf_dropout_bt(_ptr(x), x.stride(0), x.stride(1), _keep_ptr(keep), int(seed), float(p_drop), B, T, H,
                                   _ptr(out), out.stride(0), out.stride(1), _stream()), "vqf_dropout_bt")
    return out


def tanh_dropout_fwd(a, b=None, keep=None, seed=0, p_drop=0.5, out=None):
    _chk(a, b, out)
    y = torch.empty_like(a) if out is None else out           # in place (out is a) allowed
    _l.check(_lib().vqf_tanh_dropout_fwd(_ptr(a), _ptr(b), _keep_ptr(keep), int(seed), float(p_drop),
                                         a.numel(), _ptr(y), _stream()), "vqf_tanh_dropout_fwd")
    return y


def tanh_dropout_bwd(dy, y, keep=None, seed=0, p_drop=0.5, out=None):
    _chk(dy, y, out)
    dx = torch.empty_like(y) if out is None else out          # in place (out is dy) allowed
    _l.check(_lib().vqf_tanh_dropout_bwd(_ptr(dy), _ptr(y), _keep_ptr(keep), int(seed), float(p_drop),
                                         y.numel(), _ptr(dx), _stream()), "vqf_tanh_dropout_bwd")
    return dx


def gate_tanh_sigmoid_fwd(a, b):
    """y = tanh(a) * sigmoid(b)   (modules.py:103-109)"""
    _chk(a, b)
    if a.shape != b.shape:
        raise _l.VqfError("gate_tanh_sigmoid: shapes differ")
    y = torch.empty_like(a)
    _l.check(_lib().vqf_gate_tanh_sigmoid_fwd(_ptr(a), _ptr(b), a.numel(), _ptr(y), _stream()), "vqf_gate_tanh_sigmoid_fwd")
    return y


def gate_tanh_sigmoid_bwd(dy, a, b):
    _chk(dy, a, b)
    da, db = torch.empty_like(a), torch.empty_like(b)
    _l.check(_lib().vqf_gate_tanh_sigmoid_bwd(_ptr(dy), _ptr(a), _ptr(b), a.numel(), _ptr(da), _ptr(db), _stream()),
             "vqf_gate_tanh_sigmoid_bwd")
    return da, db


def _chk2s(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise _l.VqfError("2-D fp32 GPU tensor with contiguous rows expected")


def tanh_dropout_fwd2d(a, b=None, keep=None, seed=0, p_drop=0.5, out=None):
    """y = dropout(tanh(a + b)) over 2-D operands whose rows may be strided (column blocks of wider buffers); out may be b."""
    _chk2s(a, b, out)
    R, W = a.shape
    if out is None:
        out = torch.empty((R, W), dtype=torch.float32, device=a.device)
    _l.check(_lib().vqf_tanh_dropout_fwd2d(_ptr(a), a.stride(0), _ptr(b), b.stride(0) if b is not None else 0, _keep_ptr(keep),
                                           int(seed), float(p_drop), R, W, _ptr(out), out.stride(0), _stream()),
             "vqf_tanh_dropout_fwd2d")
    return out


def tanh_dropout_bwd2d(dy, y, keep=None, seed=0, p_drop=0.5, out=None):
    """dx = dy * keep / (1 - p) * (1 - tanh^2), 2-D operands with strided rows; out may be dy."""
    _chk2s(dy, y, out)
    R, W = y.shape
    if out is None:
        out = torch.empty((R, W), dtype=torch.float32, device=y.device)
    _l.check(_lib().vqf_tanh_dropout_bwd2d(_ptr(dy), dy.stride(0), _ptr(y), y.stride(0), _keep_ptr(keep), int(seed),
                                           float(p_drop), R, W, _ptr(out), out.stride(0), _stream()), "vqf_tanh_dropout_bwd2d")
    return out


def relu_bwd_rank1(dx, y, wts, dpooled, L, scale, want_bias=True, out=None):
    """dXpre = (dx + wts[m] * dpooled[m // L]) * (y > 0 ? scale : 0) (+ its column sums): include/vqa_fusion.h
    vqf_relu_bwd_rank1_f32.  out may be dx (in place)."""
    _chk(dx, y, wts, dpooled, out)
    M, C = y.shape
    dpre = torch.empty_like(y) if out is None else out
    db = torch.empty(C, dtype=torch.float32, device=y.device) if want_bias else None
    ws = workspace(y.device, _lib().vqf_colsum_ws_bytes(M, C))
    _l.check(_lib().vqf_relu_bwd_rank1_f32(_ptr(dx), _ptr(y), _ptr(wts), _ptr(dpooled), int(L), float(scale), M, C, _ptr(dpre),
                                           _ptr(db), _ptr(ws), ws.numel(), _stream()), "vqf_relu_bwd_rank1_f32")
    return dpre, db


def scale_by_device_scalar(x, s):
    """x (M,W) fp32 times the one-element GPU tensor s (a loss's incoming gradient), no host read: vqf_scale_rows with one scale
    for all rows"""
    _chk(x, s)
    if x.dim() != 2 or s.numel() != 1:
        raise _l.VqfError("scale_by_device_scalar: (M,W) tensor and a one-element scale expected")
    M, W = x.shape
    out = torch.empty_like(x)
    _l.check(_lib().vqf_scale_rows(_ptr(x), _ptr(s), M, M, W, _ptr(out), _stream()), "vqf_scale_rows")
    return out


def _ptr_array(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def multi_copy(pairs):
    """[(src, dst), ...] (<= 8 contiguous fp32 GPU tensors of equal size per pair): dst = src, ONE launch"""
    for a, b in pairs:
        _chk(a, b)
        if a.numel() != b.numel():
            raise _l.VqfError("multi_copy: sizes differ")
    n = (ctypes.c_longlong * len(pairs))(*[a.numel() for a, _ in pairs])
    _l.check(_lib().vqf_multi_copy_f32(_ptr_array([a for a, _ in pairs]), _ptr_array([b for _, b in pairs]), n, len(pairs),
                                       _stream()), "vqf_multi_copy_f32")


def multi_add(triples):
    """[(a, b, out), ...] (<= 4): out = a + b, ONE launch"""
    for a, b, o in triples:
        _chk(a, b, o)
        if not (a.numel() == b.numel() == o.numel()):
            raise _l.VqfError("multi_add: sizes differ")
    n = (ctypes.c_longlong * len(triples))(*[a.numel() for a, _, _ in triples])
    _l.check(_lib().vqf_multi_add_f32(_ptr_array([t[0] for t in triples]), _ptr_array([t[1] for t in triples]),
                                      _ptr_array([t[2] for t in triples]), n, len(triples), _stream()), "vqf_multi_add_f32")


# ---------------------------------------------------------------------------
# HieCoAtten's ladder as streaming passes (csrc/hie.hip; include/vqa_fusion.h vqf_hie_*)
def hie_stream_supported(N, L, E, T):
    return bool(_lib().vqf_hie_stream_supported(int(N), int(L), int(E), int(T)))


def hie_chunks(N, L):
    return int(_lib().vqf_hie_chunks(int(N), int(L)))


def _chk_ntl(u, N, T, L):
    _chk(u)
    if u.numel() != N * T * L:
        raise _l.VqfError("hie: the (N, T, L) coefficient tensor has the wrong size")


def _part_ld(part, E):
    """`part` of the streaming passes: (S, N*T, E) contiguous partial slabs, or -- one chunk per sample -- the 2-D destination
    of the final sums itself (rows may be strided) -> row pitch"""
    if part.dim() == 3:
        _chk(part)
        return E
    _chk2s(part)
    return part.stride(0)


def hie_hv_fwd(a, C, V, drop, N, L, T, out, part):
    """out = dropout(tanh(a + C^T V)); part = the sums of C[t,l] a[l,:] over l: (S, N*T, E) per-chunk slabs, or (S == 1) the
    final (N*T, E) rows"""
    _chk2s(a, V, out)
    _chk_ntl(C, N, T, L)
    E = a.shape[1]
    keep, seed, p = drop
    _l.check(_lib().vqf_hie_hv_fwd(_ptr(a), a.stride(0), _ptr(C), _ptr(V), V.stride(0), _keep_ptr(keep), int(seed), float(p),
                                   N, L, E, T, _ptr(out), out.stride(0), _ptr(part), _part_ld(part, E), _stream()), "vqf_hie_hv_fwd")
    return out


def hie_head_bwd(hv, dl, w, C, drop, N, L, T, out, part, wpart, part_add=None):
    """part_add (one chunk per sample only): the T-row sums are written on top of these (N*T, E) rows.
    wpart: (S*N, >= E + 4) partial rows, may be a column block of a wider buffer"""
    _chk2s(hv, out, part_add, wpart)
    _chk(dl, w)
    if wpart.shape[1] < hv.shape[1] + 4:
        raise _l.VqfError("hie_head_bwd: wpart rows hold E + 4 floats")
    _chk_ntl(C, N, T, L)
    E = hv.shape[1]
    keep, seed, p = drop
    _l.check(_lib().vqf_hie_head_bwd(_ptr(hv), hv.stride(0), _ptr(dl), _ptr(w), _ptr(C), _keep_ptr(keep), int(seed), float(p),
                                     N, L, E, T, _ptr(out), out.stride(0), _ptr(part), _part_ld(part, E), _ptr(part_add),
                                     part_add.stride(0) if part_add is not None else 0, _ptr(wpart), wpart.stride(0), _stream()),
             "vqf_hie_head_bwd")
    return out


def _colpart(colpart, N, L, E):
    """(S*N, E) column block the streaming pass writes one partial column-sum row per workgroup into -> (pointer, pitch)"""
    if colpart is None:
        return ctypes.c_void_p(0), 0
    _chk2s(colpart)
    if colpart.shape != (hie_chunks(N, L) * N, E):
        raise _l.VqfError("hie: colpart must be (chunks * N, E)")
    return _ptr(colpart), colpart.stride(0)


def hie_rank_add(a, U, V, N, L, T, out, colpart=None):
    """colpart: (S*N, E) rows (a column block of a wider buffer) receiving each workgroup's column sums of `out`"""
    _chk2s(a, V, out)
    _chk_ntl(U, N, T, L)
    cp, ldcp = _colpart(colpart, N, L, a.shape[1])
    _l.check(_lib().vqf_hie_rank_add(_ptr(a), a.stride(0), _ptr(U), _ptr(V), V.stride(0), N, L, a.shape[1], T, _ptr(out),
                                     out.stride(0), cp, ldcp, _stream()), "vqf_hie_rank_add")
    return out


def hie_rank_left(U, V, z, N, L, T, out, part, colpart=None):
    _chk2s(V, z, out)
    _chk_ntl(U, N, T, L)
    cp, ldcp = _colpart(colpart, N, L, z.shape[1])
    _l.check(_lib().vqf_hie_rank_left(_ptr(U), _ptr(V), V.stride(0), _ptr(z), z.stride(0), N, L, z.shape[1], T, _ptr(out),
                                      out.stride(0), _ptr(part), _part_ld(part, z.shape[1]), cp, ldcp, _stream()),
             "vqf_hie_rank_left")
    return out


def hie_affinity_supported(N, L, E, T, pairs=1):
    return bool(_lib().vqf_hie_affinity_supported(int(N), int(L), int(E), int(T), int(pairs)))


def hie_affinity(x1, y1, N, L, T, x2=None, y2=None, epi=0, yprev=None, drop=(None, 0, 0.0), out=None):
    """out (N, T, L) = epi(x1 y1^T [+ x2 y2^T]) per sample: x* rows n*T + t, y* rows n*L + l (2-D, rows may be strided).
    epi 0: the sums; 1: dropout(tanh(.)) with `drop` = (keep | None, seed, p); 2: the backward of epi 1 given its output yprev."""
    _chk2s(x1, y1, x2, y2)
    E = x1.shape[1]
    if x1.shape[0] != N * T or y1.shape != (N * L, E) or (x2 is not None and (x2.shape != x1.shape or y2.shape != y1.shape)):
        raise _l.VqfError("hie_affinity: operand shapes")
    if out is None:
        out = torch.empty((N, T, L), dtype=torch.float32, device=x1.device)
    _chk(out)
    if yprev is not None:
        _chk_ntl(yprev, N, T, L)
    keep, seed, p = drop
    _l.check(_lib().vqf_hie_affinity(_ptr(x1), x1.stride(0), _ptr(y1), y1.stride(0), _ptr(x2), x2.stride(0) if x2 is not None else 0,
                                     _ptr(y2), y2.stride(0) if y2 is not None else 0, int(epi), _ptr(yprev), _keep_ptr(keep),
                                     int(seed), float(p), N, L, E, T, _ptr(out), _stream()), "vqf_hie_affinity")
    return out


def hie_slab_sum(part, out, add=None):
    """out[r,:] = (add[r,:] if add is given) + sum_s part[s, r, :]; part (S, R, W) contiguous, add / out 2-D, rows may be strided"""
    _chk(part)
    _chk2s(add, out)
    S, R, W = part.shape
    _l.check(_lib().vqf_hie_slab_sum(_ptr(part), S, R, W, _ptr(add), add.stride(0) if add is not None else 0, _ptr(out),
                                     out.stride(0), _stream()), "vqf_hie_slab_sum")
    return out


def softmax_rows_fwd(x):
    _chk(x)
    R, W = x.shape
    y = torch.empty_like(x)
    _l.check(_lib().vqf_softmax_rows_fwd(_ptr(x), R, W, _ptr(y), _stream()), "vqf_softmax_rows_fwd")
    return y


def softmax_rows_bwd(dy, y):
    _chk(dy, y)
    R, W = y.shape
    dx = torch.empty_like(y)
    _l.check(_lib().vqf_softmax_rows_bwd(_ptr(dy), _ptr(y), R, W, _ptr(dx), _stream()), "vqf_softmax_rows_bwd")
    return dx


def log_softmax_rows_fwd(x):
    _chk(x)
    R, W = x.shape
    y = torch.empty_like(x)
    _l.check(_lib().vqf_log_softmax_rows_fwd(_ptr(x), R, W, _ptr(y), _stream()), "vqf_log_softmax_rows_fwd")
    return y


def log_softmax_rows_bwd(dy, y):
    _chk(dy, y)
    R, W = y.shape
    dx = torch.empty_like(y)
    _l.check(_lib().vqf_log_softmax_rows_bwd(_ptr(dy), _ptr(y), R, W, _ptr(dx), _stream()), "vqf_log_softmax_rows_bwd")
    return dx


def _keep_ptr(keep):
    if keep is None:
        return ctypes.c_void_p(0)
    if not keep.is_cuda or keep.dtype != torch.uint8 or not keep.is_contiguous():
        raise _l.VqfError("keep mask must be a contiguous uint8 GPU tensor")
    return ctypes.c_void_p(keep.data_ptr())


def mfb_fuse_fwd(P, q, N, L, O, keep=None, seed=0, p_drop=0.0, cascade=None, want_zdrop=False, pbias=None,
                 normalise=True, r_bf16=None):
    """-> (Y normalised (N*L,O), norm (N), inv (N), zdrop or None).  pbias: projection bias added on load.
    P may be bf16 (written by gemm_bf16(out_bf16=True)).  normalise=False: the first output is R, the signed square roots
    WITHOUT the per-sample 1/norm (no vqf_scale_rows pass: the consumer applies inv in its GEMM epilogue).
    r_bf16 (bf16 P, normalise=False only): a list; it receives a (N*L, O rounded up to 32) bf16 copy of R with zero pad columns,
    written by the same launch (the operand of the consumer's bf16 GEMM)."""
    (_chk_bf16 if P.dtype == torch.bfloat16 else _chk)(P)
    _chk(q, cascade, pbias)
    dev = P.device
    R = torch.empty((N * L, O), dtype=torch.float32, device=dev)
    rowssq = torch.empty(N * L * 4, dtype=torch.float32, device=dev)       # four partial sums per row (one per wave)
    zdrop = torch.empty_like(P) if want_zdrop else None
    if P.dtype == torch.bfloat16:      # the projection itself stored in bf16 (bf16 mode of the image fusion)
        if cascade is not None or want_zdrop:
            raise _l.VqfError("mfb_fuse_fwd: a bf16 P is only available without cascade / zdrop")
        if r_bf16 is not None and not normalise and (O + 31) // 32 * 32 <= 1024:
            Rb = torch.empty((N * L, (O + 31) // 32 * 32), dtype=torch.bfloat16, device=dev)
            _l.check(_lib().vqf_mfb_fuse_fwd_pbf16_rb(_ptr(P), _ptr(pbias), _ptr(q), _keep_ptr(keep), int(seed), float(p_drop),
                                                      N, L, O, _ptr(R), ctypes.c_void_p(Rb.data_ptr()), Rb.shape[1], _ptr(rowssq),
                                                      _stream()), "vqf_mfb_fuse_fwd_pbf16_rb")
            r_bf16.append(Rb)
        else:
            _l.check(_lib().vqf_mfb_fuse_fwd_pbf16(_ptr(P), _ptr(pbias), _ptr(q), _keep_ptr(keep), int(seed), float(p_drop),
                                                   N, L, O, _ptr(R), _ptr(rowssq), _stream()), "vqf_mfb_fuse_fwd_pbf16")
    else:
        _l.check(_lib().vqf_mfb_fuse_fwd(_ptr(P), _ptr(pbias), _ptr(q), _ptr(cascade), _keep_ptr(keep), int(seed),
                                         float(p_drop), N, L, O, _ptr(R), _ptr(rowssq), _ptr(zdrop), _stream()),
                 "vqf_mfb_fuse_fwd")
    norm = torch.empty(N, dtype=torch.float32, device=dev)
    inv = torch.empty(N, dtype=torch.float32, device=dev)
    _l.check(_lib().vqf_l2_group_norm(_ptr(rowssq), N, 4 * L, _ptr(norm), _ptr(inv), _stream()),
             "vqf_l2_group_norm")
    if normalise:
        _l.check(_lib().vqf_scale_rows(_ptr(R), _ptr(inv), N * L, L, O, _ptr(R), _stream()), "vqf_scale_rows")
    return R, norm, inv, zdrop


def mfb_fuse_bwd(dY, Y, norm, inv, P, q, N, L, O, keep=None, seed=0, p_drop=0.0, cascade=None,
                 want_dbias=False, dzdrop=None, pbias=None, dp_bf16=False, lin=None):
    """-> (dP (N*L,5O) fp32 | bf16, dq (N,5O), dcascade or None, dbiasP or None).
    lin = (dlogits, lin) of the consumer's attention head: the UN-NORMALISED formulation -- Y is R (mfb_fuse_fwd(normalise=
    False)), dY is dYs = dY / norm, and sum(R * dYs) per sample comes from the head's (N*L, G) tensors instead of a
    rowdot pass over the (N*L, O) ones (vqf_l2_norm_bwd_coef_lin)."""
    (_chk_bf16 if P.dtype == torch.bfloat16 else _chk)(P)
    _chk(dY, Y, norm, inv, q, cascade, dzdrop, pbias)
    if P.dtype == torch.bfloat16 and not dp_bf16:
        raise _l.VqfError("mfb_fuse_bwd: a bf16 P comes with a bf16 dP")
    dev = P.device
    cA = torch.empty(N, dtype=torch.float32, device=dev)
    cB = torch.empty(N, dtype=torch.float32, device=dev)
    if lin is not None:
        dl, ln = lin
        _chk(dl, ln)
        unit = torch.empty(N, dtype=torch.float32, device=dev)
        _l.check(_lib().vqf_l2_norm_bwd_coef_lin(_ptr(dl), _ptr(ln), dl.shape[1], _ptr(norm), _ptr(inv), N, L, _ptr(cA),
                                                 _ptr(cB), _ptr(unit), _stream()), "vqf_l2_norm_bwd_coef_lin")
        inv = unit
    else:
        rowdot = torch.empty(N * L, dtype=torch.float32, device=dev)
        _l.check(_lib().vqf_rowdot(_ptr(Y), _ptr(dY), N * L, O, _ptr(rowdot), _stream()), "vqf_rowdot")
        _l.check(_lib().vqf_l2_norm_bwd_coef(_ptr(rowdot), _ptr(norm), _ptr(inv), N, L, _ptr(cA), _ptr(cB),
                                             _stream()), "vqf_l2_norm_bwd_coef")
    dq = torch.empty((N, POOL_K * O), dtype=torch.float32, device=dev)
    db = torch.empty(POOL_K * O, dtype=torch.float32, device=dev) if want_dbias else None
    ws = workspace(dev, _lib().vqf_mfb_fuse_bwd_ws_bytes(N, L, O))
    if dp_bf16:                       # bf16 mode of the image fusion: dP goes straight to the wgrad GEMM
        if cascade is not None or dzdrop is not None:
            raise _l.VqfError("mfb_fuse_bwd: bf16 dP is only available without cascade / dzdrop")
        dP = torch.empty(P.shape, dtype=torch.bfloat16, device=dev)
        if P.dtype == torch.bfloat16:
            _l.check(_lib().vqf_mfb_fuse_bwd_pbf16(_ptr(dY), _ptr(Y), _ptr(inv), _ptr(cA), _ptr(cB), _ptr(P), _ptr(pbias),
                                                   _ptr(q), _keep_ptr(keep), int(seed), float(p_drop), N, L, O, _ptr(dP),
                                                   _ptr(dq), _ptr(db), _ptr(ws), ws.numel(), _stream()),
                     "vqf_mfb_fuse_bwd_pbf16")
            return dP, dq, None, db
        _l.check(_lib().vqf_mfb_fuse_bwd_bf16dp(_ptr(dY), _ptr(Y), _ptr(inv), _ptr(cA), _ptr(cB), _ptr(P), _ptr(pbias),
                                                _ptr(q), _keep_ptr(keep), int(seed), float(p_drop), N, L, O, _ptr(dP),
                                                _ptr(dq), _ptr(db), _ptr(ws), ws.numel(), _stream()),
                 "vqf_mfb_fuse_bwd_bf16dp")
        return dP, dq, None, db
    dP = torch.empty_like(P)
    dc = torch.empty_like(P) if cascade is not None else None
    _l.check(_lib().vqf_mfb_fuse_bwd(_ptr(dY), _ptr(dzdrop), _ptr(Y), _ptr(inv), _ptr(cA), _ptr(cB), _ptr(P), _ptr(pbias), _ptr(q),
                                     _ptr(cascade), _keep_ptr(keep), int(seed), float(p_drop), N, L, O,
                                     _ptr(dP), _ptr(dq), _ptr(dc), _ptr(db), _ptr(ws), ws.numel(), _stream()),
             "vqf_mfb_fuse_bwd")
    return dP, dq, dc, db


def lstm_seq_supported(B, H):
    return bool(_lib().vqf_lstm_seq_supported(int(B), int(H)))


def _lstm_seq_ws(B, H, device):
    nbytes = int(_lib().vqf_lstm_seq_ws_bytes(int(B), int(H)))
    return workspace(device, nbytes), nbytes


def lstm_seq_fwd(xw, w_hh, bf16=False):
    """xw (S,B,4H) = x W_ih^T + biases, w_hh (4H,H) -> hs (S,B,H), cs (S,B,H), gates (S,B,4H) activated.
    bf16: the recurrent product takes bf16 operands (fp32 accumulate; bf16 mode)."""
    _chk(xw, w_hh)
    S, B, H4 = xw.shape
    H = H4 // 4
    hs = torch.empty((S, B, H), dtype=torch.float32, device=xw.device)
    cs = torch.empty_like(hs)
    gates = torch.empty_like(xw)
    ws, nb = _lstm_seq_ws(B, H, xw.device)
    _l.check(_lib().vqf_lstm_seq_fwd(_ptr(xw), _ptr(w_hh), S, B, H, _ptr(hs), _ptr(cs), _ptr(gates),
                                     1 if bf16 else 0, _ptr(ws), nb, _stream()), "vqf_lstm_seq_fwd")
    return hs, cs, gates


def lstm_seq_bwd(dhs, gates, cs, w_hh, bf16=False):
    """-> dgates (S,B,4H): gradient w.r.t. the gate pre-activations; w_hh (4H,H) as stored."""
    _chk(dhs, gates, cs, w_hh)
    S, B, H = dhs.shape
    dgates = torch.empty_like(gates)
    carry = torch.empty((B, H), dtype=torch.float32, device=dhs.device)
    ws, nb = _lstm_seq_ws(B, H, dhs.device)
    _l.check(_lib().vqf_lstm_seq_bwd(_ptr(dhs), _ptr(gates), _ptr(cs), _ptr(w_hh), S, B, H, _ptr(dgates),
                                     _ptr(carry), 1 if bf16 else 0, _ptr(ws), nb, _stream()), "vqf_lstm_seq_bwd")
    return dgates


def lstm_cell_fwd(gates, c_prev, c_out, h_out):
    """gates (B,4H) pre-activations -> activated in place; writes c_out, h_out (B,H)."""
    _chk(gates, c_prev, c_out, h_out)
    B, H4 = gates.shape
    _l.check(_lib().vqf_lstm_cell_fwd(_ptr(gates), _ptr(c_prev), B, H4 // 4, _ptr(c_out), _ptr(h_out), _stream()),
             "vqf_lstm_cell_fwd")


def lstm_step_supported(B, H):
    return bool(_lib().vqf_lstm_step_supported(int(B), int(H)))


def lstm_step_fwd(h_prev, w_hh, gates, c_prev, c_out, h_out):
    """one LSTM step in one launch: gates (B,4H) += h_prev W_hh^T, then the cell in the product's epilogue (include/vqa_fusion.h)"""
    _chk(h_prev, w_hh, gates, c_prev, c_out, h_out)
    B, H4 = gates.shape
    _l.check(_lib().vqf_lstm_step_fwd(_ptr(h_prev), _ptr(w_hh), _ptr(gates), _ptr(c_prev), B, H4 // 4, _ptr(c_out), _ptr(h_out),
                                      _stream()), "vqf_lstm_step_fwd")


def lstm_cell_bwd(dhs_t, dh_carry, gates, c_t, c_prev, first, dc_carry, dG):
    _chk(dhs_t, dh_carry, gates, c_t, c_prev, dc_carry, dG)
    B, H = dhs_t.shape
    _l.check(_lib().vqf_lstm_cell_bwd(_ptr(dhs_t), _ptr(dh_carry), _ptr(gates), _ptr(c_t), _ptr(c_prev), int(bool(first)),
                                      B, H, _ptr(dc_carry), _ptr(dG), _stream()), "vqf_lstm_cell_bwd")


# ---------------------------------------------------------------------------
# question-encoder front end (mfb.py:68)
def _chk_ids(ids):
    if not ids.is_cuda or ids.dtype != torch.int64 or not ids.is_contiguous():
        raise _l.VqfError("contiguous int64 GPU token ids expected")


def embed_tanh_fwd(weight, ids, tanh=True, time_major=False):
    """tanh(weight[ids]) (tanh=False: weight[ids]): weight (V,E) fp32, ids any shape int64 -> ids.shape + (E,);
    time_major (ids (N,Tq), tanh only): -> (Tq, N, E), the rows in the order the batch-major LSTM walks them"""
    _chk(weight)
    _chk_ids(ids)
    V, E = weight.shape
    T = ids.numel()
    if time_major:
        if not tanh or ids.dim() != 2:
            raise _l.VqfError("embed_tanh_fwd: the time-major form takes (N, Tq) ids and applies tanh")
        N, Tq = ids.shape
        out = torch.empty((Tq, N, E), dtype=torch.float32, device=weight.device)
        _l.check(_lib().vqf_embed_tanh_fwd_tm(_ptr(weight), ctypes.c_void_p(ids.data_ptr()), N, Tq, V, E, _ptr(out), _stream()),
                 "vqf_embed_tanh_fwd_tm")
        return out
    out = torch.empty(tuple(ids.shape) + (E,), dtype=torch.float32, device=weight.device)
    fn = _lib().vqf_embed_tanh_fwd if tanh else _lib().vqf_embed_fwd
    _l.check(fn(_ptr(weight), ctypes.c_void_p(ids.data_ptr()), T, V, E, _ptr(out), _stream()), "vqf_embed_fwd")
    return out


def embed_dropout_fwd(weight, ids, keep=None, seed=0, p_drop=0.5):
    """dropout(weight[ids]) in one launch: -> (ids.numel(), E); the mask of ops.dropout over that flat tensor"""
    _chk(weight)
    _chk_ids(ids)
    V, E = weight.shape
    out = torch.empty((ids.numel(), E), dtype=torch.float32, device=weight.device)
    _l.check(_lib().vqf_embed_dropout_fwd(_ptr(weight), ctypes.c_void_p(ids.data_ptr()), ids.numel(), V, E, _keep_ptr(keep), int(seed),
                                          float(p_drop), _ptr(out), _stream()), "vqf_embed_dropout_fwd")
    return out


def embed_dropout_bwd(dout, ids, V, keep=None, seed=0, p_drop=0.5):
    """-> dW (V, E) = segment sums of dout * keep / (1 - p) over the tokens of each id, one launch"""
    _chk(dout)
    _chk_ids(ids)
    E = dout.shape[-1]
    dW = torch.empty((V, E), dtype=torch.float32, device=dout.device)
    _l.check(_lib().vqf_embed_dropout_bwd(_ptr(dout), ctypes.c_void_p(ids.data_ptr()), ids.numel(), V, E, _keep_ptr(keep), int(seed),
                                          float(p_drop), _ptr(dW), _stream()), "vqf_embed_dropout_bwd")
    return dW


def embed_tanh_bwd(dout, out, ids, V, time_major=False):
    """-> dW (V,E): deterministic segment sum of dout * (1 - out^2) over the tokens of each id (out=None: of dout, the plain lookup);
    time_major: dout / out are (Tq, N, E) for ids (N, Tq)"""
    _chk(dout, out)
    _chk_ids(ids)
    E = dout.shape[-1]
    dW = torch.empty((V, E), dtype=torch.float32, device=dout.device)
    if time_major:
        N, Tq = ids.shape
        _l.check(_lib().vqf_embed_tanh_bwd_tm(_ptr(dout), _ptr(out), ctypes.c_void_p(ids.data_ptr()), N, Tq, V, E, _ptr(dW),
                                              _stream()), "vqf_embed_tanh_bwd_tm")
        return dW
    if out is None:
        _l.check(_lib().vqf_embed_bwd(_ptr(dout), ctypes.c_void_p(ids.data_ptr()), ids.numel(), V, E, _ptr(dW), _stream()),
                 "vqf_embed_bwd")
    else:
        _l.check(_lib().vqf_embed_tanh_bwd(_ptr(dout), _ptr(out), ctypes.c_void_p(ids.data_ptr()), ids.numel(), V, E, _ptr(dW),
                                           _stream()), "vqf_embed_tanh_bwd")
    return dW


# ---------------------------------------------------------------------------
# input staging (data_loader.py:30-32)
def feat_transpose(src, out=None, bf16=False):
    """src (N, D, L) fp32 as stored by the feature extractor -> (N, L, D) fp32 | bf16."""
    _chk(src)
    if src.dim() != 3:
        raise _l.VqfError("feat_transpose: (N, D, L) expected")
    N, D, L = src.shape
    dt = torch.bfloat16 if bf16 else torch.float32
    if out is None:
        out = torch.empty((N, L, D), dtype=dt, device=src.device)
    elif out.shape != (N, L, D) or out.dtype != dt or not out.is_contiguous() or not out.is_cuda:
        raise _l.VqfError("feat_transpose: out must be a contiguous (N, L, D) %s GPU tensor" % dt)
    _l.check(_lib().vqf_feat_transpose(_ptr(src), N, D, L, 1 if bf16 else 0, _ptr(out), _stream()),
             "vqf_feat_transpose")
    return out


# ---------------------------------------------------------------------------
# training-step tail (solver.py:25-29,91-94)
def _loss_ws(N, A, device):
    nbytes = int(_lib().vqf_loss_ws_bytes(N, A))
    return workspace(device, nbytes), nbytes


def ce_loss(logits, target, want_grad=True):
    """-> (loss (1,), dlogits | None): nn.CrossEntropyLoss() forward and gradient in one pass."""
    _chk(logits)
    if logits.dim() != 2 or target.shape != (logits.shape[0],):
        raise _l.VqfError("ce_loss: logits (N,A) and targets (N,) expected")
    if not target.is_cuda or target.dtype != torch.int64 or not target.is_contiguous():
        raise _l.VqfError("ce_loss: contiguous int64 GPU targets expected")
    N, A = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    d = torch.empty_like(logits) if want_grad else None
    ws, nb = _loss_ws(N, A, logits.device)
    _l.check(_lib().vqf_ce_loss(_ptr(logits), _ptr(target), N, A, _ptr(loss), _ptr(d), _ptr(ws), nb, _stream()),
             "vqf_ce_loss")
    return loss, d


def kldiv_loss(logp, target, want_grad=True):
    """-> (loss (1,), dlogp | None): nn.KLDivLoss() (element-wise mean) forward and gradient."""
    _chk(logp, target)
    if logp.dim() != 2 or target.shape != logp.shape:
        raise _l.VqfError("kldiv_loss: log-probs and targets of the same (N,A) shape expected")
    N, A = logp.shape
    loss = torch.empty(1, dtype=torch.float32, device=logp.device)
    d = torch.empty_like(logp) if want_grad else None
    ws, nb = _loss_ws(N, A, logp.device)
    _l.check(_lib().vqf_kldiv_loss(_ptr(logp), _ptr(target), N, A, _ptr(loss), _ptr(d), _ptr(ws), nb, _stream()),
             "vqf_kldiv_loss")
    return loss, d


def adam_step(params, grads, exp_avgs, exp_avg_sqs, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """In-place torch.optim.Adam update of every tensor in the lists (one launch per 32 tensors)."""
    n = len(params)
    if not (len(grads) == len(exp_avgs) == len(exp_avg_sqs) == n):
        raise _l.VqfError("adam_step: list lengths differ")
    if n == 0:
        return
    tab = (_l.AdamTensor * n)()
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs)):
        _chk(p, g, m, v)
        if not (p.numel() == g.numel() == m.numel() == v.numel()):
            raise _l.VqfError("adam_step: tensor %d: sizes differ" % i)
        tab[i].param, tab[i].grad = p.data_ptr(), g.data_ptr()
        tab[i].exp_avg, tab[i].exp_avg_sq, tab[i].n = m.data_ptr(), v.data_ptr(), p.numel()
    _l.check(_lib().vqf_adam_step(ctypes.cast(tab, ctypes.c_void_p), n, float(lr), float(beta1), float(beta2),
                                  float(eps), float(weight_decay), int(step), _stream()), "vqf_adam_step")


# ---------------------------------------------------------------------------
# HBM yardsticks (measurement only: bench.py's hbm_yardsticks)
def hbm_copy(src, dst, nt=False):
    """dst = src with the library's plain 16-B-per-lane grid-stride copy kernel (any dtype, same byte count)."""
    nbytes = src.numel() * src.element_size()
    if not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous()) or dst.numel() * dst.element_size() != nbytes:
        raise _l.VqfError("hbm_copy: contiguous GPU tensors of the same byte count expected")
    _l.check(_lib().vqf_hbm_copy(_ptr(src), _ptr(dst), nbytes, int(bool(nt)), _stream()), "vqf_hbm_copy")
    return dst


def hbm_read_sweep(src, nt=False, out=None):
    """-> per-workgroup sums (fp32) of a pure read stream over `src` (fp32, contiguous)."""
    _chk(src)
    nbytes = src.numel() * 4
    nb = int(_lib().vqf_hbm_read_sweep_blocks(nbytes))
    if out is None:
        out = torch.empty(nb, dtype=torch.float32, device=src.device)
    _l.check(_lib().vqf_hbm_read_sweep(_ptr(src), nbytes, int(bool(nt)), _ptr(out), _stream()), "vqf_hbm_read_sweep")
    return out


# ---------------------------------------------------------------------------
def prof_enable(on=True, min_mnk=0):
    """hipEvent brackets around the library's launches.  min_mnk > 0: only GEMM launches with M * N * K >= min_mnk (an
    event pair costs the stream ~6-10 us between two kernels, so a timed region brackets its dominant launches only)."""
    _lib().vqf_prof_filter(int(min_mnk))
    _lib().vqf_prof_enable(int(on))


def prof_reset():
    _lib().vqf_prof_reset()


def prof_report():
    """{kernel name: (launches, total_ms)} for kernels launched since the last reset."""
    lib = _lib()
    out = {}
    for i in range(lib.vqf_prof_num_kernels()):
        n = ctypes.c_longlong(0)
        ms = ctypes.c_double(0.0)
        _l.check(lib.vqf_prof_get(i, ctypes.byref(n), ctypes.byref(ms)), "vqf_prof_get")
        if n.value:
            out[lib.vqf_prof_kernel_name(i).decode()] = (n.value, ms.value)
    return out


def prof_shape(kernel, d0=-1, d1=-1, d2=-1):
    """(launches, total_ms) of the launches of `kernel` (profiler name as in prof_report(), or id) whose shape tag
    matches (GEMMs tag M, N, K; -1 = any) since the last reset."""
    lib = _lib()
    if isinstance(kernel, str):
        names = [lib.vqf_prof_kernel_name(i).decode() for i in range(lib.vqf_prof_num_kernels())]
        if kernel not in names:
            raise _l.VqfError("prof_shape: no kernel named %r" % kernel)
        kernel = names.index(kernel)
    n = ctypes.c_longlong(0)
    ms = ctypes.c_double(0.0)
    _l.check(lib.vqf_prof_get_shape(int(kernel), int(d0), int(d1), int(d2), ctypes.byref(n), ctypes.byref(ms)),
             "vqf_prof_get_shape")
    return n.value, ms.value


def prof_gemm(ta, tb, M, N, K):
    """(launches, total_ms) of the GEMM launches with this layout and shape since the last reset."""
    lib = _lib()
    n = ctypes.c_longlong(0)
    ms = ctypes.c_double(0.0)
    _l.check(lib.vqf_prof_get_shape(2 * int(bool(ta)) + int(bool(tb)), M, N, K, ctypes.byref(n),
                                    ctypes.byref(ms)), "vqf_prof_get_shape")
    return n.value, ms.value
