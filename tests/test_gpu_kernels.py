"""GPU parity tests of every C-ABI entry point against plain torch (fp64 on CPU).

All calls go through ctypes into libvqa_fusion.so (vqa_amd.ops).  Tolerances:
1e-5 relative for GEMMs (fp32 MFMA == fmaf chain; reference in fp64), 1e-5 for
the HBM-bound kernels; stated per test.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import vqa_amd
    vqa_amd.lib.load()
    return vqa_amd.ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    # rounded to fp32 so that the fp64 reference sees exactly the kernel's inputs
    return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * scale).float().double()


def _pos(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.1 + 0.9 * torch.rand(shape, generator=g, dtype=torch.float64)).float().double()


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


GEMM_SHAPES = [(1, 1, 1), (5, 7, 3), (33, 65, 31), (130, 257, 70), (128, 128, 32), (256, 384, 1000),
               (100, 5000, 64), (3, 2, 4096), (300, 200, 4100),
               # >= 1024 tiles of 256x256 with ragged edges: the large-tile LDS-DMA kernel (gemm_f32_big.hip)
               (8200, 8100, 48), (16400, 4100, 32)]


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_all_layouts(ops, ta, tb, M, N, K):
    A = _rand((K, M) if ta else (M, K), 1)
    B = _rand((K, N) if tb else (N, K), 2)
    bias = _rand((N,), 3)
    ref = (A.t() if ta else A) @ (B if tb else B.t()) + bias
    out = ops.gemm(A.float().cuda(), B.float().cuda(), ta=bool(ta), tb=bool(tb), bias=bias.float().cuda())
    assert out.shape == (M, N)
    assert _rel(out, ref) <= 2e-6 * max(1.0, np.sqrt(K) / 8)


def test_gemm_relu_accumulate_and_splitk(ops):
    M, N, K = 640, 512, 8192           # few tiles, deep K -> split-K path
    A, B = _rand((K, M), 4), _rand((K, N), 5)
    ref = A.t() @ B
    out = ops.gemm(A.float().cuda(), B.float().cuda(), ta=True, tb=True)
    assert _rel(out, ref) <= 1e-5
    out_ns = ops.gemm(A.float().cuda(), B.float().cuda(), ta=True, tb=True, splitk=False)
    assert _rel(out_ns, ref) <= 1e-5
    # relu + accumulate
    C0 = _rand((M, N), 6)
    out2 = C0.float().cuda().clone()
    ops.gemm(A.float().cuda(), B.float().cuda(), ta=True, tb=True, out=out2, accumulate=True, relu=True)
    assert _rel(out2, torch.relu(ref + C0)) <= 1e-5


@pytest.mark.parametrize("ta,tb,M,N,K,extras", [
    (0, 0, 512, 5000, 2000, "bias"),            # ques_proj1's shape (mfb.py:92) with a K the one-round 128x80 kernel does not take: 160 tiles, split-K
    (0, 1, 512, 2048, 5000, ""),                # its dgrad
    (1, 1, 640, 512, 8192, "accum+relu"),       # deep-K weight-gradient layout, accumulate + ReLU in the combine
    (0, 0, 300, 77, 4100, "bias+relu"),         # nothing aligned: the element-wise combine path, ragged tiles
    (0, 0, 256, 1000, 1024, "bias"),            # HieCoAtten's fc (hieCoAtten.py:54)
    (1, 1, 1000, 1024, 3584, ""),
])
def test_splitk_combined_in_the_launch_has_the_bits_of_the_two_launch_form(ops, ta, tb, M, N, K, extras):
    """Option gemm_splitk_fused = 1: split-K products of the 128x128-tile kernel combine their K slices INSIDE the GEMM launch
    (each tile's last-arriving workgroup sums the slabs in split order): bit-identical to the slabs + vqf_splitk_reduce form
    (option 0, and this kernel's default: the in-launch combine measured 3-20 us slower per product), run to run (the arrival
    counters are left at zero), on two streams at once, and no reduce launch is left.  The large-tile kernels use the same
    combine by default (tests/test_gpu_gemm_big.py)."""
    A = _rand((K, M) if ta else (M, K), 51).float().cuda()
    B = _rand((K, N) if tb else (N, K), 52).float().cuda()
    bias = _rand((N,), 53).float().cuda() if "bias" in extras else None
    C0 = _rand((M, N), 54).float().cuda()
    kw = dict(ta=bool(ta), tb=bool(tb), bias=bias, relu="relu" in extras)

    def run():
        if "accum" in extras:
            out = C0.clone()
            return ops.gemm(A, B, out=out, accumulate=True, **kw)
        return ops.gemm(A, B, **kw)
    with ops.options(gemm_splitk_fused=0):
        ops.prof_reset(); ops.prof_enable(True)
        two = run()
        torch.cuda.synchronize()
        ops.prof_enable(False)
        assert ops.prof_report().get("splitk_reduce", (0, 0))[0] == 1, "this shape does not split K: pick another one"
    assert torch.equal(run(), two)                      # the default of this kernel family is the two-launch form
    with ops.options(gemm_splitk_fused=1):
        ops.prof_reset(); ops.prof_enable(True)
        one = run()
        torch.cuda.synchronize()
        ops.prof_enable(False)
        rep = ops.prof_report()
        ops.prof_reset()
        assert "splitk_reduce" not in rep and sum(n for n, _ in rep.values()) == 1
        assert torch.equal(one, two)
        for _ in range(3):                                  # counters back at zero: the same bits again and again
            assert torch.equal(run(), two)
        side = torch.cuda.Stream()                          # two launches in flight: disjoint counter words of the ring
        side.wait_stream(torch.cuda.current_stream())
        outs = []
        for _ in range(4):
            with torch.cuda.stream(side):
                outs.append(run())
            outs.append(run())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert all(torch.equal(o, two) for o in outs)
    Ad, Bd = A.double().cpu(), B.double().cpu()
    ref = (Ad.t() if ta else Ad) @ (Bd if tb else Bd.t())
    if bias is not None:
        ref = ref + bias.double().cpu()
    if "accum" in extras:
        ref = ref + C0.double().cpu()
    if "relu" in extras:
        ref = torch.relu(ref)
    assert _rel(one, ref) <= 2e-6 * max(1.0, np.sqrt(K) / 8)


def test_gemm_unaligned_views(ops):
    """odd leading dimensions / offsets force the scalar-load path."""
    M, N, K = 70, 50, 45
    A = _rand((M, K + 3), 7)[:, 1:K + 1].contiguous()
    B = _rand((N, K), 8)
    out = ops.gemm(A.float().cuda(), B.float().cuda())
    assert _rel(out, A @ B.t()) <= 1e-5


def test_batched_gemm(ops):
    Bn, M, N, K = 5, 22, 196, 64
    A, B = _rand((Bn, M, K), 9), _rand((Bn, N, K), 10)
    assert _rel(ops.bgemm(A.float().cuda(), B.float().cuda()), A @ B.transpose(1, 2)) <= 1e-5
    A2, B2 = _rand((Bn, K, M), 11), _rand((Bn, K, N), 12)
    assert _rel(ops.bgemm(A2.float().cuda(), B2.float().cuda(), ta=True, tb=True),
                A2.transpose(1, 2) @ B2) <= 1e-5
    # K = 14 (tokens) contraction, not a multiple of 4
    A3, B3 = _rand((Bn, 64, 14), 13), _rand((Bn, 14, 196), 14)
    assert _rel(ops.bgemm(A3.float().cuda(), B3.float().cuda(), tb=True), A3 @ B3) <= 1e-5


def test_colsum_relu_bwd(ops):
    for M, N in [(1, 5), (300, 1000), (5000, 512), (777, 33)]:
        x = _rand((M, N), 20)
        assert _rel(ops.colsum(x.float().cuda()), x.sum(0)) <= 1e-5
    dx, y = _rand((300, 512), 21), _rand((300, 512), 22)
    dpre, db = ops.relu_bwd(dx.float().cuda(), y.float().cuda())
    ref = dx * (y > 0)
    assert _rel(dpre, ref) == 0.0
    assert _rel(db, ref.sum(0)) <= 1e-5


@pytest.mark.parametrize("M,Hh", [(14, 1024), (7168, 512), (333, 64), (5, 36)])
def test_att_logits_fwd_bwd(ops, M, Hh):
    hid = torch.relu(_rand((M, Hh), 30))
    w2, b2 = _rand((2, Hh), 31, 0.1), _rand((2,), 32)
    ref = hid @ w2.t() + b2
    out = ops.att_logits_fwd(hid.float().cuda(), w2.float().cuda(), b2.float().cuda())
    assert _rel(out, ref) <= 1e-5
    dl = _rand((M, 2), 33)
    dpre, dw2, db2, db1 = ops.att_logits_bwd(dl.float().cuda(), hid.float().cuda(), w2.float().cuda())
    ref_pre = (dl @ w2) * (hid > 0)
    assert _rel(dpre, ref_pre) <= 1e-5
    assert _rel(dw2, dl.t() @ hid) <= 1e-5
    assert _rel(db2, dl.sum(0)) <= 1e-5
    assert _rel(db1, ref_pre.sum(0)) <= 1e-5


@pytest.mark.parametrize("unit", [False, True])
@pytest.mark.parametrize("N,S,C", [(3, 14, 1024), (2, 196, 2048), (5, 22, 64), (1, 1, 8), (2, 7, 36)])
def test_glimpse_pool_fwd_bwd(ops, N, S, C, unit):
    feat = _rand((N, S, C), 40).requires_grad_()
    logits = _rand((N * S, 2), 41, 3.0).requires_grad_()
    lg = logits.view(N, S, 2).permute(0, 2, 1)
    w = torch.ones_like(lg) if unit else torch.softmax(lg, dim=2)
    pooled_ref = torch.einsum("ngs,nsc->ngc", w, feat).reshape(N, 2 * C)
    wts, pooled = ops.glimpse_pool_fwd(feat.detach().float().cuda(), logits.detach().float().cuda(), unit)
    assert _rel(wts, w) <= 1e-5
    assert _rel(pooled, pooled_ref) <= 1e-5
    dp = _rand((N, 2 * C), 42)
    pooled_ref.backward(dp)
    dlogits, dfeat = ops.glimpse_pool_bwd(dp.float().cuda(), feat.detach().float().cuda(), wts, unit, True)
    if unit:
        assert float(dlogits.abs().max()) == 0.0
    else:
        assert _rel(dlogits, logits.grad) <= 2e-5
    assert _rel(dfeat, feat.grad) <= 1e-5


def _ssqrt(s):
    return torch.sqrt(torch.relu(s)) - torch.sqrt(torch.relu(-s))


@pytest.mark.parametrize("N,L,O,use_keep,use_casc", [(3, 196, 1000, False, False), (2, 20, 1000, True, False),
                                                     (4, 1, 1000, True, False), (3, 1, 1000, True, True),
                                                     (2, 5, 8, True, True), (1, 3, 1000, False, False)])
def test_mfb_fuse_fwd_bwd(ops, N, L, O, use_keep, use_casc):
    W5 = 5 * O
    # |pooled sum| stays >= 0.05: all 5 products of a pooling window share one sign, so the
    # singular derivative 0.5/sqrt|s| does not amplify fp32 rounding in this kernel-level check
    P = _pos((N * L, W5), 50).requires_grad_()
    gsign = torch.sign(_rand((N, O), 56) + 1e-3).repeat_interleave(5, 1)
    q = (_pos((N, W5), 51) * gsign).requires_grad_()
    casc = _pos((N * L, W5), 52).requires_grad_() if use_casc else None
    keep = (torch.rand((N * L, W5), generator=torch.Generator().manual_seed(53)) >= 0.1).to(torch.uint8) \
        if use_keep else None
    z = P * q.repeat_interleave(L, 0)
    if use_casc:
        z = z * casc
    if use_keep:
        z = z * (keep.double() / (1.0 - np.float32(0.1).astype(np.float64)))
    S = z.view(N * L, O, 5).sum(2)
    R = _ssqrt(S)
    Y_ref = torch.nn.functional.normalize(R.view(N, -1)).view(N * L, O)
    cu = lambda t: None if t is None else t.detach().float().cuda()
    Y, norm, inv, zdrop = ops.mfb_fuse_fwd(cu(P), cu(q), N, L, O, keep=None if keep is None else keep.cuda(),
                                           p_drop=0.1 if use_keep else 0.0, cascade=cu(casc),
                                           want_zdrop=use_casc)
    assert _rel(Y, Y_ref) <= 1e-5
    assert _rel(norm, R.view(N, -1).norm(dim=1)) <= 1e-5
    if use_casc:
        assert _rel(zdrop, z) <= 1e-5
    dY = _rand((N * L, O), 54)
    dzx = _rand((N * L, W5), 55) if use_casc else None
    loss = (Y_ref * dY).sum() + ((z * dzx).sum() if use_casc else 0.0)
    loss.backward()
    dP, dq, dc, db = ops.mfb_fuse_bwd(cu(dY), Y, norm, inv, cu(P), cu(q), N, L, O,
                                      keep=None if keep is None else keep.cuda(),
                                      p_drop=0.1 if use_keep else 0.0, cascade=cu(casc), want_dbias=True,
                                      dzdrop=cu(dzx))
    assert _rel(dP, P.grad) <= 2e-5
    assert _rel(dq, q.grad) <= 2e-5
    assert _rel(db, P.grad.sum(0)) <= 2e-5
    if use_casc:
        assert _rel(dc, casc.grad) <= 2e-5


def test_mfb_fuse_zero_pool_has_zero_grad(ops):
    """relu'(0) = 0: a pooled sum that is exactly 0 contributes no gradient (and no NaN)."""
    N, L, O = 2, 3, 8
    P = _rand((N * L, 5 * O), 60).float().cuda()
    q = _rand((N, 5 * O), 61).float().cuda()
    P[:, :5] = 0.0
    Y, norm, inv, _ = ops.mfb_fuse_fwd(P, q, N, L, O)
    assert float(Y[:, 0].abs().max()) == 0.0
    dP, dq, _, _ = ops.mfb_fuse_bwd(torch.ones_like(Y), Y, norm, inv, P, q, N, L, O)
    assert torch.isfinite(dP).all() and torch.isfinite(dq).all()
    assert float(dP[:, :5].abs().max()) == 0.0


def test_mfb_fuse_philox_dropout(ops):
    """in-kernel Philox mask: P(drop) ~ p, same mask in forward and backward, seed-dependent."""
    N, L, O = 4, 50, 1000
    P = torch.ones((N * L, 5 * O), device="cuda")
    q = torch.ones((N, 5 * O), device="cuda")
    _, _, _, z = ops.mfb_fuse_fwd(P, q, N, L, O, seed=1234, p_drop=0.1, want_zdrop=True)
    kept = (z != 0)
    frac = 1.0 - kept.float().mean().item()
    assert abs(frac - 0.1) < 2e-3, frac
    assert torch.allclose(z[kept], torch.full_like(z[kept], 1.0 / 0.9), rtol=1e-6)
    _, _, _, z2 = ops.mfb_fuse_fwd(P, q, N, L, O, seed=1234, p_drop=0.1, want_zdrop=True)
    assert torch.equal(z, z2)
    _, _, _, z3 = ops.mfb_fuse_fwd(P, q, N, L, O, seed=1235, p_drop=0.1, want_zdrop=True)
    assert not torch.equal(z, z3)
    # backward sees the same mask: dP is zero exactly where the forward dropped
    Y, norm, inv, _ = ops.mfb_fuse_fwd(P, q, N, L, O, seed=1234, p_drop=0.1)
    dY = torch.ones_like(Y)
    dP, _, _, _ = ops.mfb_fuse_bwd(dY, Y, norm, inv, P, q, N, L, O, seed=1234, p_drop=0.1)
    assert torch.equal(dP != 0, kept) or float(((dP != 0) ^ kept).float().mean()) < 1e-3


def test_l2_norm_clamp_branch(ops):
    """all-zero sample: F.normalize clamps the norm at eps; backward has no projection term."""
    N, L, O = 2, 2, 8
    P = _rand((N * L, 5 * O), 70).float().cuda()
    q = _rand((N, 5 * O), 71).float().cuda()
    q[1] = 0.0
    Y, norm, inv, _ = ops.mfb_fuse_fwd(P, q, N, L, O)
    assert float(norm[1]) == 0.0 and float(Y[L:].abs().max()) == 0.0
    dP, dq, _, _ = ops.mfb_fuse_bwd(torch.ones_like(Y), Y, norm, inv, P, q, N, L, O)
    assert torch.isfinite(dP).all() and torch.isfinite(dq).all()


def test_errors_are_loud(ops):
    import vqa_amd
    with pytest.raises(vqa_amd.VqfError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))                 # CPU tensors: no fallback
    with pytest.raises(vqa_amd.VqfError):
        ops.gemm(torch.zeros(4, 4, device="cuda"), torch.zeros(4, 5, device="cuda"))
    with pytest.raises(vqa_amd.VqfError):
        ops.mfb_fuse_fwd(torch.zeros(2, 35, device="cuda"), torch.zeros(2, 35, device="cuda"), 2, 1, 7)
    with pytest.raises(vqa_amd.VqfError):
        ops.glimpse_pool_fwd(torch.zeros(1, 2000, 4, device="cuda"), torch.zeros(2000, 2, device="cuda"), False)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0)])
def test_gemm_large_tile_kernel_splitk_relu_bias(ops, ta, tb):
    """256x256-tile kernel (1024 tiles), bias and ReLU, K = 2048; fp64 reference, 2e-5."""
    M, N, K = 8192, 8192, 2048
    A = _rand((K, M) if ta else (M, K), 21, 0.5)
    B = _rand((K, N) if tb else (N, K), 22, 0.5)
    bias = _rand((N,), 23)
    ref = torch.relu((A.t() if ta else A) @ (B if tb else B.t()) + bias)
    out = ops.gemm(A.float().cuda(), B.float().cuda(), ta=bool(ta), tb=bool(tb), bias=bias.float().cuda(), relu=True)
    assert _rel(out, ref) <= 2e-5
    out2 = ops.gemm(A.float().cuda(), B.float().cuda(), ta=bool(ta), tb=bool(tb), bias=bias.float().cuda(), relu=True)
    assert torch.equal(out, out2)                       # fixed-order slab reduction: run-to-run identical


# small-M products (gemm_f32_wave.hip: one 32x64 tile per wave, WK = 1 whole K per wave / WK = 4 in-workgroup K split)
WAVE_SHAPES = [  # (tb, M, N, K, what)
    (0, 512, 4096, 1024, "LSTM forward recurrent product: 1024 wave tiles, WK = 1"),
    (1, 512, 1024, 4096, "LSTM backward recurrent product: 256 wave tiles x 4 K ranges, WK = 4"),
    (0, 500, 4090, 1024, "ragged M and N (row clamp / guarded stores, odd N -> scalar tail)"),
    (1, 509, 1020, 4096, "ragged, K-major B, WK = 4"),
    (0, 256, 2048, 2048, "WK = 4 with a K-contiguous B"),
    (1, 1000, 2048, 512, "WK = 1 with a K-major B, M up to 1024"),
]


@pytest.mark.parametrize("tb,M,N,K,what", WAVE_SHAPES)
def test_gemm_small_m_wave_kernel(ops, tb, M, N, K, what):
    A, B, bias = _rand((M, K), 31), _rand((K, N) if tb else (N, K), 32), _rand((N,), 33)
    ref = A @ (B if tb else B.t())
    tol = 2e-6 * max(1.0, np.sqrt(K) / 8)
    Ad, Bd = A.float().cuda(), B.float().cuda()
    out = ops.gemm(Ad, Bd, tb=bool(tb))
    assert _rel(out, ref) <= tol, (what, _rel(out, ref))
    with ops.options(gemm_f32_wave=0):                         # the 128x128 split-K path on the same operands
        old = ops.gemm(Ad, Bd, tb=bool(tb))
    assert _rel(old, ref) <= tol and not torch.equal(old, out), "the wave kernel did not take this shape"
    n0 = ops.stat("gemm_f32_wave")
    with ops.options(gemm_f32_wave=1):                         # round 2's form (every wave its own B slab): the same k order, the same bits
        assert torch.equal(out, ops.gemm(Ad, Bd, tb=bool(tb))) and ops.stat("gemm_f32_wave") == n0 + 1
    assert torch.equal(out, ops.gemm(Ad, Bd, tb=bool(tb)))      # fixed-order in-workgroup K reduction
    # bias + relu, and accumulate into a row-strided output view (LstmBatchFn adds into xw[t])
    out2 = ops.gemm(Ad, Bd, tb=bool(tb), bias=bias.float().cuda(), relu=True)
    assert _rel(out2, torch.relu(ref + bias)) <= tol
    C0 = _rand((M, N + 3), 34)
    wide = C0.float().cuda().clone()
    ops.gemm(Ad, Bd, tb=bool(tb), out=wide[:, 1:N + 1], accumulate=True)
    assert _rel(wide[:, 1:N + 1], ref + C0[:, 1:N + 1]) <= tol
    assert torch.equal(wide[:, 0].cpu(), C0[:, 0].float()) and torch.equal(wide[:, N + 1:].cpu(), C0[:, N + 1:].float())


def test_log_softmax_rows_vs_torch_fp64(ops):
    """vqf_log_softmax_rows_fwd/bwd (the classifier tail of MHBCoAtt / MHB, mhb_coAtt.py:149-151) vs torch in fp64."""
    for R, W in ((512, 1000), (3, 7), (5, 64), (2, 1)):
        x = _rand((R, W), 81, 6.0)
        g = _rand((R, W), 82)
        xr = x.clone().requires_grad_(True)
        ref = torch.log_softmax(xr, dim=1)
        ref.backward(g)
        y = ops.log_softmax_rows_fwd(x.float().cuda())
        assert float((y.double().cpu() - ref.detach()).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
        dx = ops.log_softmax_rows_bwd(g.float().cuda(), y)
        assert float((dx.double().cpu() - xr.grad).abs().max()) <= 1e-5 * max(1.0, float(xr.grad.abs().max()))
        assert torch.allclose(y.exp().sum(1).cpu(), torch.ones(R), atol=1e-5)


@pytest.mark.parametrize("O", [1000, 1004, 8])
def test_mfb_fuse_philox_16bit_draws(ops, O):
    """One Philox4x32-10 call per 8 elements, a 16-bit draw each: drop rate, forward == backward mask, and no
    correlation between neighbours; O = 1004 makes rows start at element offsets = 4 (mod 8) (the window's other phase)."""
    N, L = 3, 64
    P = torch.ones((N * L, 5 * O), device="cuda")
    q = torch.ones((N, 5 * O), device="cuda")
    _, _, _, z = ops.mfb_fuse_fwd(P, q, N, L, O, seed=99, p_drop=0.25, want_zdrop=True)
    kept = (z != 0)
    k = kept.float()
    n = k.numel()
    assert abs(1.0 - float(k.mean()) - 0.25) < 4.0 * (0.25 * 0.75 / n) ** 0.5 + 1e-4
    for lag in (1, 2, 3, 4, 7, 8):                         # neighbours within and across a call's 8 draws
        a, b = k[:, :-lag].flatten(), k[:, lag:].flatten()
        corr = float(((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std() + 1e-12))
        assert abs(corr) < 5.0 / a.numel() ** 0.5 + 1e-3, (lag, corr)
    Y, norm, inv, _ = ops.mfb_fuse_fwd(P, q, N, L, O, seed=99, p_drop=0.25)
    dP, _, _, _ = ops.mfb_fuse_bwd(torch.ones_like(Y), Y, norm, inv, P, q, N, L, O, seed=99, p_drop=0.25)
    same = float(((dP != 0) == kept).float().mean())
    assert same > 0.999, same                               # (dP can be exactly 0 where the pooled sum is 0)


@pytest.mark.parametrize("pbf16", [False, True])
def test_mfb_fuse_access_variants_give_the_same_bits(ops, pbf16):
    """The three access forms of the fusion kernels (library option fuse_coal: 0 direct / strided, 1 LDS-transposed with the
    forward's register prefetch, default LDS-transposed without it) do the same arithmetic per element: bit-identical R / Y,
    dP, dq and bias gradient, fp32 and bf16 projection storage, with Philox dropout, at a ragged shape and the reference's O."""
    N, L, O = 5, 23, 1000
    g = torch.Generator(device="cuda").manual_seed(17)
    P = torch.randn((N * L, 5 * O), device="cuda", generator=g)
    q = torch.randn((N, 5 * O), device="cuda", generator=g)
    pb = torch.randn((5 * O,), device="cuda", generator=g)
    dY = torch.randn((N * L, O), device="cuda", generator=g)
    Pm = P.to(torch.bfloat16) if pbf16 else P
    res = {}
    for v in (0, 1, None):
        with ops.options(fuse_coal=v):
            Y, norm, inv, _ = ops.mfb_fuse_fwd(Pm, q, N, L, O, seed=4242, p_drop=0.1, pbias=pb)
            out = ops.mfb_fuse_bwd(dY, Y, norm, inv, Pm, q, N, L, O, seed=4242, p_drop=0.1, want_dbias=True, pbias=pb,
                                   dp_bf16=pbf16)
        res[v] = (Y.clone(), out[0].clone(), out[1].clone(), out[3].clone())
    for v in (1, None):
        for a, b in zip(res[0], res[v]):
            assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), v


@pytest.mark.parametrize("T,V,E", [(7168, 1000, 300), (5000, 3, 300), (130, 50, 7), (64, 9, 1024),
                                   (7168, 20000, 300),      # a dataset-sized vocabulary (~15-20 k words, utils.py): 10 rows per workgroup
                                   (3000, 5001, 512)])       # 3 rows per workgroup, V not a multiple of it
def test_embed_tanh_fwd_bwd_vs_torch_fp64(ops, T, V, E):
    """tanh(Embedding(q)) (mfb.py:68) and its weight gradient: vs torch in fp64; the backward is a token-ordered segment
    sum (bit-reproducible), writes every vocabulary row (zeros for unused ids) and handles ids that repeat thousands of
    times (T = 5000 tokens over 3 ids: three rounds of the 2048-id match list)."""
    g = torch.Generator().manual_seed(41)
    W = torch.randn((V, E), generator=g)
    ids = torch.randint(0, V, (T,), generator=g)
    if V > 20:
        ids[ids == 5] = 6                                   # id 5 never occurs: its gradient row must be exact zeros
    dout = torch.randn((T, E), generator=g)
    W64 = W.double().requires_grad_(True)
    ref = torch.tanh(torch.nn.functional.embedding(ids, W64))
    ref.backward(dout.double())
    out = ops.embed_tanh_fwd(W.cuda(), ids.cuda())
    assert out.shape == (T, E) and _rel(out, ref.detach()) <= 2e-6
    dW = ops.embed_tanh_bwd(dout.cuda(), out, ids.cuda(), V)
    assert _rel(dW, W64.grad) <= 2e-6 * max(1.0, np.sqrt(T / V) / 4)
    if V > 20:
        assert float(dW[5].abs().max()) == 0.0
    assert torch.equal(dW, ops.embed_tanh_bwd(dout.cuda(), out, ids.cuda(), V))
    # (N, T) id tensors keep their shape; ids outside [0, V) select nothing
    ids2 = ids[: (T // 2) * 2].view(2, -1).clone()
    ids2[0, 0], ids2[1, 1] = -1, V
    o2 = ops.embed_tanh_fwd(W.cuda(), ids2.cuda())
    assert o2.shape == (2, T // 2, E) and float(o2[0, 0].abs().max()) == 0.0 and float(o2[1, 1].abs().max()) == 0.0
    ok = (ids2 >= 0) & (ids2 < V)
    assert _rel(o2[ok.cuda()], torch.tanh(W.double()[ids2[ok]])) <= 2e-6


def test_embed_tanh_module_path_matches_torch_embedding(ops):
    """functions.embed_tanh on an nn.Embedding: same values and weight gradient as torch.tanh(embedding(q)); an embedding with
    a padding_idx stays on torch (its row gets no gradient there)."""
    import vqa_amd
    fns = vqa_amd.functions
    emb = torch.nn.Embedding(60, 24).cuda()
    q = torch.randint(0, 60, (5, 9), generator=torch.Generator().manual_seed(3)).cuda()
    up = torch.randn((5, 9, 24), generator=torch.Generator().manual_seed(4)).cuda()
    y = fns.embed_tanh(emb, q)
    y.backward(up)
    g_hip, emb.weight.grad = emb.weight.grad.clone(), None
    y2 = torch.tanh(emb(q))
    y2.backward(up)
    assert _rel(y, y2.detach().double().cpu()) <= 2e-6 and _rel(g_hip, emb.weight.grad.double().cpu()) <= 2e-6
    assert type(y.grad_fn).__name__.startswith("EmbedTanhFn")
    pad = torch.nn.Embedding(60, 24, padding_idx=0).cuda()
    assert not type(fns.embed_tanh(pad, q).grad_fn).__name__.startswith("EmbedTanhFn")


def test_embed_tanh_time_major_form(ops):
    """vqf_embed_tanh_fwd_tm / _bwd_tm: ids (N,Tq) as the reference holds them, output rows in (Tq,N) order -- bitwise the
    batch-major lookup transposed; the weight gradient vs torch in fp64 (its token order differs from the batch-major form's, so
    only the rounding may); functions.embed_tanh(time_major=True) end to end."""
    import vqa_amd
    fns = vqa_amd.functions
    g = torch.Generator().manual_seed(43)
    N, Tq, V, E = 37, 14, 50, 300
    W = torch.randn((V, E), generator=g).cuda()
    ids = torch.randint(0, V, (N, Tq), generator=g).cuda()
    out_tm = ops.embed_tanh_fwd(W, ids, True, time_major=True)
    assert out_tm.shape == (Tq, N, E)
    assert torch.equal(out_tm, ops.embed_tanh_fwd(W, ids).transpose(0, 1).contiguous())
    dout = torch.randn((Tq, N, E), generator=g).cuda()
    dW = ops.embed_tanh_bwd(dout, out_tm, ids, V, time_major=True)
    W64 = W.double().cpu().requires_grad_(True)
    torch.tanh(torch.nn.functional.embedding(ids.cpu(), W64)).backward(dout.transpose(0, 1).double().cpu())
    assert _rel(dW, W64.grad) <= 2e-6 and torch.equal(dW, ops.embed_tanh_bwd(dout, out_tm, ids, V, time_major=True))
    emb = torch.nn.Embedding(V, E).cuda()
    y = fns.embed_tanh(emb, ids, time_major=True)
    y.backward(dout)
    g_hip, emb.weight.grad = emb.weight.grad.clone(), None
    torch.tanh(emb(ids)).transpose(0, 1).backward(dout)
    assert _rel(g_hip, emb.weight.grad.double().cpu()) <= 2e-6
    pad = torch.nn.Embedding(V, E, padding_idx=0).cuda()                 # torch fallback keeps the layout contract
    assert fns.embed_tanh(pad, ids, time_major=True).shape == (Tq, N, E)


def test_embed_plain_lookup_is_exact(ops):
    """functions.embed (hieCoAtten.py:27, networks.py:23,56, mhb_coAtt.py:181): the lookup is a copy (bitwise torch's); the weight
    gradient is the token-ordered segment sum, vs torch in fp64, and bit-reproducible."""
    import vqa_amd
    fns = vqa_amd.functions
    emb = torch.nn.Embedding(300, 40).cuda()
    q = torch.randint(0, 300, (7, 22), generator=torch.Generator().manual_seed(5)).cuda()
    up = torch.randn((7, 22, 40), generator=torch.Generator().manual_seed(6)).cuda()
    y = fns.embed(emb, q)
    assert type(y.grad_fn).__name__.startswith("EmbedTanhFn") and torch.equal(y.detach(), emb(q).detach())
    y.backward(up)
    ref = torch.zeros((300, 40), dtype=torch.float64).index_add_(0, q.cpu().reshape(-1), up.double().cpu().reshape(-1, 40))
    assert _rel(emb.weight.grad, ref) <= 1e-6
    assert torch.equal(emb.weight.grad, ops.embed_tanh_bwd(up, None, q, 300))


def test_linear2_equals_the_linear_of_the_concatenation():
    """Linear2Fn (mhb_coAtt.py:147-148,213-214: linear_pred(cat((y2, y3), 1)) without the concatenated tensor): output and
    every gradient against fp64 of the concatenated form, and against LinearFn on a torch.cat of the same operands."""
    import vqa_amd
    F = vqa_amd.functions
    g = torch.Generator().manual_seed(77)
    M, K1, K2, N = 512, 1000, 1000, 1000
    x1, x2 = torch.randn((M, K1), generator=g).cuda(), torch.randn((M, K2), generator=g).cuda()
    w, b = (torch.randn((N, K1 + K2), generator=g) * 0.03).cuda(), torch.randn(N, generator=g).cuda()
    dy = torch.randn((M, N), generator=g).cuda()
    leaves = [t.clone().requires_grad_() for t in (x1, x2, w, b)]
    y = F.Linear2Fn.apply(*leaves)
    y.backward(dy)
    l64 = [t.double().requires_grad_() for t in (x1, x2, w, b)]
    y64 = torch.cat((l64[0], l64[1]), 1) @ l64[2].t() + l64[3]
    y64.backward(dy.double())
    nrel = lambda a, r: float((a.double() - r).norm() / r.norm())
    assert nrel(y.detach(), y64.detach()) <= 2e-6 * (K1 + K2) ** 0.5 / 8
    for a, r, k in zip(leaves, l64, (N, N, M, M)):
        assert nrel(a.grad, r.grad) <= 2e-6 * max(1.0, k ** 0.5 / 8), (tuple(a.shape), nrel(a.grad, r.grad))
    xc = torch.cat((x1, x2), 1).requires_grad_()
    wc, bc = w.clone().requires_grad_(), b.clone().requires_grad_()
    yc = F.LinearFn.apply(xc, wc, bc)
    yc.backward(dy)
    assert nrel(y.detach(), yc.detach().double()) <= 1e-6 and nrel(leaves[2].grad, wc.grad.double()) <= 1e-6
    assert torch.equal(leaves[3].grad, bc.grad)


def test_gate_tanh_sigmoid_kernel_vs_fp64():
    """vqf_gate_tanh_sigmoid_fwd / _bwd (modules.py:103-109, Nonlinear_layer's tanh(W1 x) * sigmoid(W2 x)) against fp64."""
    import vqa_amd
    F = vqa_amd.functions
    g = torch.Generator().manual_seed(78)
    a, b = (torch.randn((300, 516), generator=g) * 3).cuda(), (torch.randn((300, 516), generator=g) * 3).cuda()
    dy = torch.randn((300, 516), generator=g).cuda()
    la, lb = a.clone().requires_grad_(), b.clone().requires_grad_()
    y = F.GateFn.apply(la, lb)
    y.backward(dy)
    a64, b64 = a.double().requires_grad_(), b.double().requires_grad_()
    y64 = torch.tanh(a64) * torch.sigmoid(b64)
    y64.backward(dy.double())
    for got, ref in ((y.detach(), y64.detach()), (la.grad, a64.grad), (lb.grad, b64.grad)):
        assert float((got.double() - ref).abs().max()) <= 5e-7 * max(1.0, float(ref.abs().max()))
    from importlib import import_module
    m = import_module("vqa-attention-networks_amd.host.modules").Nonlinear_layer(516).cuda()
    x = torch.randn((7, 12, 516), generator=g).cuda()
    out = m(x)
    ref = torch.tanh(m.fc1(x)) * torch.sigmoid(m.fc2(x))
    assert float((out - ref).abs().max()) <= 2e-5
