"""bf16 path (BASELINE config 3): vqf_cast_f32_bf16 + vqf_gemm_bf16 in all four layouts vs fp64
matmuls of the SAME bf16-rounded values (so only fp32 accumulation differs: tolerance 2e-5), and
vs the fp32 inputs (bf16 rounding: tolerance 2e-2, stated)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import vqa_amd
    vqa_amd.lib.load()
    return vqa_amd.ops


def _r(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_cast_rounds_to_nearest_even_and_pads(ops):
    x = _r((37, 1000), 1)
    y = ops.cast_bf16(x.cuda(), pad_to=32)
    assert y.shape == (37, 1024) and y.dtype == torch.bfloat16
    assert torch.equal(y[:, :1000].cpu(), x.to(torch.bfloat16))
    assert float(y[:, 1000:].abs().max()) == 0.0
    big = _r((70000, 16), 2)                      # > 65535 rows: slab loop
    assert torch.equal(ops.cast_bf16(big.cuda()).cpu(), big.to(torch.bfloat16))


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (136, 264, 72), (392, 1000, 2048), (8, 8, 8),
                                   (1000, 1024, 392), (640, 512, 8192)])
def test_gemm_bf16_all_layouts(ops, ta, tb, M, N, K):
    A = _r((K, M) if ta else (M, K), 3).to(torch.bfloat16)
    B = _r((K, N) if tb else (N, K), 4).to(torch.bfloat16)
    bias = _r((N,), 5)
    ref = (A.double().t() if ta else A.double()) @ (B.double() if tb else B.double().t()) + bias.double()
    out = ops.gemm_bf16(A.cuda(), B.cuda(), ta=bool(ta), tb=bool(tb), bias=bias.cuda())
    assert out.shape == (M, N) and out.dtype == torch.float32
    assert _rel(out, ref) <= 2e-5 * max(1.0, np.sqrt(K) / 16)
    out2 = ops.gemm_bf16(A.cuda(), B.cuda(), ta=bool(ta), tb=bool(tb), bias=bias.cuda(), relu=True)
    assert _rel(out2, torch.relu(ref)) <= 2e-5 * max(1.0, np.sqrt(K) / 16)


def test_att_logits_bwd_bf16_rows_are_the_cast_of_the_fp32_rows(ops):
    """vqf_att_logits_bwd_rowscale_obf16 (config 3: co_att_conv1's gradient rows stored as bf16 by the kernel that makes them):
    bit-identical to vqf_cast_f32_bf16 of the fp32 form's rows, every fp32 sum identical; ragged last row block."""
    g = torch.Generator().manual_seed(5)
    M, Hh, L = 7 * 196 + 3, 512, 196
    hid = torch.relu(torch.randn((M, Hh), generator=g)).cuda()
    dl = torch.randn((M, 2), generator=g).cuda()
    w2 = torch.randn((2, Hh), generator=g).cuda()
    inv = (torch.rand((M + L - 1) // L, generator=g) + 0.5).cuda()
    f32 = ops.att_logits_bwd(dl, hid, w2, relu_mask=True, rowscale=inv, rows_per_scale=L)
    b16 = ops.att_logits_bwd(dl, hid, w2, relu_mask=True, rowscale=inv, rows_per_scale=L, out_bf16=True)
    assert b16[0].dtype == torch.bfloat16 and torch.equal(b16[0].view(torch.int16), ops.cast_bf16(f32[0]).view(torch.int16))
    for a, b in zip(f32[1:], b16[1:]):
        assert torch.equal(a, b)
    with pytest.raises(Exception):
        ops.att_logits_bwd(dl, hid, w2, relu_mask=False, out_bf16=True)


def test_gemm_bf16_padded_k_and_vs_fp32(ops):
    """K = 1000 padded to 1024 by the cast (the co_att_conv1 shape); error vs un-rounded fp32 inputs."""
    M, N, K = 392, 512, 1000
    X, W = _r((M, K), 6), _r((N, K), 7, 0.05)
    Xb, Wb = ops.cast_bf16(X.cuda(), pad_to=32), ops.cast_bf16(W.cuda(), pad_to=32)
    out = ops.gemm_bf16(Xb, Wb, K=1024)
    ref = X.double() @ W.double().t()
    assert _rel(out, ref) <= 2e-2                  # bf16 storage: 8 mantissa bits
    ref_b = Xb.double().cpu() @ Wb.double().cpu().t()
    assert _rel(out, ref_b) <= 2e-5


def test_gemm_bf16_rejects_unsupported_shapes(ops):
    import vqa_amd
    a = torch.zeros((16, 20), dtype=torch.bfloat16, device="cuda")      # K = 20 not a multiple of 8
    b = torch.zeros((16, 20), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(vqa_amd.VqfError):
        ops.gemm_bf16(a, b)
    with pytest.raises(vqa_amd.VqfError):
        ops.gemm_bf16(torch.zeros(4, 8, device="cuda"), torch.zeros(4, 8, device="cuda"))   # fp32 tensors


@pytest.mark.parametrize("mhb", [False, True])
def test_models_in_bf16_mode_track_the_fp32_oracle(mhb):
    """gemm_dtype='bf16' (config 3): bf16 operands in img_conv1d / co_att_conv1, fp32 elsewhere.
    Tolerance vs the fp32 mode (parity-proven against the oracle elsewhere): 3e-2 relative on the outputs (8-bit mantissa
    operands, K = 2048); every gradient finite; the classifier's gradient -- the one tensor downstream of every signed
    square root -- within 10 %.  The OTHER gradients are not compared with the fp32 step's here (round 3 did, with a skip
    list of four tensors and a private 15 % for co_att_conv1.bias: a bf16 rounding upstream of 0.5*|s|^-1/2 moves them by
    O(10 %), which is the loss surface, not the kernels): they are checked node by node against fp64 evaluations of the same
    nodes on the same bf16-rounded operands, with no exception list, in
    tests/test_gpu_bf16_nodes.py::test_models_in_bf16_modes_every_node_at_full_dims."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import vqa_amd, recipe
    from cases import MFB_CASES, MHBCOATT_CASES
    from golden_util import mfb_inputs
    case = dict((MHBCOATT_CASES if mhb else MFB_CASES)[-1 if mhb else -2])      # full-size dims, N=2
    cfg, img, q, glove, hard, soft = mfb_inputs(case, "cuda")
    model = (vqa_amd.MHBCoAtt if mhb else vqa_amd.MFB)(cfg)
    model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), case["salt"]))
                           for k, v in model.state_dict().items()})
    model = model.cuda().train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    if not mhb:
        model.unit_softmax = False          # live attention so that the bf16 GEMMs reach the output
    res = {}
    for mode in ("fp32", "bf16"):           # fp32 mode is parity-proven against the oracle elsewhere
        model.gemm_dtype = mode
        model.zero_grad(set_to_none=True)
        out = model.forward(img, q)
        loss = torch.nn.KLDivLoss()(out, soft) if mhb else torch.nn.CrossEntropyLoss()(out, hard)
        loss.backward()
        res[mode] = (out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
        # drop this mode's graph before the next forward: a live loss keeps the parameters' AccumulateGrad nodes -- and the stream
        # they were created under (the fp32 mode builds the projection's node under the side stream, the bf16 mode on the caller's)
        # -- alive across the mode switch, which torch reports as a stream mismatch (GPUTEST_r04's warning; a training loop keeps
        # ONE mode, and profiles/r05_c3_timeline.txt shows the weight gradient running beside the LSTM backward in config 3)
        del out, loss
    ref, gref = res["fp32"]
    out, gb = res["bf16"]
    assert _rel(out, ref) <= 3e-2
    assert not torch.equal(out, ref)        # the bf16 kernels really ran
    assert all(torch.isfinite(g).all() for g in gb.values())
    for k in ("linear_pred.weight", "linear_pred.bias"):
        assert float((gb[k] - gref[k]).norm()) <= 0.1 * float(gref[k].norm()) + 1e-9, k


@pytest.mark.parametrize("mhb", [False, True])
def test_models_in_bf16_all_mode(mhb):
    """gemm_dtype='bf16-all': bf16 operands also in ques_proj1, the final blocks' ques_proj* / img_proj* and the
    question-attention conv (forward, dgrad, weight gradient).  Outputs within the stated bf16 tolerance 3e-2 of the
    fp32 mode (which is parity-proven against the oracle); every gradient finite; the classifier's gradient (the one
    tensor downstream of every signed square root) within 10 %; the other deviations are reported."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import vqa_amd, recipe
    from cases import MFB_CASES, MHBCOATT_CASES
    from golden_util import mfb_inputs
    case = dict((MHBCOATT_CASES if mhb else MFB_CASES)[-1 if mhb else -2], N=8)      # full-size dims, N = 8 (bf16 GEMMs need M % 8 == 0)
    cfg, img, q, glove, hard, soft = mfb_inputs(case, "cuda")
    model = (vqa_amd.MHBCoAtt if mhb else vqa_amd.MFB)(cfg)
    model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), case["salt"]))
                           for k, v in model.state_dict().items()})
    model = model.cuda().train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    if not mhb:
        model.unit_softmax = False
    res = {}
    for mode in ("fp32", "bf16", "bf16-all"):
        model.gemm_dtype = mode
        model.zero_grad(set_to_none=True)
        out = model.forward(img, q)
        loss = torch.nn.KLDivLoss()(out, soft) if mhb else torch.nn.CrossEntropyLoss()(out, hard)
        loss.backward()
        res[mode] = (out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
        del out, loss                        # (no graph of the previous mode alive across the switch: see the test above)
    ref, gref = res["fp32"]
    out, gb = res["bf16-all"]
    assert _rel(out, ref) <= 3e-2
    assert not torch.equal(out, res["bf16"][0])              # the additional bf16 GEMMs really ran
    assert all(torch.isfinite(g).all() for g in gb.values())
    dev = {k: float((gb[k] - g).norm() / (g.norm() + 1e-30)) for k, g in gref.items() if float(g.norm()) > 1e-9}
    print("bf16-all vs fp32 gradients (relative deviation): " + " ".join("%s=%.3f" % kv for kv in dev.items()))
    assert dev["linear_pred.weight"] <= 0.1 and dev["linear_pred.bias"] <= 0.1


# ---- bf16 FEATURE STORAGE: the image grid kept in bf16 in HBM ------------------
@pytest.mark.parametrize("G,unit", [(2, False), (2, True), (1, False)])
@pytest.mark.parametrize("N,S,C", [(3, 196, 2048), (2, 20, 96), (2, 7, 52)])
def test_glimpse_kernels_on_bf16_features_equal_fp32_kernels_on_the_same_values(ops, N, S, C, G, unit):
    """Same arithmetic, narrower loads: results must be BIT-identical to the fp32-feature kernels run
    on the bf16-rounded values widened back to fp32."""
    feat_b = _r((N, S, C), 11).to(torch.bfloat16).cuda()
    feat_f = feat_b.float()
    logits = _r((N * S, G), 12, 2.0).cuda()
    w_b, p_b = ops.glimpse_pool_fwd(feat_b, logits, unit)
    w_f, p_f = ops.glimpse_pool_fwd(feat_f, logits, unit)
    assert torch.equal(w_b, w_f) and torch.equal(p_b, p_f)
    dpool = _r((N, G * C), 13).cuda()
    dl_b, none = ops.glimpse_pool_bwd(dpool, feat_b, w_b, unit, False)
    dl_f, _ = ops.glimpse_pool_bwd(dpool, feat_f, w_f, unit, False)
    assert none is None and torch.equal(dl_b, dl_f)
    from vqa_amd import VqfError
    with pytest.raises(VqfError):
        ops.glimpse_pool_bwd(dpool, feat_b, w_b, unit, True)        # no gradient into bf16 data


@pytest.mark.parametrize("mhb", [False, True])
def test_models_take_bf16_feature_tensors(mhb):
    """A bf16 image tensor (what FeatureStager(bf16=True) delivers) gives, in gemm_dtype='bf16', exactly
    the outputs and gradients of the fp32 tensor holding the same bf16-representable values (the per-step
    cast is then exact, so both runs multiply identical operands); fp32 mode refuses bf16 features."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import vqa_amd, recipe
    from vqa_amd import VqfError
    from cases import MFB_CASES, MHBCOATT_CASES
    from golden_util import mfb_inputs
    case = dict((MHBCOATT_CASES if mhb else MFB_CASES)[-1 if mhb else -2])
    cfg, img, q, glove, hard, soft = mfb_inputs(case, "cuda")
    model = (vqa_amd.MHBCoAtt if mhb else vqa_amd.MFB)(cfg)
    model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), case["salt"]))
                           for k, v in model.state_dict().items()})
    model = model.cuda().train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    if not mhb:
        model.unit_softmax = False
    img_b = img.to(torch.bfloat16)
    res = []
    model.gemm_dtype = "bf16"
    for x in (img_b, img_b.float()):
        model.zero_grad(set_to_none=True)
        out = model.forward(x, q)
        loss = torch.nn.KLDivLoss()(out, soft) if mhb else torch.nn.CrossEntropyLoss()(out, hard)
        loss.backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    model.gemm_dtype = "fp32"
    with pytest.raises(VqfError):
        model.forward(img_b, q)


def test_mhb_mean_pool_accepts_bf16_features():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import vqa_amd
    from cases import MFB_CASES, make_cfg
    cfg = make_cfg(dict(MFB_CASES[0], model_name="mhb"))
    N, T = 3, 7
    torch.manual_seed(0)
    m = vqa_amd.MHB(cfg).cuda().eval()
    img = torch.rand(N, cfg.img_feature_dim, cfg.img_feature_channel, device="cuda").to(torch.bfloat16)
    q = torch.randint(1, cfg.q_vocab_size, (N, T), device="cuda")
    ql = torch.tensor([T, 3, 5])
    with torch.no_grad():
        a = m.forward(img, q, ql)
        b = m.forward(img.float(), q, ql)
    assert torch.equal(a, b)


def test_bf16_projection_storage_is_exact_rounding_and_fuse_kernels_agree(ops):
    """VQF_GEMM_OUT_BF16: the large-tile kernel's bf16 output is the rounding of its fp32 result (within one bf16 ulp of
    the fp32-output launch, which may use split-K and therefore another summation order; > 99 % bit-identical to its
    RNE); the fusion kernels fed the bf16 P give bit-identical results to the fp32-P kernels fed the same values."""
    N, L, O = 8, 196, 1000
    M, K = N * L, 2048
    X = _r((M, K), 31).to(torch.bfloat16).cuda()
    W = (_r((5 * O, K), 32) * 0.05).to(torch.bfloat16).cuda()
    bias = _r((5 * O,), 33).cuda()
    Pf = ops.gemm_bf16(X, W, bias=bias)
    Pb = ops.gemm_bf16(X, W, bias=bias, out_bf16=True)
    assert Pb is not None and Pb.dtype == torch.bfloat16          # 7 x 20 tiles of 256x256 >= 64: the large-tile kernel
    assert bool(((Pb.float() - Pf).abs() <= Pf.abs() * 2.0 ** -8 + 1e-6).all())
    assert float((Pb.view(torch.int16) == Pf.to(torch.bfloat16).view(torch.int16)).float().mean()) > 0.99
    assert ops.gemm_bf16(X[:300], W, bias=bias, out_bf16=True) is None       # too few tiles: caller falls back
    q = _r((N, 5 * O), 34).cuda()
    Pr = Pb.float()
    Yb, nb, ib, _ = ops.mfb_fuse_fwd(Pb, q, N, L, O, seed=77, p_drop=0.1)
    Yf, nf, i_f, _ = ops.mfb_fuse_fwd(Pr, q, N, L, O, seed=77, p_drop=0.1)
    assert torch.equal(Yb, Yf) and torch.equal(nb, nf)
    # the un-normalised form that feeds a NormLink consumer can leave a bf16 copy of R beside the fp32 one (round 5): the bits of
    # vqf_cast_f32_bf16(R, pad 32), zero pad columns, and the fp32 outputs untouched
    rb = []
    R1, n1, i1, _ = ops.mfb_fuse_fwd(Pb, q, N, L, O, seed=77, p_drop=0.1, normalise=False, r_bf16=rb)
    R0, n0, i0, _ = ops.mfb_fuse_fwd(Pb, q, N, L, O, seed=77, p_drop=0.1, normalise=False)
    assert torch.equal(R1, R0) and torch.equal(n1, n0) and torch.equal(i1, i0) and len(rb) == 1
    assert rb[0].shape == (N * L, (O + 31) // 32 * 32) and torch.equal(rb[0].view(torch.int16), ops.cast_bf16(R0, 32).view(torch.int16))
    dY = _r((N * L, O), 35).cuda()
    dPb, dqb, _, dbb = ops.mfb_fuse_bwd(dY, Yb, nb, ib, Pb, q, N, L, O, seed=77, p_drop=0.1, want_dbias=True, dp_bf16=True)
    dPf, dqf, _, dbf = ops.mfb_fuse_bwd(dY, Yf, nf, i_f, Pr, q, N, L, O, seed=77, p_drop=0.1, want_dbias=True, dp_bf16=True)
    assert torch.equal(dPb.view(torch.int16), dPf.view(torch.int16)) and torch.equal(dqb, dqf) and torch.equal(dbb, dbf)
