"""GPU parity for rows a12-a14 of SURVEY.md section 8: HieCoAtten, modules.py, networks.py on the HIP
path vs the reference's golden vectors and vs the oracle (same seeded inputs, explicit dropout masks),
plus the element-wise / single-glimpse kernels they add.  Forward 1e-4; gradients by grad_parity."""
import numpy as np
import pytest
import torch

import recipe
from cases import HIE_CASES, ATTNET_CASES, IBOW_CASES, ATT_MODULE_CASES
from golden_util import load_golden, recipe_sd, rel_err, check_grads, grad_parity
from oracle import ref_torch as O

pytestmark = pytest.mark.gpu
OUT_TOL = 1e-4


def _vqa():
    import vqa_amd
    vqa_amd.lib.load()
    return vqa_amd


def _load(model, salt):
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = v
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            sd[k] = torch.ones_like(v)
        else:
            sd[k] = torch.from_numpy(recipe.weight_for(k, tuple(v.shape), salt))
    model.load_state_dict(sd)
    return model.cuda()


def _grads(model):
    return {k: p.grad for k, p in model.named_parameters()}


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _r(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * scale).float().double()


# ---------------------------------------------------------------- kernels
def test_dropout_and_tanh_dropout_kernels():
    ops = _vqa().ops
    x, b = _r((300, 64), 1), _r((300, 64), 2)
    keep = (torch.rand((300, 64), generator=torch.Generator().manual_seed(3)) >= 0.5).to(torch.uint8)
    sc = keep.double() * 2.0
    y = ops.dropout(x.float().cuda(), keep=keep.cuda(), p_drop=0.5)
    assert _rel(y, x * sc) <= 1e-6
    t = ops.tanh_dropout_fwd(x.float().cuda(), b.float().cuda(), keep=keep.cuda(), p_drop=0.5)
    ref = torch.tanh(x + b) * sc
    assert _rel(t, ref) <= 1e-6
    dy = _r((300, 64), 4)
    dx = ops.tanh_dropout_bwd(dy.float().cuda(), t, keep=keep.cuda(), p_drop=0.5)
    assert _rel(dx, dy * sc * (1 - torch.tanh(x + b) ** 2)) <= 1e-5
    # Philox: rate, determinism, same mask in the "backward" call
    ones = torch.ones((1000, 512), device="cuda")
    z = ops.dropout(ones, seed=77, p_drop=0.5)
    assert abs(float((z == 0).float().mean()) - 0.5) < 5e-3
    assert torch.equal(z, ops.dropout(ones, seed=77, p_drop=0.5))
    assert not torch.equal(z, ops.dropout(ones, seed=78, p_drop=0.5))
    t1 = ops.tanh_dropout_fwd(ones, None, seed=77, p_drop=0.5)
    assert torch.equal(t1 == 0, z == 0)
    # p = 0 is the identity / plain tanh
    assert torch.equal(ops.dropout(ones, p_drop=0.0), ones)


@pytest.mark.parametrize("N,L,E,T", [(3, 196, 512, 14), (2, 50, 64, 5), (5, 17, 96, 16), (300, 196, 512, 14), (1, 1000, 1024, 1)])
def test_hie_affinity_kernel_vs_fp64_and_the_elementwise_masks(N, L, E, T):
    """vqf_hie_affinity (hieCoAtten.py:32-33 and its gradient): sums vs fp64 within the fp32 bound of a K-term product (one or
    two operand pairs, strided rows), the fused dropout(tanh(.)) and its backward against the fp64 formula with an explicit
    mask, and -- Philox -- the SAME zero pattern as vqf_tanh_dropout_fwd on the contiguous (N*T, L) tensor."""
    ops = _vqa().ops
    from node_harness import gemm_tol
    assert ops.hie_affinity_supported(N, L, E, T, 2)
    wide = _r((N * T, 2 * E), 11).float().cuda()         # column blocks of wider buffers, like [Cq | que_]
    x1, x2 = wide[:, :E], wide[:, E:]
    widey = _r((N * L, 2 * E), 12).float().cuda()
    y1, y2 = widey[:, :E], widey[:, E:]
    d = lambda t, R: t.double().view(N, R, E)
    s1 = torch.einsum("nte,nle->ntl", d(x1, T), d(y1, L))
    s2 = s1 + torch.einsum("nte,nle->ntl", d(x2, T), d(y2, L))
    c1 = ops.hie_affinity(x1, y1, N, L, T)
    c2 = ops.hie_affinity(x1, y1, N, L, T, x2=x2, y2=y2)
    assert _rel(c1, s1) <= gemm_tol(E) and _rel(c2, s2) <= gemm_tol(2 * E)
    assert _rel(c1, ops.bgemm(x1.reshape(N, T, E).contiguous(), y1.reshape(N, L, E).contiguous())) <= 2 * gemm_tol(E)
    keep = (torch.rand((N, T, L), generator=torch.Generator().manual_seed(13)) >= 0.5).to(torch.uint8).cuda()
    sc = keep.double() * 2.0
    f = ops.hie_affinity(x1, y1, N, L, T, epi=1, drop=(keep, 0, 0.5))
    smax = float(s1.abs().max())                          # gemm_tol is relative to max|s|; tanh' <= 1 turns it into an absolute bound
    assert _rel(f, torch.tanh(s1) * sc) <= 2e-6 + gemm_tol(E) * smax
    b = ops.hie_affinity(x1, y1, N, L, T, x2=x2, y2=y2, epi=2, yprev=f, drop=(keep, 0, 0.5))
    assert _rel(b, s2 * sc * (1 - torch.tanh(s1) ** 2)) <= 1e-5 + 2 * gemm_tol(E) * smax
    if N * T * L % 4 == 0:                                # the flat element-wise kernels take whole groups of four
        fp = ops.hie_affinity(x1, y1, N, L, T, epi=1, drop=(None, 77, 0.5))
        flat = ops.tanh_dropout_fwd(c1.view(N * T, L), None, seed=77, p_drop=0.5)
        assert torch.equal(fp.view(N * T, L) == 0, flat == 0) and _rel(fp.view(N * T, L), flat) <= 2e-6
        bp = ops.hie_affinity(x1, y1, N, L, T, x2=x2, y2=y2, epi=2, yprev=fp, drop=(None, 77, 0.5))
        assert _rel(bp, ops.tanh_dropout_bwd(c2.view(N * T, L).clone(), fp.view(N * T, L), seed=77, p_drop=0.5).view(N, T, L)) <= 1e-5
    with pytest.raises(_vqa().lib.VqfError):
        ops.hie_affinity(wide[:, :40], widey[:, :40], N, L, T)         # E % 32 != 0


def test_softmax_rows_kernel():
    ops = _vqa().ops
    for R, W in [(21, 196), (5, 7), (300, 22), (2, 1000)]:
        x = _r((R, W), 5, 4.0).requires_grad_()
        ref = torch.softmax(x, 1)
        y = ops.softmax_rows_fwd(x.detach().float().cuda())
        assert _rel(y, ref) <= 1e-5
        dy = _r((R, W), 6)
        ref.backward(dy)
        assert _rel(ops.softmax_rows_bwd(dy.float().cuda(), y), x.grad) <= 2e-5


@pytest.mark.parametrize("N,S,C", [(3, 196, 512), (2, 22, 64), (1, 1, 8)])
def test_single_glimpse_attention_with_weight_gradient(N, S, C):
    """G = 1 head without ReLU, gradient arriving through pooled AND through the returned weights."""
    ops = _vqa().ops
    x = _r((N * S, C), 7).requires_grad_()
    w, b = _r((1, C), 8, 0.3).requires_grad_(), _r((1,), 9).requires_grad_()
    logits = x @ w.t() + b
    wt = torch.softmax(logits.view(N, S), 1)
    pooled = torch.einsum("ns,nsc->nc", wt, x.view(N, S, C))
    lg = ops.att_logits_fwd(x.detach().float().cuda(), w.detach().float().cuda(), b.detach().float().cuda())
    assert _rel(lg, logits) <= 1e-5
    wts, pl = ops.glimpse_pool_fwd(x.detach().float().cuda().view(N, S, C), lg, False)
    assert _rel(wts.view(N, S), wt) <= 1e-5 and _rel(pl, pooled) <= 1e-5
    dp, dw_ = _r((N, C), 10), _r((N, 1, S), 11)
    (pooled * dp).sum().backward(retain_graph=True)
    gx_p = x.grad.clone()
    x.grad = None
    w.grad = None
    b.grad = None
    ((pooled * dp).sum() + (wt.view(N, 1, S) * dw_).sum()).backward()
    dlog, dfeat = ops.glimpse_pool_bwd(dp.float().cuda(), x.detach().float().cuda().view(N, S, C), wts, False,
                                       True, dwts=dw_.float().cuda())
    dx, dw2, db2, _ = ops.att_logits_bwd(dlog, x.detach().float().cuda(), w.detach().float().cuda(),
                                         relu_mask=False)
    assert _rel(dw2, w.grad) <= 3e-5
    assert _rel(db2, b.grad) <= 3e-5 or float(b.grad.abs().max()) < 1e-9
    assert _rel(dx + dfeat.view(N * S, C), x.grad) <= 3e-5
    assert gx_p is not None


# ---------------------------------------------------------------- HieCoAtten
def _hie_inputs(case, dev="cuda"):
    N = case["N"]
    img = torch.from_numpy(recipe.img_features(N, case["L"], case["img_size"], case["salt"])).to(dev)
    q = torch.from_numpy(recipe.question_tokens(N, case["T"], case["V"], case["salt"])).to(dev)
    ans = torch.from_numpy(recipe.hard_answers(N, case["A"], case["salt"])).to(dev)
    return img, q, ans


def _hie_oracle_pair(case, img, q, ans, drop=None):
    res = []
    for dt in (torch.float32, torch.float64):
        sd = {k: v.to(dt).requires_grad_(True)
              for k, v in recipe_sd(O.hiecoatten_shapes(case["img_size"], case["V"], case["E"], case["A"]),
                                    case["salt"]).items()}
        x, av, aq = O.hiecoatten_forward(sd, img.cpu().to(dt), q.cpu(), drop=drop)
        O.ce_loss(x, ans.cpu()).backward()
        res.append(((x.detach(), av.detach(), aq.detach()), {k: v.grad for k, v in sd.items()}))
    return res[0][0], res[0][1], res[1][1]


@pytest.fixture(params=["stream", "stream-bgemm-affinity", "bgemm", "staged"])
def hie_form(request):
    """The executions of the ladder: one node with the tiny-T stages as streaming passes and the affinity products on
    vqf_hie_affinity (default; csrc/hie.hip), the same with the affinity products as batched GEMMs (round 4's form), one node
    with batched GEMMs + element-wise kernels throughout (what unsupported shapes get: T > 16 ...), one node per stage
    (rounds 1-3)."""
    vqa = _vqa()
    old = vqa.functions.HieCoreFn.STREAM, vqa.functions.HieCoreFn.AFFINITY
    vqa.functions.HieCoreFn.STREAM = request.param.startswith("stream")
    vqa.functions.HieCoreFn.AFFINITY = request.param == "stream"
    yield request.param
    vqa.functions.HieCoreFn.STREAM, vqa.functions.HieCoreFn.AFFINITY = old


@pytest.mark.parametrize("case", [pytest.param(c, id=c["name"]) for c in HIE_CASES])
def test_hiecoatten_matches_reference_golden(case, hie_form):
    vqa = _vqa()
    gold = load_golden("hie_" + case["name"])
    model = _load(vqa.HieCoAtten(block_num=case["L"], word_num=case["T"], img_size=case["img_size"],
                                 vocab_size=case["V"], embed_size=case["E"], output_size=case["A"]), case["salt"])
    model.fused = hie_form != "staged"
    model.drop_p = 0.0            # goldens were captured with the functional dropout patched to identity
    img, q, ans = _hie_inputs(case)
    N = case["N"]
    x, av, aq = model.forward(img, q)
    assert rel_err(x.detach().cpu().numpy(), gold["x"]) <= OUT_TOL
    assert rel_err(av.detach().cpu().numpy().reshape(N, -1), gold["av"].reshape(N, -1)) <= OUT_TOL
    assert rel_err(aq.detach().cpu().numpy().reshape(N, -1), gold["aq"].reshape(N, -1)) <= OUT_TOL
    loss = torch.nn.CrossEntropyLoss()(x, ans)
    assert abs(loss.item() - float(gold["loss"])) <= OUT_TOL * max(1.0, float(gold["loss"]))
    loss.backward()
    check_grads(_grads(model), gold, 1e-2)
    _, g32, g64 = _hie_oracle_pair(case, img, q, ans)
    grad_parity(_grads(model), g32, g64)
    assert model.fc_Wbq.weight.grad is None            # hieCoAtten.py:31: never used


@pytest.mark.parametrize("ci", [2, 0])
def test_hiecoatten_always_on_dropout_with_explicit_masks(ci, hie_form):
    vqa = _vqa()
    case = HIE_CASES[ci]                  # [2]: T = 22 (batched-GEMM form even when "stream" is asked for), [0]: T = 7
    N, L, T, E = case["N"], case["L"], case["T"], case["E"]
    model = _load(vqa.HieCoAtten(block_num=L, word_num=T, img_size=case["img_size"], vocab_size=case["V"],
                                 embed_size=E, output_size=case["A"]), case["salt"]).eval()   # eval: still drops
    model.fused = hie_form != "staged"
    img, q, ans = _hie_inputs(case)
    shapes = dict(img=(N * L, E), que=(N * T, E), C=(N * T, L), Hv=(N * L, E), Hq=(N * T, E))
    masks = {k: torch.from_numpy(recipe.keep_mask(s, 0.5, "hie_" + k)) for k, s in shapes.items()}
    model.set_keep_masks(**{k: m.cuda() for k, m in masks.items()})
    x, av, aq = model.forward(img, q)
    torch.nn.CrossEntropyLoss()(x, ans).backward()
    drop = dict(img=masks["img"].view(N, L, E), que=masks["que"].view(N, T, E), C=masks["C"].view(N, T, L),
                Hv=masks["Hv"].view(N, L, E), Hq=masks["Hq"].view(N, T, E))
    (ox, oav, oaq), g32, g64 = _hie_oracle_pair(case, img, q, ans, drop=drop)
    assert rel_err(x.detach().cpu().numpy(), ox.numpy()) <= OUT_TOL
    assert rel_err(av.detach().cpu().numpy(), oav.numpy()) <= OUT_TOL
    assert rel_err(aq.detach().cpu().numpy(), oaq.numpy()) <= OUT_TOL
    grad_parity(_grads(model), g32, g64)
    # without masks two eval() calls differ (Philox, always on) -- the reference's behaviour
    model.set_keep_masks()
    a = model.forward(img, q)[0]
    b = model.forward(img, q)[0]
    assert not torch.equal(a, b)


def test_hiecoatten_attention_outputs_are_differentiable(hie_form):
    vqa = _vqa()
    case = HIE_CASES[1]
    model = _load(vqa.HieCoAtten(block_num=case["L"], word_num=case["T"], img_size=case["img_size"],
                                 vocab_size=case["V"], embed_size=case["E"], output_size=case["A"]), case["salt"])
    model.fused = hie_form != "staged"
    model.drop_p = 0.0
    img, q, ans = _hie_inputs(case)
    x, av, aq = model.forward(img, q)
    wv = torch.linspace(-1, 1, av.numel(), device="cuda").view_as(av)
    wq = torch.linspace(1, -1, aq.numel(), device="cuda").view_as(aq)
    ((av * wv).sum() + (aq * wq).sum() + x.sum() * 1e-3).backward()
    res = []
    for dt in (torch.float32, torch.float64):
        sd = {k: v.to(dt).requires_grad_(True)
              for k, v in recipe_sd(O.hiecoatten_shapes(case["img_size"], case["V"], case["E"], case["A"]),
                                    case["salt"]).items()}
        ox, oav, oaq = O.hiecoatten_forward(sd, img.cpu().to(dt), q.cpu())
        ((oav * wv.cpu().to(dt)).sum() + (oaq * wq.cpu().to(dt)).sum() + ox.sum() * 1e-3).backward()
        res.append({k: v.grad for k, v in sd.items()})
    grad_parity(_grads(model), res[0], res[1])


# ---------------------------------------------------------------- networks.py / modules.py
@pytest.mark.parametrize("case", [pytest.param(c, id=c["name"]) for c in ATTNET_CASES])
def test_attentionnet_matches_reference_golden(case):
    vqa = _vqa()
    gold = load_golden("attnet_" + case["name"])
    model = _load(vqa.AttentionNet(block_num=case["L"], word_num=case["T"], img_size=case["img_size"],
                                   vocab_size=case["V"], embed_size=case["E"], att_num=case["att_num"],
                                   output_size=case["A"]), case["salt"]).train()
    model.drop_p = 0.0
    N = case["N"]
    img = torch.from_numpy(recipe.img_features(N, case["L"], case["img_size"], case["salt"])).cuda()
    q = torch.from_numpy(recipe.question_tokens(N, case["T"], case["V"], case["salt"], pad_tail=False)).cuda()
    ans = torch.from_numpy(recipe.hard_answers(N, case["A"], case["salt"])).cuda()
    x, qa, ia = model.forward(img, q)
    assert rel_err(qa.detach().cpu().numpy(), gold["que_att"]) <= OUT_TOL
    assert rel_err(ia.detach().cpu().numpy(), gold["img_att"]) <= OUT_TOL
    assert rel_err(x.detach().cpu().numpy(), gold["x"]) <= 5e-4        # BatchNorm over N<=4 rows amplifies
    torch.nn.CrossEntropyLoss()(x, ans).backward()
    res = []
    for dt in (torch.float32, torch.float64):
        sd = {k: v.to(dt).requires_grad_(True)
              for k, v in recipe_sd(O.attentionnet_shapes(case["L"], case["T"], case["img_size"], case["V"],
                                                          case["E"], case["att_num"], case["A"]),
                                    case["salt"]).items()}
        ox, _, _ = O.attentionnet_forward(sd, img.cpu().to(dt), q.cpu(), att_num=case["att_num"])
        O.ce_loss(ox, ans.cpu()).backward()
        res.append({k: v.grad for k, v in sd.items()})
    grad_parity(_grads(model), res[0], res[1], k=6.0, floor=1e-3)
    if case["name"].startswith("full"):
        # the shapes the reference trains with (L = 196, T = 14, embed 512, 6 layers): also straight against the reference's
        # own fp32 gradient digests (norm + 16 sampled entries per tensor)
        check_grads(_grads(model), gold, 2e-2)


@pytest.mark.parametrize("case", [pytest.param(c, id=c["name"]) for c in IBOW_CASES])
def test_ibowimg_matches_reference_golden(case):
    vqa = _vqa()
    gold = load_golden("ibow_" + case["name"])
    model = _load(vqa.iBOWIMG(case["img_size"], case["V"], case["E"], case["A"]), case["salt"]).train()
    model.drop_p = 0.0
    N = case["N"]
    img = torch.from_numpy(recipe.sym_tensor((N, case["img_size"]), 1.0, recipe.name_seed("ibow_img", case["salt"]))).cuda()
    q = torch.from_numpy(recipe.question_tokens(N, case["T"], case["V"], case["salt"])).cuda()
    x = model.forward(img, q)
    assert rel_err(x.detach().cpu().numpy(), gold["x"]) <= 5e-4
    torch.nn.CrossEntropyLoss()(x, torch.from_numpy(recipe.hard_answers(N, case["A"], case["salt"])).cuda()).backward()
    check_grads(_grads(model), gold, 1e-2)


@pytest.mark.parametrize("case", [pytest.param(c, id=c["name"]) for c in ATT_MODULE_CASES])
def test_attention_modules_match_reference_golden(case):
    vqa = _vqa()
    gold = load_golden("mod_" + case["name"])
    kind, Dm = case["kind"], case["D"]
    N, L, T = case["N"], case["L"], case["T"]
    f1 = torch.from_numpy(recipe.sym_tensor((N, L, Dm), 1.0, recipe.name_seed("f1", case["salt"]))).cuda().requires_grad_()
    f2 = torch.from_numpy(recipe.sym_tensor((N, T, Dm), 1.0, recipe.name_seed("f2", case["salt"]))).cuda().requires_grad_()
    if kind == "attention_1":
        m = _load(vqa.Attention_1(Dm), case["salt"])
        fh, att = m.forward(f1, f2)
        assert rel_err(fh.detach().cpu().numpy(), gold["f_hat"]) <= OUT_TOL
        assert rel_err(att.detach().cpu().numpy(), gold["att"]) <= OUT_TOL
        ((fh * fh).sum() + (att * att).sum()).backward()
    elif kind == "attention_2":
        m = _load(vqa.Attention_2(Dm), case["salt"])
        fh, att = m.forward(f1, f2)
        assert rel_err(fh.detach().cpu().numpy(), gold["f_hat"]) <= OUT_TOL
        assert rel_err(att.detach().cpu().numpy(), gold["att"]) <= OUT_TOL
        ((fh * fh).sum() + (att * att).sum()).backward()
        assert rel_err(f2.grad.cpu().numpy(), gold["df2"]) <= 5e-4
    elif kind.startswith("attention_layer"):
        m = _load(vqa.Attention_layer(Dm, 1 if kind.endswith("1") else 2), case["salt"])
        a, b, att = m.forward(f1, f2)
        assert rel_err(a.detach().cpu().numpy(), gold["a"]) <= OUT_TOL
        assert rel_err(b.detach().cpu().numpy(), gold["b"]) <= OUT_TOL
        assert rel_err(att.detach().cpu().numpy(), gold["att"]) <= OUT_TOL
        ((b * b).sum() + (att * att).sum()).backward()
        assert rel_err(f2.grad.cpu().numpy(), gold["df2"]) <= 5e-4
    else:
        m = _load(vqa.Nonlinear_layer(Dm), case["salt"])
        o = m.forward(f1)
        assert rel_err(o.detach().cpu().numpy(), gold["o"]) <= OUT_TOL
        (o * o).sum().backward()
    assert rel_err(f1.grad.cpu().numpy(), gold["df1"]) <= 5e-4
    check_grads(_grads(m), gold, 5e-3)


def test_config4_hiecoatten_full_batch_256_row_pairing():
    """BASELINE config 4 shapes (HieCoAtten, B=256, img 2048 -> 512): attention weights normalise and,
    because x = cat((v,q),0).view(N,-1) pairs rows (hieCoAtten.py:52-53), output row i < N/2 only depends
    on samples 2i and 2i+1: it must equal row 0 of the oracle run on those two samples."""
    vqa = _vqa()
    N, L, T = 256, 196, 14
    case = dict(salt=25, img_size=2048, V=1000, E=512, A=1000)
    model = _load(vqa.HieCoAtten(block_num=L, word_num=T, img_size=2048, vocab_size=1000, embed_size=512,
                                 output_size=1000), case["salt"])
    model.drop_p = 0.0
    g = torch.Generator().manual_seed(4321)
    img = torch.relu(torch.randn((N, L, 2048), generator=g))
    q = torch.randint(1, 1000, (N, T), generator=torch.Generator().manual_seed(4322))
    ans = torch.randint(0, 1000, (N,), generator=torch.Generator().manual_seed(4323))
    x, av, aq = model.forward(img.cuda(), q.cuda())
    torch.nn.CrossEntropyLoss()(x, ans.cuda()).backward()
    assert torch.allclose(av.detach().sum(1).cpu(), torch.ones(N), atol=1e-4)
    assert torch.allclose(aq.detach().sum(1).cpu(), torch.ones(N), atol=1e-4)
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())
    sd = recipe_sd(O.hiecoatten_shapes(2048, 1000, 512, 1000), case["salt"])
    for i in (0, 77, 127):
        ox, oav, _ = O.hiecoatten_forward(sd, img[2 * i:2 * i + 2], q[2 * i:2 * i + 2])
        assert rel_err(x[i].detach().cpu().numpy(), ox[0].numpy()) <= OUT_TOL
        assert rel_err(av[2 * i:2 * i + 2].detach().cpu().numpy(), oav.numpy()) <= OUT_TOL


def test_hiecoatten_philox_masks_are_the_same_in_every_form():
    """The in-kernel Philox masks are indexed by the element's position in the logical tensor, so the three executions of the
    ladder draw the SAME masks from the same seeds: outputs and gradients agree to rounding (re-association only)."""
    vqa = _vqa()
    case = HIE_CASES[0]
    img, q, ans = _hie_inputs(case)
    res = {}
    for form in ("stream", "bgemm", "staged"):
        model = _load(vqa.HieCoAtten(block_num=case["L"], word_num=case["T"], img_size=case["img_size"], vocab_size=case["V"],
                                     embed_size=case["E"], output_size=case["A"]), case["salt"])
        model.fused = form != "staged"
        vqa.functions.HieCoreFn.STREAM = form == "stream"
        try:
            torch.manual_seed(77)                      # the seeds of the five masks come from torch's CPU generator
            x, av, aq = model.forward(img, q)
            torch.nn.CrossEntropyLoss()(x, ans).backward()
        finally:
            vqa.functions.HieCoreFn.STREAM = True
        res[form] = (x.detach().clone(), {k: v.clone() for k, v in _grads(model).items() if v is not None})
    assert float(res["staged"][0].abs().max()) > 0
    gmax = max(float(g.norm()) for g in res["staged"][1].values())
    for form in ("stream", "bgemm"):
        assert rel_err(res[form][0].cpu().numpy(), res["staged"][0].cpu().numpy()) <= 1e-5, form
        for k, g in res["staged"][1].items():
            d = float((res[form][1][k] - g).norm())      # (the softmax biases carry a mathematically-zero gradient: gmax floor)
            assert d <= 1e-4 * float(g.norm()) + 1e-6 * gmax, (form, k, d, float(g.norm()))


def test_config4_hiecoatten_full_batch_256_gradients_vs_oracle(monkeypatch):
    """BASELINE config 4 at its full batch, every output and every gradient: HieCoAtten, B=256, 196 regions x 2048,
    fp32, dropout off, against the oracle run on the same batch in fp32 and fp64 (forward 1e-4, gradients by
    grad_parity).  These are the shapes of the bench line for this model: the 50176-row img_emb GEMM, its
    K = 50176 weight gradient and the 50176 x 512 x 512 ladder products."""
    vqa = _vqa()
    N, L, T = 256, 196, 14
    case = dict(salt=26, img_size=2048, V=1000, E=512, A=1000)
    model = _load(vqa.HieCoAtten(block_num=L, word_num=T, img_size=2048, vocab_size=1000, embed_size=512,
                                 output_size=1000), case["salt"])
    model.drop_p = 0.0
    img = torch.relu(torch.randn((N, L, 2048), generator=torch.Generator().manual_seed(4331)))
    q = torch.randint(1, 1000, (N, T), generator=torch.Generator().manual_seed(4332))
    ans = torch.randint(0, 1000, (N,), generator=torch.Generator().manual_seed(4333))
    from node_harness import Recorder, check_every_node, ALL_NODES
    recd = Recorder(monkeypatch, vqa.functions, ALL_NODES)
    x, av, aq = model.forward(img.cuda(), q.cuda())
    torch.nn.CrossEntropyLoss()(x, ans.cuda()).backward()
    torch.cuda.synchronize()
    assert [r["cls"].__name__ for r in recd.records] == ["HieCoreFn", "LinearFn"]
    # the ladder (one autograd node) and the classifier against their own fp64 evaluations on the same operands
    _, covered = check_every_node(model, recd, "config 4 (HieCoAtten, fp32) node checks at B=256", min_links=1, skip_params=())
    del recd
    (ox, oav, oaq), g32, g64 = _hie_oracle_pair(case, img, q, ans)
    assert rel_err(x.detach().cpu().numpy(), ox.numpy()) <= OUT_TOL
    assert rel_err(av.detach().cpu().numpy(), oav.numpy()) <= OUT_TOL
    assert rel_err(aq.detach().cpu().numpy(), oaq.numpy()) <= OUT_TOL
    grads = _grads(model)
    assert float(grads["img_emb.weight"].abs().max()) > 0.0 and model.fc_Wbq.weight.grad is None
    assert covered == {k for k, g in grads.items() if g is not None}
    grad_parity(grads, g32, g64, node_checked=covered)
