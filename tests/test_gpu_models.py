"""GPU parity of the drop-in modules (HIP path through the C ABI) against
  (1) the committed golden vectors captured from the reference, and
  (2) the CPU oracle on the same seeded inputs (incl. train mode with explicit
      dropout masks and the full-size batch through per-sample independence).

Tolerances: forward 1e-4 relative (north_star).  Gradients: golden_util.grad_parity --
the HIP gradient must be as close to the exact (fp64) gradient as the CPU fp32
path is (<= 8x its fp32-vs-fp64 distance, floor 5e-4), because the signed-sqrt
derivative is singular at 0 and ANY two fp32 summation orders differ by up to
~1e-2 there; plus a loose 2e-2 check straight against the reference's fp32 digests.
"""
import numpy as np
import pytest
import torch

import recipe
from cases import MFB_CASES, MHBCOATT_CASES, MHB_CASES, make_cfg
from golden_util import load_golden, recipe_sd, mfb_inputs, rel_err, check_grads, grad_parity
from oracle import ref_torch as O
from node_harness import Recorder, check_every_node, ALL_NODES

pytestmark = pytest.mark.gpu

OUT_TOL = 1e-4
GRAD_TOL = 2e-2      # direct digest check vs the reference's fp32 gradients (sanity only)


def _oracle_pair(case, mhb, img, q, glove, target, drop=None, cfg=None):
    """oracle gradients in fp32 and fp64 on the same inputs -> (out32, g32, g64)."""
    cfg = cfg or make_cfg(case)
    res = []
    for dt in (torch.float32, torch.float64):
        sd = {k: v.to(dt).requires_grad_(True)
              for k, v in recipe_sd(O.mfb_shapes(cfg, mhb=mhb), case["salt"]).items()}
        gl = None if glove is None else glove.cpu().to(dt)
        if mhb:
            out = O.mhbcoatt_forward(sd, cfg, img.cpu().to(dt), q.cpu(), glove=gl, drop=drop)
            loss = O.kldiv_loss(out, target.cpu().to(dt))
        else:
            out = O.mfb_forward(sd, cfg, img.cpu().to(dt), q.cpu(), drop=drop)
            loss = O.ce_loss(out, target.cpu())
        loss.backward()
        res.append((out.detach(), {k: v.grad for k, v in sd.items()}))
    return res[0][0], res[0][1], res[1][1]


def _vqa():
    import vqa_amd
    vqa_amd.lib.load()
    return vqa_amd


def _load(model, salt):
    sd = {k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), salt)) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    return model.cuda()


def _no_dropout_train(model):
    """MIOpen's (like cuDNN's) LSTM backward only runs in training mode, so the eval-mode goldens
    are reproduced in train() with every dropout rate set to 0."""
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def _named_grads(model):
    return {k: p.grad for k, p in model.named_parameters()}


@pytest.mark.parametrize("case", [pytest.param(c, id=c["name"]) for c in MFB_CASES])
def test_mfb_matches_reference_golden(case):
    vqa = _vqa()
    gold = load_golden("mfb_" + case["name"])
    cfg, img, q, _, hard, _ = mfb_inputs(case, "cuda")
    model = _no_dropout_train(_load(vqa.MFB(cfg), case["salt"]))
    logits = model.forward(img, q)
    assert rel_err(logits.detach().cpu().numpy(), gold["out"]) <= OUT_TOL
    loss = torch.nn.CrossEntropyLoss()(logits, hard)
    assert abs(loss.item() - float(gold["loss"])) <= OUT_TOL * max(1.0, float(gold["loss"]))
    loss.backward()
    check_grads(_named_grads(model), gold, GRAD_TOL)
    _, g32, g64 = _oracle_pair(case, False, img, q, None, hard)
    grad_parity(_named_grads(model), g32, g64)


@pytest.mark.parametrize("case", [pytest.param(c, id=c["name"]) for c in MHBCOATT_CASES])
def test_mhbcoatt_matches_reference_golden(case):
    vqa = _vqa()
    gold = load_golden("mhbcoatt_" + case["name"])
    cfg, img, q, glove, _, soft = mfb_inputs(case, "cuda")
    model = _no_dropout_train(_load(vqa.MHBCoAtt(cfg), case["salt"]))
    out = model.forward(img, q, glove_matrix=glove)
    assert rel_err(out.detach().cpu().numpy(), gold["out"]) <= OUT_TOL
    loss = torch.nn.KLDivLoss()(out, soft)
    assert abs(loss.item() - float(gold["loss"])) <= 2e-4 * max(1e-3, abs(float(gold["loss"])))
    loss.backward()
    check_grads(_named_grads(model), gold, GRAD_TOL)
    _, g32, g64 = _oracle_pair(case, True, img, q, glove, soft)
    grad_parity(_named_grads(model), g32, g64)


def _oracle_grads(fn, sd, loss_fn):
    out = fn(sd)
    loss = loss_fn(out)
    loss.backward()
    return out.detach(), {k: v.grad for k, v in sd.items()}


def _cmp_grads(model, ogr, tol):
    gmax = max(float(g.norm()) for g in ogr.values() if g is not None)
    for k, p in model.named_parameters():
        g_ref = ogr[k]
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        n_ref = float(g_ref.norm())
        d = float((p.grad.detach().cpu() - g_ref).norm())
        assert d <= tol * max(n_ref, 1e-6 * gmax), (k, d, n_ref)


@pytest.mark.parametrize("mhb", [False, True])
def test_train_mode_with_explicit_dropout_masks(mhb):
    """nn.Dropout active: same keep-masks fed to the HIP kernels and to the oracle."""
    vqa = _vqa()
    case = dict((MHBCOATT_CASES if mhb else MFB_CASES)[2])
    cfg, img, q, glove, hard, soft = mfb_inputs(case, "cuda")
    N, T, L = case["N"], case["T"], cfg.img_feature_dim
    model = _load((vqa.MHBCoAtt if mhb else vqa.MFB)(cfg), case["salt"]).train()
    # the LSTM-output dropout (mfb.py:70 / mhb_coAtt.py:75, p = 0.3) runs on the HIP path too (vqf_dropout_bt): its mask 'l' is
    # indexed (sample, token, unit) there; the oracle applies it where the reference does -- for MHBCoAtt on the (T, N, H) tensor
    H = cfg.hidden_dim
    ml = torch.from_numpy(recipe.keep_mask((N, T, H), 0.3, "l"))
    m1 = torch.from_numpy(recipe.keep_mask((N * L, 5000), 0.1, "m1"))
    m2 = torch.from_numpy(recipe.keep_mask((N, 5000), 0.1, "m2"))
    m3 = torch.from_numpy(recipe.keep_mask((N, 5000), 0.1, "m3"))
    masks = dict(m1=m1.cuda(), m2=m2.cuda(), l=ml.view(N * T, H).cuda())
    if mhb:
        masks["m3"] = m3.cuda()
    model.set_keep_masks(**masks)
    out = model.forward(img, q) if not mhb else model.forward(img, q, glove_matrix=glove)
    loss = torch.nn.KLDivLoss()(out, soft) if mhb else torch.nn.CrossEntropyLoss()(out, hard)
    loss.backward()

    drop = dict(m1=m1.view(N, L, 5000), m2=m2, m3=m3, l=ml.permute(1, 0, 2) if mhb else ml)
    o_out, g32, g64 = _oracle_pair(case, mhb, img, q, glove, soft if mhb else hard, drop=drop)
    assert rel_err(out.detach().cpu().numpy(), o_out.numpy()) <= OUT_TOL
    grad_parity(_named_grads(model), g32, g64)


def test_train_mode_philox_is_seeded_and_differentiable():
    vqa = _vqa()
    case = MHBCOATT_CASES[1]
    cfg, img, q, glove, _, soft = mfb_inputs(case, "cuda")
    model = _load(vqa.MHBCoAtt(cfg), case["salt"]).train()
    torch.manual_seed(7)
    a = model.forward(img, q)
    torch.manual_seed(7)
    b = model.forward(img, q)
    torch.manual_seed(8)
    c = model.forward(img, q)
    assert torch.equal(a, b) and not torch.equal(a, c)
    torch.nn.KLDivLoss()(a, soft).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_mfb_unit_softmax_switch_gives_live_attention():
    """reference_compat off: real softmaxes (over T and over the 196 regions) vs the oracle's mhb-style ladder."""
    vqa = _vqa()
    case = MFB_CASES[2]
    cfg, img, q, _, hard, _ = mfb_inputs(case, "cuda")
    model = _no_dropout_train(_load(vqa.MFB(cfg), case["salt"]))
    compat = model.forward(img, q)
    model.unit_softmax = False
    live = model.forward(img, q)
    assert not torch.allclose(compat, live)
    torch.nn.CrossEntropyLoss()(live, hard).backward()
    # with live softmaxes the image-projection weights receive a real gradient
    assert float(model.img_conv1d.weight.grad.abs().max()) > 0.0
    assert float(model.co_att_conv1.weight.grad.abs().max()) > 0.0


@pytest.mark.parametrize("mhb", [False, True])
def test_fold_norm_equals_materialised_normalisation(mhb):
    """F.normalize folded into co_att_conv1's GEMM epilogue (functions.NormLink, the default) against the round-2 form that
    writes fusion_normed and runs the scale / rowdot passes (`fold_norm = False`): same outputs to rounding, and BOTH forms'
    gradients inside the oracle's conditioning-aware bound.  MFB runs with live softmaxes here so that the co-attention
    branch carries a gradient (under the reference's singleton softmax it is exactly zero in both forms)."""
    vqa = _vqa()
    case = dict((MHBCOATT_CASES if mhb else MFB_CASES)[2])
    cfg, img, q, glove, hard, soft = mfb_inputs(case, "cuda")
    model = _no_dropout_train(_load((vqa.MHBCoAtt if mhb else vqa.MFB)(cfg), case["salt"]))
    if not mhb:
        model.unit_softmax = False
    outs = {}
    for fold in (True, False):
        model.fold_norm = fold
        model.zero_grad(set_to_none=True)
        n_scale = vqa.ops.stat("gemm_f32_tile128")
        vqa.ops.prof_reset(); vqa.ops.prof_enable(True)
        out = model.forward(img, q, glove_matrix=glove) if mhb else model.forward(img, q)
        (torch.nn.KLDivLoss()(out, soft) if mhb else torch.nn.CrossEntropyLoss()(out, hard)).backward()
        torch.cuda.synchronize()
        rep = vqa.ops.prof_report(); vqa.ops.prof_enable(False)
        # the folded form launches scale_rows / rowdot only for the final (N, 1000) blocks, never for the (N*L, 1000) tensor
        n_final = 2 if mhb else 1
        assert rep["scale_rows"][0] == (n_final if fold else n_final + 1), rep["scale_rows"]
        assert rep["rowdot"][0] == (n_final if fold else n_final + 1), rep["rowdot"]
        outs[fold] = (out.detach().clone(), {k: g.clone() for k, g in _named_grads(model).items()})
    assert rel_err(outs[True][0].cpu().numpy(), outs[False][0].cpu().numpy()) <= 2e-6
    if mhb:
        _, g32, g64 = _oracle_pair(case, True, img, q, glove, soft)
    else:
        res = []
        for dt in (torch.float32, torch.float64):
            sd = {k: v.to(dt).requires_grad_(True) for k, v in recipe_sd(O.mfb_shapes(cfg), case["salt"]).items()}
            o = O.mfb_forward(sd, cfg, img.cpu().to(dt), q.cpu(), live_softmax=True)
            O.ce_loss(o, hard.cpu()).backward()
            res.append({k: v.grad for k, v in sd.items()})
        g32, g64 = res
    for fold in (True, False):
        grad_parity(outs[fold][1], g32, g64, label="fold_norm=%s %s" % (fold, "mhb_coAtt" if mhb else "mfb live softmax"))


def test_full_size_batch_512_by_sample_independence():
    """BASELINE config 2 shapes (B=512, 196x2048, H=1024, T=14): MFB samples are independent, so
    rows of the B=512 GPU result must equal the oracle run on those samples alone; every sample's
    fused feature is unit-norm (F.normalize)."""
    vqa = _vqa()
    case = dict(name="b512", salt=77, N=512, model_name="mfb", glove=False,
                H=1024, E=300, D=2048, L=196, V=1000, A=1000, T=14)
    cfg = make_cfg(case)
    model = _load(vqa.MFB(cfg), case["salt"]).eval()
    g = torch.Generator().manual_seed(1234)
    img = torch.relu(torch.randn((512, 196, 2048), generator=g))
    q = torch.randint(1, 1000, (512, 14), generator=torch.Generator().manual_seed(1235))
    with torch.no_grad():
        logits = model.forward(img.cuda(), q.cuda()).cpu()
    pick = [0, 137, 300, 511]
    sd = recipe_sd(O.mfb_shapes(cfg), case["salt"])
    ref = O.mfb_forward(sd, cfg, img[pick], q[pick])
    assert rel_err(logits[pick].numpy(), ref.numpy()) <= OUT_TOL
    assert torch.isfinite(logits).all()


def test_full_dims_gradients_vs_oracle_n16():
    """full feature sizes, N=16, MHBCoAtt (every weight live): HIP gradients vs oracle gradients."""
    vqa = _vqa()
    case = dict(name="n16", salt=78, N=16, model_name="mhb_coAtt", glove=False,
                H=1024, E=300, D=2048, L=196, V=1000, A=1000, T=14)
    cfg, img, q, glove, hard, soft = mfb_inputs(case, "cuda")
    model = _no_dropout_train(_load(vqa.MHBCoAtt(cfg), case["salt"]))
    out = model.forward(img, q)
    torch.nn.KLDivLoss()(out, soft).backward()
    o_out, g32, g64 = _oracle_pair(case, True, img, q, None, soft)
    assert rel_err(out.detach().cpu().numpy(), o_out.numpy()) <= OUT_TOL
    grad_parity(_named_grads(model), g32, g64)


@pytest.mark.parametrize("case", [pytest.param(c, id=c["name"]) for c in MHB_CASES])
def test_mhb_matches_reference_golden(case):
    """MHB (mhb_coAtt.py:153-217) against goldens of the reference class made executable by the two edits stated in
    tests/golden/make_golden.py (shim 6): outputs 1e-4, gradient digests, and grad_parity against the pinned oracle."""
    vqa = _vqa()
    gold = load_golden("mhb_" + case["name"])
    cfg = make_cfg(case)
    N, T = case["N"], case["T"]
    model = _no_dropout_train(_load(vqa.MHB(cfg), case["salt"]))
    img = torch.from_numpy(recipe.img_features(N, cfg.img_feature_dim, cfg.img_feature_channel, case["salt"]))
    qn = recipe.question_tokens(N, T, cfg.q_vocab_size, case["salt"])
    q, ql = torch.from_numpy(qn), torch.from_numpy(recipe.question_lengths(qn))
    assert np.array_equal(ql.numpy(), gold["q_length"])
    soft = torch.from_numpy(recipe.soft_answers(N, cfg.a_vocab_size, case["salt"]))
    out = model.forward(img.cuda(), q.cuda(), ql.cuda())
    assert rel_err(out.detach().cpu().numpy(), gold["out"]) <= OUT_TOL
    loss = torch.nn.KLDivLoss()(out, soft.cuda())
    assert abs(loss.item() - float(gold["loss"])) <= 2e-4 * max(1e-3, abs(float(gold["loss"])))
    loss.backward()
    check_grads(_named_grads(model), gold, GRAD_TOL)
    res = []
    for dt in (torch.float32, torch.float64):
        sd = {k: v.to(dt).requires_grad_(True) for k, v in recipe_sd(O.mhb_shapes(cfg), case["salt"]).items()}
        O.kldiv_loss(O.mhb_forward(sd, cfg, img.to(dt), q, ql), soft.to(dt)).backward()
        res.append({k: v.grad for k, v in sd.items()})
    grad_parity(_named_grads(model), res[0], res[1])


def test_mhb_module_vs_oracle():
    """MHB vs the oracle restatement on one more seeded case (the oracle itself is pinned: test_oracle_golden.py)."""
    vqa = _vqa()
    import types
    cfg = types.SimpleNamespace(q_vocab_size=50, a_vocab_size=30, emb_dim=24, hidden_dim=64, num_layers=1,
                                img_feature_channel=96, img_feature_dim=196, model_name="mhb", glove=False)
    N, T = 5, 7
    model = _no_dropout_train(_load(vqa.MHB(cfg), 61))
    img = torch.from_numpy(recipe.img_features(N, 196, 96, 61))
    qn = recipe.question_tokens(N, T, 50, 61)
    q, ql = torch.from_numpy(qn), torch.from_numpy(recipe.question_lengths(qn))
    soft = torch.from_numpy(recipe.soft_answers(N, 30, 61))
    out = model.forward(img.cuda(), q.cuda(), ql.cuda())
    torch.nn.KLDivLoss()(out, soft.cuda()).backward()
    res = []
    for dt in (torch.float32, torch.float64):
        sd = {k: v.to(dt).requires_grad_(True) for k, v in recipe_sd(O.mhb_shapes(cfg), 61).items()}
        o = O.mhb_forward(sd, cfg, img.to(dt), q, ql)
        O.kldiv_loss(o, soft.to(dt)).backward()
        res.append((o.detach(), {k: v.grad for k, v in sd.items()}))
    assert rel_err(out.detach().cpu().numpy(), res[0][0].numpy()) <= OUT_TOL
    grad_parity(_named_grads(model), res[0][1], res[1][1])


def test_state_dict_keys_match_reference_layout():
    vqa = _vqa()
    cfg = make_cfg(MFB_CASES[0])
    assert set(vqa.MFB(cfg).state_dict().keys()) == set(O.mfb_shapes(cfg).keys())
    cfgm = make_cfg(MFB_CASES[4])
    assert set(vqa.MFB(cfgm).state_dict().keys()) == set(O.mfb_shapes(cfgm).keys())
    cfgh = make_cfg(MHBCOATT_CASES[3])
    m = vqa.MHBCoAtt(cfgh)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == \
        {k: tuple(v) for k, v in O.mfb_shapes(cfgh, mhb=True).items()}


def test_config3_mhbcoatt_full_batch_512_prefix_causality():
    """BASELINE config 3 shapes (MHBCoAtt, B=512, fp32 here): the batch-axis LSTM recursion
    (mhb_coAtt.py:72-74) is causal, so rows 0..3 of the B=512 result equal the oracle run on the first
    4 samples alone; log-probs normalise; gradients are finite for every parameter."""
    vqa = _vqa()
    case = dict(name="c3", salt=79, N=512, model_name="mhb_coAtt", glove=False,
                H=1024, E=300, D=2048, L=196, V=1000, A=1000, T=14)
    cfg = make_cfg(case)
    model = _no_dropout_train(_load(vqa.MHBCoAtt(cfg), case["salt"]))
    g = torch.Generator().manual_seed(1234)
    img = torch.relu(torch.randn((512, 196, 2048), generator=g))
    q = torch.randint(1, 1000, (512, 14), generator=torch.Generator().manual_seed(1235))
    soft = torch.softmax(torch.randn((512, 1000), generator=torch.Generator().manual_seed(1236)), 1)
    out = model.forward(img.cuda(), q.cuda())
    torch.nn.KLDivLoss()(out, soft.cuda()).backward()
    assert torch.allclose(out.detach().exp().sum(1).cpu(), torch.ones(512), atol=1e-4)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    sd = recipe_sd(O.mfb_shapes(cfg, mhb=True), case["salt"])
    ref = O.mhbcoatt_forward(sd, cfg, img[:4], q[:4])
    assert rel_err(out[:4].detach().cpu().numpy(), ref.numpy()) <= OUT_TOL


@pytest.mark.parametrize("mhb,dtype", [(False, "fp32"), (True, "fp32"), (True, "bf16"), (True, "bf16-all"), (False, "bf16-all")])
def test_training_step_is_bitwise_reproducible(mhb, dtype):
    """No atomics on data anywhere on the path (split-K slabs, two-stage column reductions, fixed-order
    sums in the LSTM / glimpse / loss kernels) and Philox dropout keyed by torch's CPU generator: the same
    seed gives bit-identical logits and gradients, run after run, with dropout ACTIVE (full dims, N = 6)."""
    import vqa_amd
    case = dict((MHBCOATT_CASES if mhb else MFB_CASES)[-1 if mhb else -2], N=6, salt=77)
    cfg, img, q, glove, hard, soft = mfb_inputs(case, "cuda")
    model = (vqa_amd.MHBCoAtt if mhb else vqa_amd.MFB)(cfg)
    model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), case["salt"]))
                           for k, v in model.state_dict().items()})
    model = model.cuda().train()
    model.gemm_dtype = dtype
    for m in model.modules():                  # nn.Dropout after the LSTM is torch's (seeded below too)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.1
    crit = vqa_amd.KLDivLoss() if mhb else vqa_amd.CrossEntropyLoss()
    runs = []
    for _ in range(2):
        torch.manual_seed(1234)
        torch.cuda.manual_seed(1234)
        model.zero_grad(set_to_none=True)
        out = model.forward(img, q)
        loss = crit(out, soft if mhb else hard)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


@pytest.mark.parametrize("multilayer", [False, True])
def test_pruned_mode_is_bit_identical_to_faithful(multilayer):
    """MFB.pruned skips the provably dead work under the singleton-axis softmaxes (mfb.py:84,118): logits and
    every gradient must be BIT-identical to the faithful execution, the 12 (16) dead tensors exactly zero --
    with dropout active and the same seed (the regions' dropout draw is still consumed)."""
    import vqa_amd
    case = dict(MFB_CASES[-1 if multilayer else -2], N=5, salt=55)
    cfg, img, q, glove, hard, soft = mfb_inputs(case, "cuda")
    model = vqa_amd.MFB(cfg)
    model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), case["salt"]))
                           for k, v in model.state_dict().items()})
    model = model.cuda().train()                # every dropout active, the LSTM-output one (HIP, seeded like the others) included
    res = {}
    for pruned in (False, True):
        model.pruned = pruned
        torch.manual_seed(99)
        model.zero_grad(set_to_none=True)
        out = model.forward(img, q)
        torch.nn.CrossEntropyLoss()(out, hard).backward()
        res[pruned] = (out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
        del out          # (the faithful graph builds img_conv1d's node under the side stream; not alive across the switch: tests/test_gpu_bf16.py)
    assert torch.equal(res[True][0], res[False][0])
    dead = ("ques_att_conv", "ques_att_multiconv", "ques_proj1", "img_conv1d", "co_att_conv", "co_att_multiconv")
    for k, g in res[False][1].items():
        assert torch.equal(res[True][1][k], g), k
        if k.startswith(dead):
            assert float(g.abs().max()) == 0.0, k
    assert any(float(g.abs().max()) > 0 for k, g in res[True][1].items() if not k.startswith(dead))


def test_config2_mfb_batch_512_gradients_vs_oracle_live_softmax(monkeypatch):
    """(round 5: every autograd node of this step is also checked against its own fp64 evaluation, tests/node_harness.py,
    and every tensor whose model-level fp32 noise exceeds 5 % of its gradient must be covered by such a check.)
    BASELINE config 2 at its full batch, gradients: MFB, B=512, 196x2048, fp32, with
    `unit_softmax=False` so that EVERY tensor is live -- in faithful mode dY == 0 and the image projection's
    weight gradient multiplies zeros.  The HIP gradients (large-tile forward GEMM, K=100352 split-K weight
    gradient, fusion kernels at N*L = 100352 rows) against the oracle in fp32 and fp64 on the same inputs,
    grad_parity criterion; forward 1e-4.  The oracle needs ~25 GB of host memory and 1-2 minutes."""
    vqa = _vqa()
    case = dict(name="b512g", salt=81, N=512, model_name="mfb", glove=False,
                H=1024, E=300, D=2048, L=196, V=1000, A=1000, T=14)
    cfg = make_cfg(case)
    model = _no_dropout_train(_load(vqa.MFB(cfg), case["salt"]))
    model.unit_softmax = False
    g = torch.Generator().manual_seed(1234)
    img = torch.relu(torch.randn((512, 196, 2048), generator=g))
    q = torch.randint(1, 1000, (512, 14), generator=torch.Generator().manual_seed(1235))
    a = torch.randint(0, 1000, (512,), generator=torch.Generator().manual_seed(1236))
    recd = Recorder(monkeypatch, vqa.functions, ALL_NODES)
    out = model.forward(img.cuda(), q.cuda())
    torch.nn.CrossEntropyLoss()(out, a.cuda()).backward()
    torch.cuda.synchronize()
    grads = _named_grads(model)
    _, covered = check_every_node(model, recd, "config 2 (MFB live softmax, fp32) node checks at B=512", skip_params=())
    del recd
    res = []
    for dt in (torch.float32, torch.float64):
        sd = {k: v.to(dt).requires_grad_(True) for k, v in recipe_sd(O.mfb_shapes(cfg), case["salt"]).items()}
        o = O.mfb_forward(sd, cfg, img.to(dt), q, live_softmax=True)
        O.ce_loss(o, a).backward()
        res.append((o.detach(), {k: v.grad for k, v in sd.items()}))
        del sd, o
    assert rel_err(out.detach().cpu().numpy(), res[0][0].numpy()) <= OUT_TOL
    assert float(grads["img_conv1d.weight"].abs().max()) > 0.0 and float(grads["co_att_conv1.weight"].abs().max()) > 0.0
    assert covered == set(grads)
    grad_parity(grads, res[0][1], res[1][1], node_checked=covered)


def test_config2_mfb_batch_512_faithful_gradients_vs_oracle(monkeypatch):
    """The exact mode bench.py times (VERDICT r02 weak #1a): MFB-baseline, B=512, fp32, FAITHFUL (`unit_softmax=True`: the
    reference's singleton-axis softmaxes, mfb.py:84,118), forward + backward, against the oracle in fp32 and fp64 on the
    same inputs.  Live tensors (embedding, LSTM, ques_proj2, img_proj2, linear_pred) by grad_parity; the 12 tensors whose
    gradient the reference's autograd computes as exact zeros (both attention MLPs, ques_proj1, img_conv1d) must be
    EXACTLY 0.0 here too -- the image projection's weight-gradient GEMM multiplies a dP of zeros.  Train mode with every
    dropout rate 0 (the LSTM backward needs train mode; eval-mode goldens are reproduced this way)."""
    vqa = _vqa()
    case = dict(name="b512f", salt=84, N=512, model_name="mfb", glove=False,
                H=1024, E=300, D=2048, L=196, V=1000, A=1000, T=14)
    cfg = make_cfg(case)
    model = _no_dropout_train(_load(vqa.MFB(cfg), case["salt"]))
    assert model.unit_softmax is True and model.pruned is False and model.gemm_dtype == "fp32"
    model.overlap_streams = "same-stream"            # the stream configuration bench.py runs
    img = torch.relu(torch.randn((512, 196, 2048), generator=torch.Generator().manual_seed(1234)))
    q = torch.randint(1, 1000, (512, 14), generator=torch.Generator().manual_seed(1235))
    a = torch.randint(0, 1000, (512,), generator=torch.Generator().manual_seed(1236))
    recd = Recorder(monkeypatch, vqa.functions, ALL_NODES)
    out = model.forward(img.cuda(), q.cuda())
    vqa.CrossEntropyLoss()(out, a.cuda()).backward()      # the HIP criterion, as in bench.py
    torch.cuda.synchronize()
    grads = _named_grads(model)
    kinds = [r["cls"].__name__ for r in recd.records]
    assert kinds == ["ImgProjDeferFn", "EmbedTanhFn", "LstmBatchFn", "DropoutBTFn", "AttHeadFn", "LinearFn", "MfbFuseFn", "AttHeadFn",
                     "FinalMfbFn", "LinearFn"], kinds
    # every node of the step bench.py times against its own fp64 evaluation (the dead ones: exact zeros)
    _, covered = check_every_node(model, recd, "config 2 (MFB faithful, fp32, the headline step) node checks at B=512", skip_params=())
    del recd
    res = []
    for dt in (torch.float32, torch.float64):
        sd = {k: v.to(dt).requires_grad_(True) for k, v in recipe_sd(O.mfb_shapes(cfg), case["salt"]).items()}
        o = O.mfb_forward(sd, cfg, img.to(dt), q)
        O.ce_loss(o, a).backward()
        res.append((o.detach(), {k: v.grad for k, v in sd.items()}))
        del sd, o
    assert rel_err(out.detach().cpu().numpy(), res[0][0].numpy()) <= OUT_TOL
    assert rel_err(out.detach().cpu().numpy(), res[1][0].float().numpy()) <= OUT_TOL
    dead = ("ques_att_conv1.", "ques_att_conv2.", "ques_proj1.", "img_conv1d.", "co_att_conv1.", "co_att_conv2.")
    n_dead = 0
    for k, g in grads.items():
        if k.startswith(dead):
            n_dead += 1
            assert g is not None and float(g.abs().max()) == 0.0, (k, "dead tensor must receive exact zeros")
            assert res[1][1][k] is None or float(res[1][1][k].abs().max()) == 0.0, (k, "oracle disagrees that it is dead")
        else:
            assert float(g.abs().max()) > 0.0, (k, "live tensor without a gradient")
    assert n_dead == 12
    assert covered == set(grads)
    grad_parity({k: g for k, g in grads.items() if not k.startswith(dead)}, res[0][1], res[1][1], node_checked=covered)


def test_config3_shapes_mhbcoatt_batch_512_fp32_gradients_vs_oracle(monkeypatch):
    """MHBCoAtt at the full B=512 in fp32, every output row and every gradient against the oracle run on the SAME
    512-sample batch in fp32 and fp64 (the batch-axis LSTM recursion, mhb_coAtt.py:72-74, makes row n depend on rows
    0..n-1, so only the whole batch exercises the 512-step chain and its backward): forward 1e-4, gradients by
    grad_parity.  Live softmax over T and L (MHBCoAtt normalises over the right axes), KLDiv loss as solver.py:27.
    The oracle needs ~40 GB of host memory and 1-2 minutes."""
    vqa = _vqa()
    case = dict(name="c3g", salt=83, N=512, model_name="mhb_coAtt", glove=False,
                H=1024, E=300, D=2048, L=196, V=1000, A=1000, T=14)
    cfg = make_cfg(case)
    model = _no_dropout_train(_load(vqa.MHBCoAtt(cfg), case["salt"]))
    img = torch.relu(torch.randn((512, 196, 2048), generator=torch.Generator().manual_seed(1234)))
    q = torch.randint(1, 1000, (512, 14), generator=torch.Generator().manual_seed(1235))
    soft = torch.softmax(torch.randn((512, 1000), generator=torch.Generator().manual_seed(1236)), 1)
    recd = Recorder(monkeypatch, vqa.functions, ALL_NODES)
    out = model.forward(img.cuda(), q.cuda())
    torch.nn.KLDivLoss()(out, soft.cuda()).backward()
    torch.cuda.synchronize()
    grads = _named_grads(model)
    _, covered = check_every_node(model, recd, "config 3 shapes (MHBCoAtt, fp32) node checks at B=512", skip_params=())
    del recd
    res = []
    for dt in (torch.float32, torch.float64):
        sd = {k: v.to(dt).requires_grad_(True) for k, v in recipe_sd(O.mfb_shapes(cfg, mhb=True), case["salt"]).items()}
        o = O.mhbcoatt_forward(sd, cfg, img.to(dt), q)
        O.kldiv_loss(o, soft.to(dt)).backward()
        res.append((o.detach(), {k: v.grad for k, v in sd.items()}))
        del sd, o
    assert rel_err(out.detach().cpu().numpy(), res[0][0].numpy()) <= OUT_TOL
    assert rel_err(out.detach().cpu().numpy(), res[1][0].float().numpy()) <= OUT_TOL
    assert float(grads["img_conv1d.weight"].abs().max()) > 0.0 and float(grads["lstm.weight_hh_l0"].abs().max()) > 0.0
    assert covered == set(grads)
    grad_parity(grads, res[0][1], res[1][1], node_checked=covered)


@pytest.mark.parametrize("bf16_mode", ["bf16", "bf16-all"])
def test_config3_mhbcoatt_batch_512_bf16_mode(bf16_mode):
    """BASELINE config 3 as stated: MHBCoAtt, B=512, gemm_dtype='bf16' (bf16 operands in img_conv1d /
    co_att_conv1 through gemm_bf16_big.hip incl. its weight-gradient layout, bf16 image / projection storage,
    bf16 dP), fwd+bwd.  Rows 0..3 against the fp32 oracle on the 4-sample prefix (the batch-axis recursion is
    causal) at the stated bf16 tolerance 3e-2; log-probs normalise; every gradient finite; and the
    bf16 step tracks the same model's fp32 step (outputs 3e-2; well-conditioned gradients 10 % in norm)."""
    vqa = _vqa()
    case = dict(name="c3b", salt=82, N=512, model_name="mhb_coAtt", glove=False,
                H=1024, E=300, D=2048, L=196, V=1000, A=1000, T=14)
    cfg = make_cfg(case)
    model = _no_dropout_train(_load(vqa.MHBCoAtt(cfg), case["salt"]))
    g = torch.Generator().manual_seed(1234)
    img = torch.relu(torch.randn((512, 196, 2048), generator=g))
    q = torch.randint(1, 1000, (512, 14), generator=torch.Generator().manual_seed(1235))
    soft = torch.softmax(torch.randn((512, 1000), generator=torch.Generator().manual_seed(1236)), 1)
    img_d, q_d, soft_d = img.cuda(), q.cuda(), soft.cuda()
    res = {}
    img_r = img_d.to(torch.bfloat16).float()          # the image tensor rounded to bf16 values, fed to the FP32 kernels: the conditioning yardstick below
    for mode in (bf16_mode, "fp32", "fp32-rounded-image"):
        model.gemm_dtype = "fp32" if mode.startswith("fp32") else mode
        model.zero_grad(set_to_none=True)
        x = (vqa.ops.cast_bf16(img_d.view(-1, 2048)).view(img_d.shape) if not mode.startswith("fp32")   # bf16 feature storage
             else (img_r if mode == "fp32-rounded-image" else img_d))
        out = model.forward(x, q_d)
        torch.nn.KLDivLoss()(out, soft_d).backward()
        torch.cuda.synchronize()
        res[mode] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
        del out                              # (no graph of the previous mode alive across the switch: tests/test_gpu_bf16.py)
    out, gb = res[bf16_mode]
    assert torch.allclose(out.exp().sum(1).cpu(), torch.ones(512), atol=1e-4)
    assert all(torch.isfinite(v).all() for v in gb.values())
    sd = recipe_sd(O.mfb_shapes(cfg, mhb=True), case["salt"])
    ref = O.mhbcoatt_forward(sd, cfg, img[:4], q[:4])
    assert rel_err(out[:4].cpu().numpy(), ref.numpy()) <= 3e-2
    assert rel_err(res["fp32"][0][:4].cpu().numpy(), ref.numpy()) <= OUT_TOL
    assert not torch.equal(out, res["fp32"][0])                       # the bf16 kernels really ran
    assert rel_err(out.cpu().numpy(), res["fp32"][0].cpu().numpy()) <= 3e-2
    # gradients, bf16 step vs fp32 step of the same model.  The classifier -- the one tensor downstream of every signed square
    # root -- within 10 % in norm.  Every other gradient passes through 0.5*|s|^-1/2 of the final MFB blocks' 512 x 2000 pooled
    # sums (and, further upstream, of the 512 x 196 000 regional ones), whose near-zero entries ANY rounding of an upstream
    # product replaces by noise (co_att_conv1 76 %, word_embedding 65 %; at N = 2 the same tensors stay within 10 %,
    # test_gpu_bf16.py): a property of the model's loss surface at this batch.  Round 5 asserts those tensors too, against a
    # yardstick that has no bf16 kernel in it: the FP32 kernels on the image tensor rounded to bf16 values (one of the several
    # roundings the bf16 mode performs).  A tensor may deviate from the fp32 step by at most 2.5x what that single input rounding
    # already causes (or 10 %; measured 0.9-1.5x: the rounding moves these tensors by 45-130 %, the bf16 mode by 50-115 %): a wrong
    # bf16 kernel on a tensor the rounding moves by 5 % fails; where the rounding alone moves a
    # tensor by 70 % nothing at model level can discriminate -- those are the tensors tests/test_gpu_bf16_nodes.py covers node by
    # node against fp64.
    from golden_util import _report_parity
    checked = ("linear_pred",)
    g_round = res["fp32-rounded-image"][1]
    worst, worst_k, info = 0.0, "-", []
    worst_y, worst_yk = 0.0, "-"
    for k, g32 in res["fp32"][1].items():
        if float(g32.norm()) < 1e-9:
            continue
        d = float((gb[k] - g32).norm()) / float(g32.norm())
        dr = float((g_round[k] - g32).norm()) / float(g32.norm())
        info.append("%s=%.3f(%.3f)" % (k, d, dr))
        bound = 0.1 if k.startswith(checked) else max(0.1, 2.5 * dr)
        assert d <= bound, (k, d, "bound", bound, "input rounding alone", dr)
        if k.startswith(checked) and d > worst:
            worst, worst_k = d, k
        if d / bound > worst_y:
            worst_y, worst_yk = d / bound, k
    print("config 3 bf16 vs fp32 gradients at B=512, relative deviation per tensor (that of the fp32 step on the bf16-rounded image): "
          + " ".join(info))
    _report_parity("test_config3_mhbcoatt_batch_512_bf16_mode[%s] (bf16 vs fp32 grads, bound 0.1)" % bf16_mode, worst / 0.1, worst_k)
    _report_parity("test_config3_mhbcoatt_batch_512_bf16_mode[%s] (every tensor vs max(0.1, 2.5 x input-rounding deviation))" % bf16_mode,
                   worst_y, worst_yk)
