"""Pin the CPU oracle (oracle/ref_torch.py) against vectors captured from the
imported reference (tests/golden/make_golden.py).  CPU only, no GPU needed.

Tolerances
  forward outputs : 2e-5 relative (same fp32 ops, different association in the
                    hand-written LSTM / einsum) -- well inside north_star's 1e-4.
  gradients       : 3e-3 on norm / sampled-entry digests.  The signed square
                    root's derivative 0.5*|s|^-1/2 (mfb.py:104,133) amplifies fp32
                    rounding wherever a pooled sum s is close to 0; on these cases
                    the REFERENCE's own fp32 gradient is 2e-4 away from its fp64
                    value, and any re-association moves it by up to ~1e-3.  The
                    fp64 run of the oracle (test_*_fp64) pins the restatement to
                    the REFERENCE RUN IN FLOAT64 (model.double(); `out64`, `g64*`
                    digests) at 1e-9 / 1e-8: the restatement is exact, only fp32
                    rounding differs.
"""
import numpy as np
import pytest
import torch

import recipe
from cases import (MFB_CASES, MHBCOATT_CASES, MHB_CASES, HIE_CASES, ATTNET_CASES, IBOW_CASES,
                   ATT_MODULE_CASES, make_cfg)
from golden_util import (load_golden, recipe_sd, mfb_inputs, rel_err, check_tensor_digest,
                         check_grads, check_grads64)
from oracle import ref_torch as O

OUT_TOL = 2e-5
GRAD_TOL = 3e-3


def _fast(cases):
    # the two "full" cases cost a few seconds each; keep them, they are the real shapes
    return [pytest.param(c, id=c["name"]) for c in cases]


@pytest.mark.parametrize("case", _fast(MFB_CASES))
def test_mfb_oracle_matches_reference(case):
    gold = load_golden("mfb_" + case["name"])
    cfg, img, q, _, hard, _ = mfb_inputs(case)
    sd = recipe_sd(O.mfb_shapes(cfg), case["salt"], requires_grad=True)
    t = O.mfb_forward(sd, cfg, img, q, return_all=True)
    assert rel_err(t["logits"].detach().numpy(), gold["out"]) <= OUT_TOL
    loss = O.ce_loss(t["logits"], hard)
    assert abs(loss.item() - float(gold["loss"])) <= OUT_TOL * max(1.0, abs(float(gold["loss"])))
    check_tensor_digest("ques_att_feature", t["qa"], gold, OUT_TOL)
    # reference layout of fusion_normed is (N,1000,L,1); the oracle keeps (N,L,1000)
    check_tensor_digest("fusion_normed", t["Y"].permute(0, 2, 1).contiguous(), gold, OUT_TOL)
    check_tensor_digest("co_att_logits", t["clog"].permute(0, 2, 1).contiguous(), gold, 5e-5)
    check_tensor_digest("co_att_feature", t["va"], gold, OUT_TOL)
    check_tensor_digest("att_normed", t["y"], gold, OUT_TOL)
    loss.backward()
    check_grads({k: v.grad for k, v in sd.items()}, gold, GRAD_TOL)


def test_mfb_dead_weights_have_exactly_zero_grad():
    """softmax over the singleton axis kills these gradients exactly."""
    gold = load_golden("mfb_small_n3")
    dead = ["ques_att_conv1", "ques_att_conv2", "ques_proj1", "img_conv1d", "co_att_conv1", "co_att_conv2"]
    for d in dead:
        for s in ("weight", "bias"):
            assert float(gold["gnorm/%s.%s" % (d, s)]) == 0.0


@pytest.mark.parametrize("case", _fast(MHBCOATT_CASES))
def test_mhbcoatt_oracle_matches_reference(case):
    gold = load_golden("mhbcoatt_" + case["name"])
    cfg, img, q, glove, _, soft = mfb_inputs(case)
    sd = recipe_sd(O.mfb_shapes(cfg, mhb=True), case["salt"], requires_grad=True)
    t = O.mhbcoatt_forward(sd, cfg, img, q, glove=glove, return_all=True)
    assert rel_err(t["out"].detach().numpy(), gold["out"]) <= OUT_TOL
    loss = O.kldiv_loss(t["out"], soft)
    assert abs(loss.item() - float(gold["loss"])) <= 1e-4 * max(1e-3, abs(float(gold["loss"])))
    check_tensor_digest("ques_att_feature", t["qa"], gold, OUT_TOL)
    check_tensor_digest("fusion_normed", t["Y"].permute(0, 2, 1).contiguous(), gold, OUT_TOL)
    check_tensor_digest("co_att_logits", t["clog"].permute(0, 2, 1).contiguous(), gold, 5e-5)
    check_tensor_digest("co_att_feature", t["va"], gold, OUT_TOL)
    loss.backward()
    check_grads({k: v.grad for k, v in sd.items()}, gold, GRAD_TOL)


def _mhb_inputs(case):
    cfg = make_cfg(case)
    N, T = case["N"], case["T"]
    img = torch.from_numpy(recipe.img_features(N, cfg.img_feature_dim, cfg.img_feature_channel, case["salt"]))
    qn = recipe.question_tokens(N, T, cfg.q_vocab_size, case["salt"])
    return cfg, img, torch.from_numpy(qn), torch.from_numpy(recipe.question_lengths(qn)), \
        torch.from_numpy(recipe.soft_answers(N, cfg.a_vocab_size, case["salt"]))


@pytest.mark.parametrize("case", _fast(MHB_CASES))
def test_mhb_oracle_matches_reference(case):
    """MHB (mhb_coAtt.py:153-217), pinned since round 4: the goldens come from the reference class compiled from its own
    text with the two edits that make it executable (tests/golden/make_golden.py::load_mhb_class)."""
    gold = load_golden("mhb_" + case["name"])
    cfg, img, q, ql, soft = _mhb_inputs(case)
    assert np.array_equal(ql.numpy(), gold["q_length"])
    sd = recipe_sd(O.mhb_shapes(cfg), case["salt"], requires_grad=True)
    t = O.mhb_forward(sd, cfg, img, q, ql, return_all=True)
    assert rel_err(t["out"].detach().numpy(), gold["out"]) <= OUT_TOL
    loss = O.kldiv_loss(t["out"], soft)
    assert abs(loss.item() - float(gold["loss"])) <= 1e-4 * max(1e-3, abs(float(gold["loss"])))
    check_tensor_digest("lstm_out", t["last"], gold, OUT_TOL)
    check_tensor_digest("i_mean_pooled", t["i_mean"], gold, OUT_TOL)
    check_tensor_digest("mhb_12", t["y"], gold, OUT_TOL)
    loss.backward()
    check_grads({k: v.grad for k, v in sd.items()}, gold, GRAD_TOL)


@pytest.mark.parametrize("case", _fast(MHB_CASES))
def test_mhb_oracle_fp64_matches_reference_fp64(case):
    gold = load_golden("mhb_" + case["name"])
    cfg, img, q, ql, soft = _mhb_inputs(case)
    sd = {k: v.double().requires_grad_(True) for k, v in recipe_sd(O.mhb_shapes(cfg), case["salt"]).items()}
    out = O.mhb_forward(sd, cfg, img.double(), q, ql)
    assert rel_err(out.detach().numpy(), gold["out64"]) <= 1e-9
    loss = O.kldiv_loss(out, soft.double())
    assert abs(loss.item() - float(gold["loss64"])) <= 1e-10 * max(1.0, abs(float(gold["loss64"])))
    loss.backward()
    check_grads64({k: v.grad for k, v in sd.items()}, gold)


def test_mhbcoatt_recurs_over_the_batch_axis():
    """mhb_coAtt.py:72-74: changing sample 0's question changes every later row."""
    case = MHBCOATT_CASES[2]
    cfg, img, q, glove, _, _ = mfb_inputs(case)
    sd = recipe_sd(O.mfb_shapes(cfg, mhb=True), case["salt"])
    a = O.mhbcoatt_forward(sd, cfg, img, q)
    q2 = q.clone()
    q2[0, 0] = (q2[0, 0] % (cfg.q_vocab_size - 1)) + 1
    b = O.mhbcoatt_forward(sd, cfg, img, q2)
    assert all(float((a[i] - b[i]).abs().max()) > 0 for i in range(case["N"]))
    q3 = q.clone()
    q3[-1, 0] = (q3[-1, 0] % (cfg.q_vocab_size - 1)) + 1
    c = O.mhbcoatt_forward(sd, cfg, img, q3)
    assert all(float((a[i] - c[i]).abs().max()) == 0 for i in range(case["N"] - 1))
    assert float((a[-1] - c[-1]).abs().max()) > 0


@pytest.mark.parametrize("case", _fast(HIE_CASES))
def test_hiecoatten_oracle_matches_reference(case):
    gold = load_golden("hie_" + case["name"])
    sd = recipe_sd(O.hiecoatten_shapes(case["img_size"], case["V"], case["E"], case["A"]),
                   case["salt"], requires_grad=True)
    N = case["N"]
    img = torch.from_numpy(recipe.img_features(N, case["L"], case["img_size"], case["salt"]))
    q = torch.from_numpy(recipe.question_tokens(N, case["T"], case["V"], case["salt"]))
    x, av, aq = O.hiecoatten_forward(sd, img, q)
    assert rel_err(x.detach().numpy(), gold["x"]) <= OUT_TOL
    assert rel_err(av.detach().numpy(), gold["av"].reshape(N, -1)) <= OUT_TOL
    assert rel_err(aq.detach().numpy(), gold["aq"].reshape(N, -1)) <= OUT_TOL
    ans = torch.from_numpy(recipe.hard_answers(N, case["A"], case["salt"]))
    loss = O.ce_loss(x, ans)
    assert abs(loss.item() - float(gold["loss"])) <= OUT_TOL * max(1.0, float(gold["loss"]))
    loss.backward()
    check_grads({k: v.grad for k, v in sd.items()}, gold, GRAD_TOL)
    assert "gnone/fc_Wbq.weight" in gold and sd["fc_Wbq.weight"].grad is None


@pytest.mark.parametrize("case", _fast(ATTNET_CASES))
def test_attentionnet_oracle_matches_reference(case):
    gold = load_golden("attnet_" + case["name"])
    sd = recipe_sd(O.attentionnet_shapes(case["L"], case["T"], case["img_size"], case["V"],
                                         case["E"], case["att_num"], case["A"]),
                   case["salt"], requires_grad=True)
    N = case["N"]
    img = torch.from_numpy(recipe.img_features(N, case["L"], case["img_size"], case["salt"]))
    q = torch.from_numpy(recipe.question_tokens(N, case["T"], case["V"], case["salt"], pad_tail=False))
    x, qa, ia = O.attentionnet_forward(sd, img, q, att_num=case["att_num"])
    assert rel_err(qa.detach().numpy(), gold["que_att"]) <= OUT_TOL
    assert rel_err(ia.detach().numpy(), gold["img_att"]) <= OUT_TOL
    assert rel_err(x.detach().numpy(), gold["x"]) <= 2e-4     # BatchNorm over N<=4 rows amplifies
    ans = torch.from_numpy(recipe.hard_answers(N, case["A"], case["salt"]))
    loss = O.ce_loss(x, ans)
    loss.backward()
    check_grads({k: v.grad for k, v in sd.items()}, gold, 2e-3)


@pytest.mark.parametrize("case", _fast(IBOW_CASES))
def test_ibowimg_oracle_matches_reference(case):
    gold = load_golden("ibow_" + case["name"])
    E = case["E"]
    shapes = {"img_emb.weight": (E, case["img_size"]), "img_emb.bias": (E,),
              "img_bn.weight": (E,), "img_bn.bias": (E,),
              "que_emb.weight": (case["V"], E),
              "fc.weight": (case["A"], 2 * E), "fc.bias": (case["A"],)}
    sd = recipe_sd(shapes, case["salt"], requires_grad=True)
    N = case["N"]
    img = torch.from_numpy(recipe.sym_tensor((N, case["img_size"]), 1.0, recipe.name_seed("ibow_img", case["salt"])))
    q = torch.from_numpy(recipe.question_tokens(N, case["T"], case["V"], case["salt"]))
    x = O.ibowimg_forward(sd, img, q)
    assert rel_err(x.detach().numpy(), gold["x"]) <= 1e-4
    ans = torch.from_numpy(recipe.hard_answers(N, case["A"], case["salt"]))
    O.ce_loss(x, ans).backward()
    check_grads({k: v.grad for k, v in sd.items()}, gold, 2e-3)


@pytest.mark.parametrize("case", _fast(ATT_MODULE_CASES))
def test_attention_modules_oracle_matches_reference(case):
    gold = load_golden("mod_" + case["name"])
    kind, Dm = case["kind"], case["D"]
    N, L, T = case["N"], case["L"], case["T"]
    f1 = torch.from_numpy(recipe.sym_tensor((N, L, Dm), 1.0, recipe.name_seed("f1", case["salt"]))).requires_grad_()
    f2 = torch.from_numpy(recipe.sym_tensor((N, T, Dm), 1.0, recipe.name_seed("f2", case["salt"]))).requires_grad_()
    if kind == "attention_1":
        sd = recipe_sd({"fc.weight": (1, Dm), "fc.bias": (1,)}, case["salt"], requires_grad=True)
        fh, att = O.attention_1(sd, "", f1, f2)
        assert rel_err(fh.detach().numpy(), gold["f_hat"]) <= OUT_TOL
        assert rel_err(att.detach().numpy(), gold["att"]) <= OUT_TOL
        ((fh * fh).sum() + (att * att).sum()).backward()
    elif kind == "attention_2":
        sd = recipe_sd({"fc1.weight": (Dm, Dm), "fc2.weight": (1, Dm), "fc2.bias": (1,)},
                       case["salt"], requires_grad=True)
        fh, att = O.attention_2(sd, "", f1, f2)
        assert rel_err(fh.detach().numpy(), gold["f_hat"]) <= OUT_TOL
        assert rel_err(att.detach().numpy(), gold["att"]) <= OUT_TOL
        ((fh * fh).sum() + (att * att).sum()).backward()
        assert rel_err(f2.grad.numpy(), gold["df2"]) <= GRAD_TOL
    elif kind.startswith("attention_layer"):
        at = 1 if kind.endswith("1") else 2
        shapes = ({"att_layer.fc.weight": (1, Dm), "att_layer.fc.bias": (1,)} if at == 1 else
                  {"att_layer.fc1.weight": (Dm, Dm), "att_layer.fc2.weight": (1, Dm), "att_layer.fc2.bias": (1,)})
        sd = recipe_sd(shapes, case["salt"], requires_grad=True)
        a, b, att = O.attention_layer(sd, "", f1, f2, att_type=at)
        assert rel_err(a.detach().numpy(), gold["a"]) <= OUT_TOL
        assert rel_err(b.detach().numpy(), gold["b"]) <= OUT_TOL
        assert rel_err(att.detach().numpy(), gold["att"]) <= OUT_TOL
        ((b * b).sum() + (att * att).sum()).backward()
        assert rel_err(f2.grad.numpy(), gold["df2"]) <= GRAD_TOL
    else:
        sd = recipe_sd({"fc1.weight": (Dm, Dm), "fc1.bias": (Dm,), "fc2.weight": (Dm, Dm), "fc2.bias": (Dm,)},
                       case["salt"], requires_grad=True)
        o = O.nonlinear_layer(sd, "", f1)
        assert rel_err(o.detach().numpy(), gold["o"]) <= OUT_TOL
        (o * o).sum().backward()
    assert rel_err(f1.grad.numpy(), gold["df1"]) <= GRAD_TOL


def test_attention_1_is_separable():
    """the additive score's softmax over L does not depend on t."""
    case = ATT_MODULE_CASES[0]
    gold = load_golden("mod_attention_1")
    att = gold["att"]
    assert np.abs(att - att[:, :1, :]).max() <= 1e-6


def test_mhb_oracle_runs_and_is_normalised():
    """MHB: parity unpinned (reference class cannot execute); sanity only."""
    import types
    cfg = types.SimpleNamespace(q_vocab_size=50, a_vocab_size=30, emb_dim=24, hidden_dim=64,
                                img_feature_channel=96, img_feature_dim=196, model_name="mhb")
    sd = recipe_sd(O.mhb_shapes(cfg), 61)
    N, T = 4, 7
    img = torch.from_numpy(recipe.img_features(N, 196, 96, 61))
    qn = recipe.question_tokens(N, T, 50, 61)
    q = torch.from_numpy(qn)
    ql = torch.from_numpy(recipe.question_lengths(qn))
    out = O.mhb_forward(sd, cfg, img, q, ql)
    assert out.shape == (N, 30)
    assert torch.allclose(out.exp().sum(1), torch.ones(N), atol=1e-5)


@pytest.mark.parametrize("case", [pytest.param(c, id=c["name"]) for c in MFB_CASES])
def test_mfb_oracle_fp64_matches_reference_fp64(case):
    """fp64 oracle vs the reference modules run in fp64: exact restatement (1e-9 / 1e-8)."""
    gold = load_golden("mfb_" + case["name"])
    cfg, img, q, _, hard, _ = mfb_inputs(case)
    sd = {k: v.double().requires_grad_(True) for k, v in recipe_sd(O.mfb_shapes(cfg), case["salt"]).items()}
    logits = O.mfb_forward(sd, cfg, img.double(), q)
    assert rel_err(logits.detach().numpy(), gold["out64"]) <= 1e-9
    loss = O.ce_loss(logits, hard)
    assert abs(loss.item() - float(gold["loss64"])) <= 1e-9
    loss.backward()
    check_grads64({k: v.grad for k, v in sd.items()}, gold)


@pytest.mark.parametrize("case", [pytest.param(c, id=c["name"]) for c in MHBCOATT_CASES])
def test_mhbcoatt_oracle_fp64_matches_reference_fp64(case):
    gold = load_golden("mhbcoatt_" + case["name"])
    cfg, img, q, glove, _, soft = mfb_inputs(case)
    sd = {k: v.double().requires_grad_(True)
          for k, v in recipe_sd(O.mfb_shapes(cfg, mhb=True), case["salt"]).items()}
    out = O.mhbcoatt_forward(sd, cfg, img.double(), q, glove=None if glove is None else glove.double())
    assert rel_err(out.detach().numpy(), gold["out64"]) <= 1e-9
    loss = O.kldiv_loss(out, soft.double())
    assert abs(loss.item() - float(gold["loss64"])) <= 1e-9 * max(1.0, abs(float(gold["loss64"])))
    loss.backward()
    check_grads64({k: v.grad for k, v in sd.items()}, gold)


def _sym(name, shape, amp):
    return torch.from_numpy(recipe.sym_tensor(tuple(shape), amp, recipe.name_seed(name, 0)))


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_restatement_matches_torch_optim(wd):
    """solver.py:29,93: the optimizer is torch's; the oracle's restatement must reproduce it."""
    shapes = [(7,), (33, 5), (4, 3, 1, 1), (1025,)]
    ps = [torch.nn.Parameter(_sym("adam.p%d" % i, s, 0.5)) for i, s in enumerate(shapes)]
    mine = [p.detach().clone() for p in ps]
    state = [dict() for _ in ps]
    opt = torch.optim.Adam(ps, lr=7e-4, weight_decay=wd)
    for step in range(6):
        gs = [_sym("adam.g%d.%d" % (i, step), s, 0.1) for i, s in enumerate(shapes)]
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        lr = 7e-4 * (0.5 if step >= 3 else 1.0)          # solver.py:47-50 adjusts lr in place
        for grp in opt.param_groups:
            grp["lr"] = lr
        opt.step()
        O.adam_step(mine, gs, state, lr, weight_decay=wd)
    for p, m in zip(ps, mine):
        assert rel_err(m, p.detach()) < 1e-6


def test_loss_restatements_match_closed_form():
    """solver.py:26-28: mean CE over rows; KLDivLoss() default = mean over ALL N*A elements."""
    x = _sym("loss.x", (5, 11), 2.0)
    a = torch.tensor([0, 10, 3, 3, 7])
    lse = torch.logsumexp(x.double(), 1)
    want = (lse - x.double()[torch.arange(5), a]).mean()
    assert abs(O.ce_loss(x, a).double() - want) < 1e-6
    t = torch.softmax(_sym("loss.t", (5, 11), 1.0), 1)
    t[0, :4] = 0.0
    logp = torch.log_softmax(x, 1)
    want = torch.where(t > 0, t.double() * (t.double().log() - logp.double()), torch.zeros((), dtype=torch.float64)).sum() / 55
    assert abs(O.kldiv_loss(logp, t).double() - want) < 1e-7
