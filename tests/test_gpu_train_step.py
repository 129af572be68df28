"""Solver tail on the HIP path: criteria + Adam vs the oracle / torch.optim.Adam.

Tolerances: losses 1e-5 relative (fp32 sums in a different order), loss gradients 1e-5,
Adam parameters after several steps 2e-6 relative (same op order as torch's single-tensor path,
no FMA contraction)."""
import numpy as np
import pytest
import torch

import recipe
from golden_util import rel_err
from oracle import ref_torch as O

pytestmark = pytest.mark.gpu


def _sym(name, shape, amp):
    return torch.from_numpy(recipe.sym_tensor(tuple(shape), amp, recipe.name_seed(name, 0)))


@pytest.fixture(scope="module")
def ts():
    import vqa_amd
    return vqa_amd.train_step


@pytest.mark.parametrize("N,A", [(1, 7), (5, 1000), (512, 1000), (33, 3001)])
def test_cross_entropy_matches_oracle(ts, N, A):
    x = _sym("ce.x.%d.%d" % (N, A), (N, A), 3.0)
    a = torch.from_numpy(recipe.hard_answers(N, A, N + A))
    xr = x.clone().requires_grad_(True)
    want = O.ce_loss(xr, a)
    want.backward()
    xg = x.cuda().requires_grad_(True)
    got = ts.CrossEntropyLoss()(xg, a.cuda())
    got.backward()
    assert got.shape == ()
    assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item())
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-5


def test_cross_entropy_ignore_index_and_scaling(ts):
    N, A = 9, 50
    x = _sym("ce.ign", (N, A), 2.0)
    a = torch.from_numpy(recipe.hard_answers(N, A, 3))
    a[2] = -100
    a[7] = -100
    xr = x.clone().requires_grad_(True)
    (O.ce_loss(xr, a) * 2.5).backward()
    xg = x.cuda().requires_grad_(True)
    loss = ts.CrossEntropyLoss()(xg, a.cuda())
    (loss * 2.5).backward()
    assert abs(loss.item() - O.ce_loss(x, a).item()) < 1e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-5
    assert float(xg.grad[2].abs().max()) == 0.0 and float(xg.grad[7].abs().max()) == 0.0


@pytest.mark.parametrize("N,A", [(1, 5), (6, 1000), (512, 1000), (17, 4099)])
def test_kldiv_matches_oracle(ts, N, A):
    logp = torch.log_softmax(_sym("kl.x.%d.%d" % (N, A), (N, A), 3.0), 1)
    t = torch.from_numpy(recipe.soft_answers(N, A, N * 7 + A))
    t[0, : min(3, A)] = 0.0                       # exact zeros contribute 0 (xlogy), not NaN
    lr = logp.clone().requires_grad_(True)
    want = O.kldiv_loss(lr, t)
    want.backward()
    lg = logp.cuda().requires_grad_(True)
    got = ts.KLDivLoss()(lg, t.cuda())
    got.backward()
    assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item()) + 1e-9
    assert rel_err(lg.grad.cpu(), lr.grad) < 1e-6


def test_criterion_for_follows_solver(ts):
    assert isinstance(ts.criterion_for("mhb_coAtt"), ts.KLDivLoss)
    assert isinstance(ts.criterion_for("mhb"), ts.KLDivLoss)
    assert isinstance(ts.criterion_for("mfb"), ts.CrossEntropyLoss)
    assert isinstance(ts.criterion_for("hieCoAtten"), ts.CrossEntropyLoss)


def _adam_case(shapes, steps, wd, ts, lr0=7e-4, via_views=False):
    ps_ref = [torch.nn.Parameter(_sym("adam.p%d" % i, s, 0.5)) for i, s in enumerate(shapes)]
    ps_gpu = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ps_ref]
    oracle_p = [p.detach().clone() for p in ps_ref]
    oracle_state = [dict() for _ in ps_ref]
    ref = torch.optim.Adam(ps_ref, lr=lr0, weight_decay=wd)
    opt = ts.Adam(ps_gpu, lr=lr0, weight_decay=wd)
    flat = None
    if via_views:          # gradients as views at odd offsets of one flat bucket (what the all-reducer hands over)
        flat = torch.zeros(sum(p.numel() for p in ps_gpu) + 3, device="cuda")
    for step in range(steps):
        gs = [_sym("adam.g%d.%d" % (i, step), s, 0.1) for i, s in enumerate(shapes)]
        lr = lr0 * (0.5 if step >= 3 else 1.0)
        for o in (ref, opt):
            for grp in o.param_groups:
                grp["lr"] = lr
        off = 3
        for pr, pg, g in zip(ps_ref, ps_gpu, gs):
            pr.grad = g.clone()
            if via_views:
                v = flat[off:off + g.numel()].view(g.shape)
                v.copy_(g)
                pg.grad = v
                off += g.numel()
            else:
                pg.grad = g.cuda()
        ref.step()
        opt.step()
        O.adam_step(oracle_p, gs, oracle_state, lr, weight_decay=wd)
    torch.cuda.synchronize()
    for pr, pg, po in zip(ps_ref, ps_gpu, oracle_p):
        assert rel_err(pg.detach().cpu(), pr.detach()) < 2e-6
        assert rel_err(pg.detach().cpu(), po) < 2e-6
    return ref, opt, ps_ref, ps_gpu


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_matches_torch_and_oracle(ts, wd):
    _adam_case([(7,), (33, 5), (4, 3, 1, 1), (4096,), (4097,), (3, 5000)], 6, wd, ts)


def test_adam_many_tensors_and_unaligned_views(ts):
    shapes = [(i % 7 + 1, 3 + i) for i in range(70)]           # > 2 launches of 32 tensors, ragged sizes
    _adam_case(shapes, 3, 0.0, ts, via_views=True)


def test_adam_state_dict_interchanges_with_torch(ts):
    shapes = [(10, 3), (5,)]
    ref, opt, ps_ref, ps_gpu = _adam_case(shapes, 2, 0.0, ts)
    sd = opt.state_dict()
    assert sorted(sd["state"][0].keys()) == ["exp_avg", "exp_avg_sq", "step"]
    # continue the run with torch's Adam from our state, and with ours from torch's state
    t_from_ours = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ps_gpu], lr=1e-3)
    t_from_ours.load_state_dict(sd)
    ours_from_t = ts.Adam([torch.nn.Parameter(p.detach().clone().cuda()) for p in ps_ref], lr=1e-3)
    ours_from_t.load_state_dict(ref.state_dict())
    gs = [_sym("adam.g2.%d" % i, s, 0.1) for i, s in enumerate(shapes)]
    for o in (t_from_ours, ours_from_t):
        for p, g in zip(o.param_groups[0]["params"], gs):
            p.grad = g.to(p.device)
        o.step()
    for a, b in zip(t_from_ours.param_groups[0]["params"], ours_from_t.param_groups[0]["params"]):
        assert rel_err(a.detach().cpu(), b.detach().cpu()) < 2e-6
    assert float(ours_from_t.state[ours_from_t.param_groups[0]["params"][0]]["step"]) == 3.0


def test_adam_skips_params_without_grad_and_rejects_cpu(ts):
    a = torch.nn.Parameter(torch.ones(8, device="cuda"))
    b = torch.nn.Parameter(torch.ones(8, device="cuda"))
    opt = ts.Adam([a, b], lr=0.1)
    a.grad = torch.ones(8, device="cuda")
    opt.step()
    assert float(b.detach().sum()) == 8.0 and float(a.detach()[0]) < 1.0
    c = torch.nn.Parameter(torch.ones(8))
    c.grad = torch.ones(8)
    from vqa_amd import VqfError
    with pytest.raises(VqfError):
        ts.Adam([c]).step()


def test_full_train_steps_track_torch_solver_tail(ts):
    """Three complete solver iterations (forward, criterion, backward, Adam) on the HIP MFB vs the
    same model driven by torch's CrossEntropyLoss + torch.optim.Adam: logits after the updates agree."""
    import vqa_amd
    from cases import MFB_CASES, make_cfg
    from golden_util import recipe_sd, mfb_inputs
    case = MFB_CASES[1]
    cfg, img, q, _, hard, _ = mfb_inputs(case)
    sd = recipe_sd(O.mfb_shapes(cfg), case["salt"])
    outs = []
    for mine in (True, False):
        m = vqa_amd.MFB(cfg).cuda()
        m.load_state_dict(sd)
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        crit = ts.CrossEntropyLoss() if mine else torch.nn.CrossEntropyLoss()
        opt = (ts.Adam if mine else torch.optim.Adam)(m.parameters(), lr=1e-4)
        for _ in range(3):
            loss = crit(m.forward(img.cuda(), q.cuda()), hard.cuda())
            opt.zero_grad()
            loss.backward()
            opt.step()
        with torch.no_grad():
            outs.append(m.forward(img.cuda(), q.cuda()).cpu())
    assert rel_err(outs[0], outs[1]) < 1e-4
