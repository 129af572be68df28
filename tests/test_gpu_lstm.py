"""LSTM sequence kernels (MHBCoAtt's batch-axis recursion) vs torch.nn.LSTM in fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("S,B,I,H", [(40, 14, 300, 256), (7, 22, 48, 512), (3, 32, 16, 256), (5, 1, 8, 256),
                                     (1, 14, 300, 1024), (64, 14, 600, 1024)])
def test_lstm_seq_matches_torch_lstm(S, B, I, H):
    import vqa_amd
    vqa_amd.lib.load()
    fn = vqa_amd.functions.LstmSeqFn
    torch.manual_seed(S * 1000 + B)
    ref = torch.nn.LSTM(I, H, 1).double()
    x = (torch.rand(S, B, I, dtype=torch.float64) * 2 - 1).float().double().requires_grad_()
    for p in ref.parameters():
        p.data = p.data.float().double()
    out_ref, _ = ref(x)
    w = torch.linspace(-1, 1, out_ref.numel(), dtype=torch.float64).view_as(out_ref)
    (out_ref * w).sum().backward()
    xs = x.detach().float().cuda().requires_grad_()
    ps = [p.detach().float().cuda().requires_grad_() for p in
          (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)]
    hs = fn.apply(xs, *ps)
    assert _rel(hs, out_ref) <= 2e-5
    (hs * w.float().cuda()).sum().backward()
    assert _rel(xs.grad, x.grad) <= 1e-4
    for p, r in zip(ps, (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)):
        assert _rel(p.grad, r.grad) <= 1e-4


def test_unsupported_shapes_are_reported():
    import vqa_amd
    ops = vqa_amd.ops
    assert ops.lstm_seq_supported(14, 1024) and ops.lstm_seq_supported(32, 256)
    assert not ops.lstm_seq_supported(33, 1024) and not ops.lstm_seq_supported(14, 64)
    with pytest.raises(vqa_amd.VqfError):
        ops.lstm_seq_fwd(torch.zeros(2, 40, 4 * 256, device="cuda"), torch.zeros(4 * 256, 256, device="cuda"))


def test_mhbcoatt_hip_lstm_equals_miopen_lstm():
    """MHBCoAtt with the HIP sequence kernel vs the same module on nn.LSTM (full dims, N=6)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import vqa_amd, recipe
    from cases import MHBCOATT_CASES
    from golden_util import mfb_inputs
    case = dict(MHBCOATT_CASES[-1], N=6, salt=91)
    cfg, img, q, glove, hard, soft = mfb_inputs(case, "cuda")
    model = vqa_amd.MHBCoAtt(cfg)
    model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), case["salt"]))
                           for k, v in model.state_dict().items()})
    model = model.cuda().train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    res = {}
    for hip in (True, False):
        model.use_hip_lstm = hip
        model.zero_grad(set_to_none=True)
        out = model.forward(img, q)
        torch.nn.KLDivLoss()(out, soft).backward()
        res[hip] = (out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
    assert _rel(res[True][0], res[False][0]) <= 1e-5
    for k in res[True][1]:
        a, b = res[True][1][k], res[False][1][k]
        assert float((a - b).norm()) <= 2e-3 * float(b.norm()) + 1e-9, k
