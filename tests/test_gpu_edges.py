"""Edge cases on the HIP path: single sample / single token / single region, all-padding questions,
the S <= 1024 limit, non-contiguous inputs, and loud errors instead of silent fallbacks."""
import types

import pytest
import torch

import recipe
from golden_util import recipe_sd, rel_err
from oracle import ref_torch as O

pytestmark = pytest.mark.gpu


def _vqa():
    import vqa_amd
    vqa_amd.lib.load()
    return vqa_amd


def _cfg(model_name, H=64, E=24, D=96, L=20, V=50, A=30):
    return types.SimpleNamespace(q_vocab_size=V, a_vocab_size=A, emb_dim=E, hidden_dim=H, num_layers=1,
                                 model_name=model_name, glove=False, img_feature_channel=D, img_feature_dim=L)


def _load(model, salt):
    model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), salt))
                           for k, v in model.state_dict().items()})
    model = model.cuda().train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


@pytest.mark.parametrize("N,T,L", [(1, 1, 1), (1, 7, 20), (2, 1, 196), (3, 22, 1), (2, 5, 1024)])
@pytest.mark.parametrize("mhb", [False, True])
def test_degenerate_shapes_match_oracle(mhb, N, T, L):
    vqa = _vqa()
    cfg = _cfg("mhb_coAtt" if mhb else "mfb", L=L)
    model = _load((vqa.MHBCoAtt if mhb else vqa.MFB)(cfg), 101)
    img = torch.from_numpy(recipe.img_features(N, L, 96, 101))
    q = torch.from_numpy(recipe.question_tokens(N, T, 50, 101))
    out = model.forward(img.cuda(), q.cuda())
    out.sum().backward()
    sd = recipe_sd(O.mfb_shapes(cfg, mhb=mhb), 101)
    ref = (O.mhbcoatt_forward if mhb else O.mfb_forward)(sd, cfg, img, q)
    assert out.shape == ref.shape
    assert rel_err(out.detach().cpu().numpy(), ref.numpy()) <= 1e-4
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_all_padding_questions_and_zero_images():
    """token 0 everywhere (utils.py:185 pad id) and an all-zero image grid: finite, equal to the oracle."""
    vqa = _vqa()
    cfg = _cfg("mfb")
    model = _load(vqa.MFB(cfg), 102)
    img = torch.zeros(3, 20, 96)
    img[1] = torch.from_numpy(recipe.img_features(1, 20, 96, 102))[0]
    q = torch.zeros(3, 7, dtype=torch.long)
    out = model.forward(img.cuda(), q.cuda())
    out.sum().backward()
    ref = O.mfb_forward(recipe_sd(O.mfb_shapes(cfg), 102), cfg, img, q)
    assert torch.isfinite(out).all()
    assert rel_err(out.detach().cpu().numpy(), ref.numpy()) <= 1e-4
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_non_contiguous_inputs_are_accepted():
    vqa = _vqa()
    cfg = _cfg("mfb")
    model = _load(vqa.MFB(cfg), 103)
    base = torch.from_numpy(recipe.img_features(2, 96, 20, 103)).cuda()          # (N, D, L) storage
    img = base.permute(0, 2, 1)                                                 # (N, L, D) strided view
    q = torch.from_numpy(recipe.question_tokens(2, 7, 50, 103)).cuda()
    out = model.forward(img, q)
    ref = model.forward(img.contiguous(), q)
    assert torch.equal(out, ref)


def test_limits_and_misuse_fail_loudly():
    vqa = _vqa()
    cfg = _cfg("mfb", L=1025)                                   # softmax rows are staged in LDS: S <= 1024
    model = _load(vqa.MFB(cfg), 104)
    with pytest.raises(vqa.VqfError):
        model.forward(torch.zeros(1, 1025, 96, device="cuda"), torch.ones(1, 3, dtype=torch.long, device="cuda"))
    cfg = _cfg("mfb")
    model = _load(vqa.MFB(cfg), 104)
    with pytest.raises(vqa.VqfError):                           # the image is data on this path
        model.forward(torch.zeros(1, 20, 96, device="cuda", requires_grad=True),
                      torch.ones(1, 3, dtype=torch.long, device="cuda"))
    with pytest.raises(vqa.VqfError):                           # fp64 tensors are refused, not down-cast
        model.double().forward(torch.zeros(1, 20, 96, device="cuda", dtype=torch.float64),
                               torch.ones(1, 3, dtype=torch.long, device="cuda"))


def test_hiecoatten_single_sample_squeeze_semantics():
    """N = 1: the reference's torch.squeeze drops the batch axis of av/aq (hieCoAtten.py:43,50)."""
    vqa = _vqa()
    model = vqa.HieCoAtten(block_num=20, word_num=7, img_size=96, vocab_size=50, embed_size=64, output_size=30)
    model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), 105))
                           for k, v in model.state_dict().items()})
    model = model.cuda()
    model.drop_p = 0.0
    img = torch.from_numpy(recipe.img_features(1, 20, 96, 105))
    q = torch.from_numpy(recipe.question_tokens(1, 7, 50, 105))
    x, av, aq = model.forward(img.cuda(), q.cuda())
    assert x.shape == (1, 30) and av.shape == (20,) and aq.shape == (7,)
    ox, oav, oaq = O.hiecoatten_forward(recipe_sd(O.hiecoatten_shapes(96, 50, 64, 30), 105), img, q)
    assert rel_err(x.detach().cpu().numpy(), ox.numpy()) <= 1e-4
    assert rel_err(av.detach().cpu().numpy(), oav.numpy().reshape(-1)) <= 1e-4


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process (nn.DataParallel-style use)")
def test_gemm_kernels_on_a_second_device_of_the_same_process():
    """The > 64 KB dynamic-LDS attribute is per device (common.h::vqf_set_dyn_lds): every GEMM family must launch
    on device 1 after it has been used on device 0 (the reference drives several GPUs from one process,
    solver.py:34-36)."""
    import vqa_amd
    ops = vqa_amd.ops
    for dev in (0, 1):
        with torch.cuda.device(dev):
            d = torch.device("cuda", dev)
            g = torch.Generator().manual_seed(5)
            A = torch.rand((1024, 256), generator=g).to(d)
            B = torch.rand((512, 256), generator=g).to(d)
            ref = A.double() @ B.double().t()
            out = ops.gemm(A, B)                                             # 128x128 kernel, 72 KB LDS
            assert float((out.double() - ref).abs().max() / ref.abs().max()) <= 1e-5
            ob = ops.gemm_bf16(A.to(torch.bfloat16), B.to(torch.bfloat16))
            assert float((ob.double() - ref).abs().max() / ref.abs().max()) <= 2e-2
            A2 = torch.rand((8192, 64), generator=g).to(d)
            B2 = torch.rand((8192, 64), generator=g).to(d)                   # 1024 tiles: the large-tile kernels
            r2 = A2.double() @ B2.double().t()
            assert float((ops.gemm(A2, B2).double() - r2).abs().max() / r2.abs().max()) <= 1e-5
            o2 = ops.gemm_bf16(A2.to(torch.bfloat16), B2.to(torch.bfloat16))
            assert float((o2.double() - r2).abs().max() / r2.abs().max()) <= 2e-2
