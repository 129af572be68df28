"""Input staging vs the reference loader's CPU transpose (data_loader.py:30-32).

fp32 staging is a permutation -> bit-exact.  bf16 staging = round-to-nearest-even of the same
values -> bit-exact against torch's fp32->bf16 conversion."""
import numpy as np
import pytest
import torch

import recipe

pytestmark = pytest.mark.gpu


def _loader_transpose(x):
    """data_loader.py:30-32 on one [D,14,14] array."""
    t = np.transpose(x, (1, 2, 0))
    return t.reshape(-1, t.shape[-1])


def _raw(n, D, hw, salt):
    return recipe.sym_tensor((n, D, hw, hw), 3.0, recipe.name_seed("rawfeat", salt))


@pytest.fixture(scope="module")
def dl():
    import vqa_amd
    return vqa_amd.data_loader


@pytest.mark.parametrize("n,D,hw", [(1, 2048, 14), (3, 2048, 14), (2, 96, 5), (2, 70, 3), (5, 64, 8), (1, 1, 1)])
def test_transpose_matches_loader_fp32(dl, n, D, hw):
    import vqa_amd
    raw = _raw(n, D, hw, n + D)
    want = np.stack([_loader_transpose(raw[i]) for i in range(n)])
    got = vqa_amd.ops.feat_transpose(torch.from_numpy(raw).reshape(n, D, hw * hw).cuda())
    assert got.shape == (n, hw * hw, D)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("n,D,hw", [(2, 2048, 14), (2, 70, 3)])
def test_transpose_bf16_is_rne_of_loader_output(dl, n, D, hw):
    import vqa_amd
    raw = _raw(n, D, hw, 7)
    raw[0, 0, 0, 0] = np.float32(1.00390625)          # exactly half-way between two bf16 values -> ties to even
    want = torch.from_numpy(np.stack([_loader_transpose(raw[i]) for i in range(n)])).to(torch.bfloat16)
    got = vqa_amd.ops.feat_transpose(torch.from_numpy(raw).reshape(n, D, hw * hw).cuda(), bf16=True)
    assert got.dtype == torch.bfloat16
    assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))


def test_stager_double_buffering_keeps_order_and_content(dl):
    n, D, hw = 4, 256, 14
    st = dl.FeatureStager(n, channels=D, regions=hw * hw, depth=2)
    batches = [_raw(n, D, hw, 100 + k) for k in range(5)]
    outs = []
    st.submit(batches[0])
    for k in range(5):
        if k + 1 < 5:
            st.submit(batches[k + 1])              # copy of k+1 queued while k is consumed
        x = st.next()
        outs.append((x * 1.0).cpu().numpy())       # some work on the compute stream
    for k in range(5):
        want = np.stack([_loader_transpose(batches[k][i]) for i in range(n)])
        assert np.array_equal(outs[k], want), k


def test_stager_partial_batch_list_input_and_errors(dl):
    from vqa_amd import VqfError
    D, hw = 64, 14
    st = dl.FeatureStager(8, channels=D, regions=hw * hw, depth=1)
    imgs = [_raw(1, D, hw, 200 + k)[0] for k in range(3)]    # last batch of an epoch: 3 of 8 rows
    st.submit(imgs)
    with pytest.raises(VqfError):
        st.host_slot()                                        # single slot still in flight
    x = st.next()
    assert x.shape == (3, hw * hw, D)
    assert np.array_equal(x.cpu().numpy(), np.stack([_loader_transpose(a) for a in imgs]))
    with pytest.raises(VqfError):
        st.next()
    with pytest.raises(VqfError):
        st.commit()


def test_staged_batch_feeds_the_model(dl):
    """raw [D,14,14] features -> stager -> MFB gives the same logits as the loader's CPU transpose."""
    import vqa_amd
    from cases import MFB_CASES
    from golden_util import recipe_sd, mfb_inputs
    from oracle import ref_torch as O
    case = [c for c in MFB_CASES if c.get("L", 196) == 196][0]
    cfg, img, q, _, _, _ = mfb_inputs(case)
    N, L, D = img.shape
    raw = np.ascontiguousarray(img.numpy().transpose(0, 2, 1)).reshape(N, D, 14, 14)   # what the .npy files hold
    m = vqa_amd.MFB(cfg).cuda().eval()
    m.load_state_dict(recipe_sd(O.mfb_shapes(cfg), case["salt"]))
    with torch.no_grad():
        a = m.forward(dl.stage_features(raw), q.cuda())
        b = m.forward(img.cuda(), q.cuda())
    assert torch.equal(a, b)
