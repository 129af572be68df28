"""Case table for the golden vectors (inputs/weights come from recipe.py)."""
import types
import zlib
import numpy as np

_SMALL = dict(H=64, E=24, D=96, L=20, V=50, A=30, T=7)
_FULL = dict(H=1024, E=300, D=2048, L=196, V=1000, A=1000, T=14)


def _c(name, salt, N, model_name="mfb", glove=False, **kw):
    d = dict(_SMALL)
    d.update(kw)
    d.update(name=name, salt=salt, N=N, model_name=model_name, glove=glove)
    return d


MFB_CASES = [
    _c("small_n1", 1, 1),
    _c("small_n2", 2, 2),
    _c("small_n3", 3, 3),
    _c("small_n5", 4, 5),
    _c("small_multilayer_n3", 5, 3, model_name="mfb-multilayer"),
    _c("small_t22_l196_n2", 6, 2, T=22, L=196),
    _c("full_n2", 7, 2, **_FULL),
    _c("full_multilayer_n1", 8, 1, model_name="mfb-multilayer", **_FULL),
]

MHBCOATT_CASES = [
    _c("small_n1", 11, 1, model_name="mhb_coAtt"),
    _c("small_n3", 12, 3, model_name="mhb_coAtt"),
    _c("small_n5", 13, 5, model_name="mhb_coAtt"),
    _c("small_glove_n3", 14, 3, model_name="mhb_coAtt", glove=True),
    _c("small_t22_l196_n2", 15, 2, model_name="mhb_coAtt", T=22, L=196),
    _c("full_n2", 16, 2, model_name="mhb_coAtt", **_FULL),
]

# MHB (mhb_coAtt.py:153-217): the class views the grid as (N, 14, 14, C) (:178), so L = 196 in every case; N = 1 cannot run in
# the reference (torch.squeeze at :195 drops the batch axis and F.normalize(dim=1) raises), hence N >= 2
MHB_CASES = [
    _c("small_n2", 62, 2, model_name="mhb", L=196),
    _c("small_n3", 63, 3, model_name="mhb", L=196),
    _c("small_n5", 64, 5, model_name="mhb", L=196, T=9),
    _c("full_n4", 65, 4, model_name="mhb", **_FULL),
]

HIE_CASES = [
    dict(name="small_n2", salt=21, N=2, L=20, T=7, img_size=96, V=50, E=64, A=30),
    dict(name="small_n3", salt=22, N=3, L=20, T=7, img_size=96, V=50, E=64, A=30),
    dict(name="small_n5_l196", salt=23, N=5, L=196, T=22, img_size=96, V=50, E=64, A=30),
    dict(name="full_n4", salt=24, N=4, L=196, T=14, img_size=2048, V=1000, E=512, A=1000),
]

ATTNET_CASES = [
    dict(name="small_n3", salt=31, N=3, L=20, T=7, img_size=96, V=50, E=64, A=30, att_num=6),
    dict(name="small_n4_att2", salt=32, N=4, L=12, T=5, img_size=48, V=50, E=32, A=10, att_num=2),
    # the shapes the reference trains with (networks.py:31 defaults except img_size / vocab / output scaled to the synthetic
    # configs): 196 regions, 14 words, embed 512, 6 attention layers
    dict(name="full_n4", salt=33, N=4, L=196, T=14, img_size=2048, V=1000, E=512, A=1000, att_num=6),
]

IBOW_CASES = [
    dict(name="small_n4", salt=41, N=4, T=7, img_size=96, V=50, E=64, A=30),
]

ATT_MODULE_CASES = [
    dict(name="attention_1", kind="attention_1", salt=51, N=3, L=20, T=7, D=64),
    dict(name="attention_2", kind="attention_2", salt=52, N=3, L=20, T=7, D=64),
    dict(name="attention_layer1", kind="attention_layer1", salt=53, N=2, L=20, T=7, D=64),
    dict(name="attention_layer2", kind="attention_layer2", salt=54, N=2, L=20, T=7, D=64),
    dict(name="nonlinear", kind="nonlinear", salt=55, N=2, L=20, T=7, D=64),
]


def make_cfg(case):
    """Attribute bag with the fields the hot-path constructors read."""
    return types.SimpleNamespace(
        q_vocab_size=case["V"], a_vocab_size=case["A"], emb_dim=case["E"],
        hidden_dim=case["H"], num_layers=1, model_name=case["model_name"],
        glove=case["glove"], img_feature_channel=case["D"], img_feature_dim=case["L"],
        batch_size=case["N"])


def sample_indices(name, numel, k):
    """k deterministic indices into a flat tensor of numel entries."""
    seed = zlib.crc32(("idx:" + name).encode()) & 0xFFFFFFFF
    i = np.arange(k, dtype=np.uint64)
    h = (i * np.uint64(0x9E3779B1) + np.uint64(seed)) * np.uint64(0x85EBCA6B)
    h ^= h >> np.uint64(13)
    return (h % np.uint64(max(numel, 1))).astype(np.int64)
