"""The driver's build entry point and the drop-in shims, on CPU.

build() is what the driver calls every round ("does it build") and what lib.load() tells a user to run on a
fresh checkout; the shims in vqa-attention-networks_amd/dropin/ are what a maintainer puts in front of the
reference's own modules on sys.path, so the names and signatures train_models.py:8-9,44-52, solver.py and
networks.py:4 rely on are checked here."""
import inspect
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "vqa-attention-networks_amd", "dropin")


def test_graft_entry_build_runs():
    import __graft_entry__ as g
    g.build()                                   # raises on a compile error, a missing symbol or an ABI mismatch
    import vqa_amd
    assert os.path.exists(vqa_amd.lib.LIB_PATH)
    assert vqa_amd.lib.load().vqf_abi_version() == vqa_amd.lib.ABI_VERSION


def test_graft_entry_build_from_a_fresh_interpreter():
    """`python -c 'import __graft_entry__ as g; g.build()'` is the command lib.load() prints."""
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "built" in r.stdout and "gfx950" in r.stdout


_PROBE = r"""
import inspect, sys
sys.path.insert(0, %r)
import mfb, mhb_coAtt, hieCoAtten, modules, networks
from mhb_coAtt import MHBCoAtt, MHB            # train_models.py:9
from hieCoAtten import HieCoAtten              # train_models.py:8
from modules import Attention_layer            # networks.py:4
assert mfb.__file__.startswith(%r), mfb.__file__
def params(f):
    return [p for p in inspect.signature(f).parameters if p != "self"]
assert params(mfb.MFB.__init__) == ["cfg"]                                              # mfb.py:7
assert params(MHBCoAtt.__init__) == ["cfg"] and params(MHB.__init__) == ["cfg"]         # mhb_coAtt.py:7,154
assert params(mfb.MFB.forward) == ["img_features", "questions", "is_training"]          # mfb.py:61
assert params(MHBCoAtt.forward) == ["img_features", "questions", "glove_matrix", "is_training"]   # mhb_coAtt.py:61
assert params(MHB.forward) == ["img_feature", "questions", "q_length"]                  # mhb_coAtt.py:174
assert params(HieCoAtten.__init__) == ["block_num", "word_num", "img_size", "vocab_size", "embed_size",
                                       "att_num", "output_size"]                         # hieCoAtten.py:6
assert params(HieCoAtten.forward) == ["img_features", "que_features"]                   # hieCoAtten.py:18
assert params(networks.AttentionNet.__init__) == params(HieCoAtten.__init__)            # networks.py:31
assert params(networks.AttentionNet.forward) == ["img_features", "que_features"]        # networks.py:47
assert params(networks.iBOWIMG.__init__) == ["img_size", "vocab_size", "embed_size", "output_size"]   # networks.py:8
assert params(Attention_layer.__init__) == ["feature_size", "att_type"]                 # modules.py:9
assert params(Attention_layer.forward) == ["feature_1", "feature_2"]                    # modules.py:26
assert params(modules.Attention_1.__init__) == ["feature_size"] == params(modules.Attention_2.__init__)
assert params(modules.Nonlinear_layer.__init__) == ["f_size"]                           # modules.py:98
d = inspect.signature(HieCoAtten.__init__).parameters
assert (d["block_num"].default, d["word_num"].default, d["img_size"].default, d["vocab_size"].default,
        d["embed_size"].default, d["att_num"].default, d["output_size"].default) == (196, 22, 1024, 15881, 512, 6, 3000)
import types, torch
cfg = types.SimpleNamespace(q_vocab_size=50, a_vocab_size=30, emb_dim=24, hidden_dim=64, num_layers=1,
                            model_name="mfb", glove=False, img_feature_channel=96, img_feature_dim=196)
for cls in (mfb.MFB, MHBCoAtt, MHB):
    m = cls(cfg)                               # train_models.py:44-47 call shape
    assert isinstance(m, torch.nn.Module)
    for name, p in m.named_parameters():       # train_models.py:54-56
        if name.find("bias") == -1:
            torch.nn.init.xavier_uniform_(p)
assert isinstance(HieCoAtten(block_num=20, word_num=7, img_size=96, vocab_size=50, embed_size=64, output_size=30),
                  torch.nn.Module)
print("dropin ok")
"""


def test_dropin_shims_import_first_on_sys_path():
    r = subprocess.run([sys.executable, "-c", _PROBE % (DROPIN, DROPIN)], cwd="/tmp",
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dropin ok" in r.stdout
