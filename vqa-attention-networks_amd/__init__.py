"""MI355X-native attention-fusion path of klory/vqa-attention-networks.

Import with importlib (the directory name carries a hyphen) or through the
alias module `vqa_amd` at the repo root:

    import vqa_amd
    model = vqa_amd.MFB(cfg).cuda()

The classes keep the reference's names, constructor arguments, forward()
signatures and state_dict keys; the fusion arithmetic runs in hand-written HIP
kernels (csrc/) behind the C ABI of include/vqa_fusion.h.
"""
from .host import lib
from .host.lib import build, VqfError


def __getattr__(name):
    # lazy: importing torch-backed modules only when asked for
    import importlib
    table = {
        "MFB": ".host.mfb", "MHBCoAtt": ".host.mhb_coAtt", "MHB": ".host.mhb_coAtt",
        "HieCoAtten": ".host.hieCoAtten",
        "Attention_layer": ".host.modules", "Attention_1": ".host.modules",
        "Attention_2": ".host.modules", "Nonlinear_layer": ".host.modules",
        "AttentionNet": ".host.networks", "iBOWIMG": ".host.networks",
        "ops": ".host.ops", "functions": ".host.functions", "parallel": ".host.parallel",
        "train_step": ".host.train_step", "CrossEntropyLoss": ".host.train_step",
        "KLDivLoss": ".host.train_step", "Adam": ".host.train_step",
        "data_loader": ".host.data_loader", "FeatureStager": ".host.data_loader",
    }
    if name in table:
        mod = importlib.import_module(table[name], __name__)
        return mod if name in ("ops", "functions", "parallel", "train_step", "data_loader") else getattr(mod, name)
    raise AttributeError(name)
