"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports
every symbol include/vqa_fusion.h declares; host modules keep the reference's
constructor signatures and state_dict layout; the product path refuses CPU tensors."""
import os
import re
import types

import pytest
import torch

from cases import MFB_CASES, MHBCOATT_CASES, make_cfg
from oracle import ref_torch as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vqa():
    import vqa_amd
    vqa_amd.build()
    return vqa_amd


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "vqa_fusion.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vqf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(vqa):
    import ctypes
    lib = ctypes.CDLL(vqa.lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libvqa_fusion.so does not export %s" % s


def test_binding_table_covers_the_header(vqa):
    assert sorted(vqa.lib.SIGNATURES.keys()) == _header_symbols()
    lib = vqa.lib.load()
    assert lib.vqf_abi_version() == vqa.lib.ABI_VERSION == 7
    assert b"gfx950" in lib.vqf_build_info()
    assert lib.vqf_prof_num_kernels() > 10
    names = [lib.vqf_prof_kernel_name(i) for i in range(lib.vqf_prof_num_kernels())]
    assert all(n for n in names)


def test_library_options_are_explicit_and_restorable(vqa):
    """vqf_set_option / vqf_get_option (include/vqa_fusion.h): launch policy is a cached library option, not an
    environment lookup on the launch path; the option ids of the header and of ops.OPTIONS agree."""
    ops = vqa.ops
    hdr = open(os.path.join(ROOT, "include", "vqa_fusion.h")).read()
    ids = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"#define VQF_OPT_([A-Z0-9_]+) (\d+)", hdr)}
    count = ids.pop("count")
    assert ids == ops.OPTIONS and sorted(ids.values()) == list(range(count))
    # the environment variable read for an option at load time is "VQF_" + the name of its constant (two of them
    # used to be spelt differently, so `VQF_GEMM_F32_LOOP=1 python bench.py` was silently ignored)
    lib0 = vqa.lib.load()
    for name, i in ids.items():
        assert lib0.vqf_option_env_name(i).decode() == "VQF_" + name.upper()
    assert lib0.vqf_option_env_name(count) == b""
    before = {k: ops.get_option(k) for k in ops.OPTIONS}
    env = dict(os.environ)
    with ops.options(gemm_f32_persist=0, gemm_cu_limit=240):
        assert ops.get_option("gemm_f32_persist") == 0 and ops.get_option("gemm_cu_limit") == 240
        assert ops.set_option("gemm_f32_persist", 1) == 0            # returns what it replaced
    assert {k: ops.get_option(k) for k in ops.OPTIONS} == before and dict(os.environ) == env
    assert ops.set_option("fuse_ls", -7) == before["fuse_ls"] and ops.get_option("fuse_ls") == -1   # negative = default
    ops.set_option("fuse_ls", before["fuse_ls"])
    import ctypes
    lib = vqa.lib.load()
    assert lib.vqf_set_option(count, 1, None) == -1 and lib.vqf_get_option(-1, ctypes.byref(ctypes.c_int(0))) == -1
    assert lib.vqf_get_option(0, None) == -1
    # no getenv on any launch path: the only call is the load-time initialiser in prof.hip
    csrc = os.path.join(ROOT, "vqa-attention-networks_amd", "csrc")
    hits = [f for f in os.listdir(csrc) if f.endswith((".hip", ".h")) and "getenv" in open(os.path.join(csrc, f)).read()]
    assert hits == ["prof.hip"]


def test_workspace_size_queries_need_no_gpu(vqa):
    lib = vqa.lib.load()
    assert lib.vqf_colsum_ws_bytes(1000, 512) == (63 + 32) * 512 * 4       # 16 rows per workgroup on a short tensor (reduce.hip cs_rows_vec)
    assert lib.vqf_colsum_ws_bytes(100352, 512) == (1568 + 32) * 512 * 4   # 64 on a tall one
    assert lib.vqf_mfb_fuse_bwd_ws_bytes(512, 196, 1000) == (2 * 512 * 4 + 32) * 5000 * 4
    assert lib.vqf_att_logits_bwd_ws_bytes(100352, 1024) > 0
    assert lib.vqf_colsum_ws_bytes(0, 5) == 0


def test_state_dict_layout_matches_reference(vqa):
    cfg = make_cfg(MFB_CASES[0])
    assert {k: tuple(v.shape) for k, v in vqa.MFB(cfg).state_dict().items()} == \
        {k: tuple(v) for k, v in O.mfb_shapes(cfg).items()}
    cfgm = make_cfg(MFB_CASES[4])
    assert {k: tuple(v.shape) for k, v in vqa.MFB(cfgm).state_dict().items()} == \
        {k: tuple(v) for k, v in O.mfb_shapes(cfgm).items()}
    cfgh = make_cfg(MHBCOATT_CASES[3])
    assert {k: tuple(v.shape) for k, v in vqa.MHBCoAtt(cfgh).state_dict().items()} == \
        {k: tuple(v) for k, v in O.mfb_shapes(cfgh, mhb=True).items()}
    cfgb = types.SimpleNamespace(q_vocab_size=50, a_vocab_size=30, emb_dim=24, hidden_dim=64,
                                 img_feature_channel=96, img_feature_dim=196, model_name="mhb")
    assert {k: tuple(v.shape) for k, v in vqa.MHB(cfgb).state_dict().items()} == \
        {k: tuple(v) for k, v in O.mhb_shapes(cfgb).items()}


def test_product_path_has_no_cpu_fallback(vqa):
    cfg = make_cfg(MFB_CASES[0])
    m = vqa.MFB(cfg)
    with pytest.raises(vqa.VqfError):
        m(torch.zeros(2, cfg.img_feature_dim, cfg.img_feature_channel), torch.ones(2, 7, dtype=torch.long))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vqa-attention-networks_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(dp, f)
                assert "/root/reference" not in src


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout only exists in the build container")
def test_checkpoints_interchange_with_the_reference_classes(vqa):
    """solver.save() / train_models.py:58-60: a state_dict written by either implementation loads
    strictly (same keys, same shapes) into the other."""
    import importlib.util
    import sys
    sys.dont_write_bytecode = True

    def ref_module(name):
        spec = importlib.util.spec_from_file_location("_ref_" + name, "/root/reference/%s.py" % name)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    cfg = make_cfg(MFB_CASES[0])
    ours, theirs = vqa.MFB(cfg), ref_module("mfb").MFB(cfg)
    theirs.load_state_dict(ours.state_dict(), strict=True)
    ours.load_state_dict(theirs.state_dict(), strict=True)
    cfgh = make_cfg(MHBCOATT_CASES[3])
    mh = ref_module("mhb_coAtt")
    ours, theirs = vqa.MHBCoAtt(cfgh), mh.MHBCoAtt(cfgh)
    theirs.load_state_dict(ours.state_dict(), strict=True)
    ours.load_state_dict(theirs.state_dict(), strict=True)
    cfgb = types.SimpleNamespace(q_vocab_size=50, a_vocab_size=30, emb_dim=24, hidden_dim=64,
                                 img_feature_channel=96, img_feature_dim=196, model_name="mhb")
    vqa.MHB(cfgb).load_state_dict(mh.MHB(cfgb).state_dict(), strict=True)
    rh = ref_module("hieCoAtten")
    kw = dict(block_num=20, word_num=7, img_size=96, vocab_size=50, embed_size=64, output_size=30)
    vqa.HieCoAtten(**kw).load_state_dict(rh.HieCoAtten(**kw).state_dict(), strict=True)
    sys.path.insert(0, "/root/reference")           # networks.py does `from modules import Attention_layer`
    try:
        rn = ref_module("networks")
    finally:
        sys.path.remove("/root/reference")
        sys.modules.pop("modules", None)
    kwn = dict(block_num=20, word_num=7, img_size=96, vocab_size=50, embed_size=64, att_num=6, output_size=30)
    vqa.AttentionNet(**kwn).load_state_dict(rn.AttentionNet(**kwn).state_dict(), strict=True)
    vqa.iBOWIMG(96, 50, 64, 30).load_state_dict(rn.iBOWIMG(96, 50, 64, 30).state_dict(), strict=True)
