#!/usr/bin/env python3
"""Capture golden vectors from the IMPORTED reference (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (/root/reference) never travels: this script is the only place
that touches it, and only its OUTPUTS are committed (small .npz fixtures).
Inputs and weights are regenerated everywhere from tests/golden/recipe.py.

Harness shims, none of which alter arithmetic:
  1. sys.path + dont_write_bytecode (reference dir is read-only);
  2. Tensor.view falls back to .reshape on the stride RuntimeError that
     mfb.py:105 / mhb_coAtt.py:107 hit on torch >= 1.x;
  3. cfg is a SimpleNamespace (easydict is not installed);
  4. torch.nn.functional.dropout -> identity while HieCoAtten / AttentionNet /
     iBOWIMG run (their functional dropout is always on, hieCoAtten.py:26...);
  5. .eval() for the nn.Dropout models (MFB, MHBCoAtt, MHB); autograd still works.
  6. class MHB (mhb_coAtt.py:153-217) cannot execute as shipped: :176 moves a fresh tensor to the GPU with a hard
     `.cuda()` and :214 reads a name that does not exist.  `load_mhb_class()` parses the reference file with `ast`
     and applies EXACTLY these two edits to MHB.forward before compiling it (nothing is written anywhere):

         176: -    lstm_out = torch.zeros((batch_size, self.cfg.hidden_dim), dtype=torch.float).cuda()
              +    lstm_out = torch.zeros((batch_size, self.cfg.hidden_dim), dtype=torch.float).to(img_feature.device)
         214: -    logits = self.linear_out(mhb_22)
              +    logits = self.linear_out(mhb_12)

     For the float64 digests (`out64`, `g64*`) one more edit is needed, in that pass only: `dtype=torch.float` at
     :176 would truncate the LSTM states of a model.double() run to fp32 (and then fail in the fp64 Linear), so the
     fp64 class reads `dtype=torch.double` there.  The fp32 goldens come from the two-edit class.
"""
import os
import sys
import types
import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = os.environ.get("VQA_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import recipe  # noqa: E402
from cases import (MFB_CASES, MHBCOATT_CASES, MHB_CASES, HIE_CASES, ATTNET_CASES, IBOW_CASES,  # noqa: E402
                   ATT_MODULE_CASES, make_cfg, sample_indices)

torch.set_num_threads(8)

_orig_view = torch.Tensor.view


def _view_or_reshape(self, *shape, **kw):
    try:
        return _orig_view(self, *shape, **kw)
    except RuntimeError:
        return self.reshape(*shape)


torch.Tensor.view = _view_or_reshape

import mfb as ref_mfb  # noqa: E402
import mhb_coAtt as ref_mhb  # noqa: E402
import hieCoAtten as ref_hie  # noqa: E402
import networks as ref_net  # noqa: E402
import modules as ref_mod  # noqa: E402


def load_mhb_class(fp64=False):
    """class MHB of the reference with the two edits of shim 6 (three for fp64=True), compiled from its own text."""
    import ast
    path = os.path.join(REF, "mhb_coAtt.py")
    tree = ast.parse(open(path).read(), filename=path)
    counts = {"cuda": 0, "mhb_22": 0, "float": 0}

    class Fix(ast.NodeTransformer):
        def visit_Call(self, node):
            self.generic_visit(node)
            if isinstance(node.func, ast.Attribute) and node.func.attr == "cuda" and not node.args and not node.keywords:
                counts["cuda"] += 1
                dev = ast.Attribute(value=ast.Name(id="img_feature", ctx=ast.Load()), attr="device", ctx=ast.Load())
                return ast.copy_location(ast.Call(func=ast.Attribute(value=node.func.value, attr="to", ctx=ast.Load()),
                                                  args=[dev], keywords=[]), node)
            return node

        def visit_Name(self, node):
            if node.id == "mhb_22":
                counts["mhb_22"] += 1
                return ast.copy_location(ast.Name(id="mhb_12", ctx=node.ctx), node)
            return node

        def visit_Attribute(self, node):
            self.generic_visit(node)
            if fp64 and node.attr == "float" and isinstance(node.value, ast.Name) and node.value.id == "torch":
                counts["float"] += 1
                return ast.copy_location(ast.Attribute(value=node.value, attr="double", ctx=node.ctx), node)
            return node

    for cls in tree.body:
        if isinstance(cls, ast.ClassDef) and cls.name == "MHB":
            for fn in cls.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == "forward":
                    Fix().visit(fn)
    assert counts == {"cuda": 1, "mhb_22": 1, "float": 1 if fp64 else 0}, counts
    ast.fix_missing_locations(tree)
    ns = {"__name__": "_ref_mhb_fixed"}
    exec(compile(tree, path, "exec"), ns)
    return ns["MHB"]


def run_mhb(case):
    cfg = make_cfg(case)
    N, T = case["N"], case["T"]
    img = torch.from_numpy(recipe.img_features(N, cfg.img_feature_dim, cfg.img_feature_channel, case["salt"]))
    qn = recipe.question_tokens(N, T, cfg.q_vocab_size, case["salt"])
    q, ql = torch.from_numpy(qn), torch.from_numpy(recipe.question_lengths(qn))
    soft = torch.from_numpy(recipe.soft_answers(N, cfg.a_vocab_size, case["salt"]))
    out = {}
    for fp64 in (False, True):
        model = load_mhb_class(fp64)(cfg)
        load_recipe(model, case["salt"])
        model.eval()
        if fp64:
            model.double()
        store = {}
        hs = hook_io(model, ["linear_q_1", "linear_i_1", "linear_out"], store)
        outp = model.forward(img.double() if fp64 else img, q, ql)
        loss = torch.nn.KLDivLoss()(outp, soft.double() if fp64 else soft)          # solver.py:26-27 (mhb -> KLDiv)
        loss.backward()
        for h in hs:
            h.remove()
        if fp64:
            out["out64"], out["loss64"] = outp.detach().numpy(), np.array(loss.item())
            grad_digest(model, out, tag="64")
        else:
            out["out"], out["loss"] = outp.detach().numpy(), np.array(loss.item())
            out["q_length"] = ql.numpy()
            tensor_digest("lstm_out", store["linear_q_1.in"], out)
            tensor_digest("i_mean_pooled", store["linear_i_1.in"], out)
            tensor_digest("mhb_12", store["linear_out.in"], out)
            grad_digest(model, out)
    return out


class _no_functional_dropout:
    def __enter__(self):
        self._d = F.dropout
        F.dropout = lambda x, *a, **k: x
        torch.nn.functional.dropout = F.dropout

    def __exit__(self, *a):
        F.dropout = self._d
        torch.nn.functional.dropout = self._d


def load_recipe(model, salt):
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            new[k] = v
        elif k.endswith("running_mean"):
            new[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            new[k] = torch.ones_like(v)
        else:
            new[k] = torch.from_numpy(recipe.weight_for(k, tuple(v.shape), salt))
    model.load_state_dict(new)


def grad_digest(model, out, tag=""):
    """per-parameter: is-None flag, l2 norm, abs-sum, 16 sampled entries.
    tag="64": digests of the reference run in float64 (model.double())."""
    for k, p in model.named_parameters():
        if p.grad is None:
            out["gnone/" + k] = np.array(1, dtype=np.int64)
            continue
        g = p.grad.detach().reshape(-1).double().numpy()
        out["g%snorm/" % tag + k] = np.array(np.sqrt((g * g).sum()))
        out["g%sabs/" % tag + k] = np.array(np.abs(g).sum())
        idx = sample_indices(k, g.size, 16)
        out["g%ssamp/" % tag + k] = g[idx].astype(np.float64 if tag else np.float32)


def tensor_digest(prefix, t, out, full_limit=4096):
    a = t.detach().reshape(-1).double().numpy()
    out[prefix + "/sum"] = np.array(a.sum())
    out[prefix + "/abs"] = np.array(np.abs(a).sum())
    if a.size <= full_limit:
        out[prefix + "/full"] = t.detach().numpy().astype(np.float32)
    else:
        idx = sample_indices(prefix, a.size, 64)
        out[prefix + "/samp"] = a[idx].astype(np.float32)


def hook_io(model, names, store):
    hs = []
    for n in names:
        m = dict(model.named_modules())[n]

        def fn(mod, inp, outp, n=n):
            store[n + ".in"] = inp[0].detach()
            store[n + ".out"] = outp.detach()
        hs.append(m.register_forward_hook(fn))
    return hs


def run_mfb_like(case, mhb):
    cfg = make_cfg(case)
    model = (ref_mhb.MHBCoAtt if mhb else ref_mfb.MFB)(cfg)
    load_recipe(model, case["salt"])
    model.eval()
    N, T = case["N"], case["T"]
    img = torch.from_numpy(recipe.img_features(N, cfg.img_feature_dim, cfg.img_feature_channel, case["salt"]))
    q = torch.from_numpy(recipe.question_tokens(N, T, cfg.q_vocab_size, case["salt"]))
    store = {}
    hs = hook_io(model, ["ques_proj1", "co_att_conv1", "co_att_conv2", "img_proj2", "linear_pred"], store)
    if mhb:
        glove = None
        if cfg.glove:
            glove = torch.from_numpy(recipe.sym_tensor((N, T, cfg.emb_dim), 0.5,
                                                       recipe.name_seed("glove", case["salt"])))
        outp = model.forward(img, q, glove_matrix=glove)
        soft = torch.from_numpy(recipe.soft_answers(N, cfg.a_vocab_size, case["salt"]))
        loss = torch.nn.KLDivLoss()(outp, soft)                 # solver.py:27
    else:
        outp = model.forward(img, q)
        ans = torch.from_numpy(recipe.hard_answers(N, cfg.a_vocab_size, case["salt"]))
        loss = torch.nn.CrossEntropyLoss()(outp, ans)           # solver.py:29
    loss.backward()
    for h in hs:
        h.remove()
    out = {"out": outp.detach().numpy(), "loss": np.array(loss.item())}
    tensor_digest("ques_att_feature", store["ques_proj1.in"], out)
    tensor_digest("fusion_normed", store["co_att_conv1.in"], out)
    tensor_digest("co_att_logits", store["co_att_conv2.out"], out)
    tensor_digest("co_att_feature", store["img_proj2.in"], out)
    tensor_digest("att_normed", store["linear_pred.in"], out)
    grad_digest(model, out)
    # the same reference modules in float64: conditioning-free gradients (see tests/test_oracle_golden.py)
    model.double()
    model.zero_grad()
    for p_ in model.parameters():
        p_.grad = None
    if mhb:
        o64 = model.forward(img.double(), q, glove_matrix=None if glove is None else glove.double())
        l64 = torch.nn.KLDivLoss()(o64, soft.double())
    else:
        o64 = model.forward(img.double(), q)
        l64 = torch.nn.CrossEntropyLoss()(o64, ans)
    l64.backward()
    out["out64"] = o64.detach().numpy()
    out["loss64"] = np.array(l64.item())
    grad_digest(model, out, tag="64")
    return out


def run_hie(case):
    model = ref_hie.HieCoAtten(block_num=case["L"], word_num=case["T"], img_size=case["img_size"],
                               vocab_size=case["V"], embed_size=case["E"], output_size=case["A"])
    load_recipe(model, case["salt"])
    N = case["N"]
    img = torch.from_numpy(recipe.img_features(N, case["L"], case["img_size"], case["salt"]))
    q = torch.from_numpy(recipe.question_tokens(N, case["T"], case["V"], case["salt"]))
    with _no_functional_dropout():
        x, av, aq = model.forward(img, q)
    ans = torch.from_numpy(recipe.hard_answers(N, case["A"], case["salt"]))
    loss = torch.nn.CrossEntropyLoss()(x, ans)
    loss.backward()
    out = {"x": x.detach().numpy(), "av": av.detach().numpy(), "aq": aq.detach().numpy(),
           "loss": np.array(loss.item())}
    grad_digest(model, out)
    return out


def run_attnet(case):
    model = ref_net.AttentionNet(block_num=case["L"], word_num=case["T"], img_size=case["img_size"],
                                 vocab_size=case["V"], embed_size=case["E"], att_num=case["att_num"],
                                 output_size=case["A"])
    load_recipe(model, case["salt"])
    model.train()                                   # BatchNorm1d batch statistics (solver.py:67)
    N = case["N"]
    img = torch.from_numpy(recipe.img_features(N, case["L"], case["img_size"], case["salt"]))
    q = torch.from_numpy(recipe.question_tokens(N, case["T"], case["V"], case["salt"], pad_tail=False))
    with _no_functional_dropout():
        x, qa, ia = model.forward(img, q)
    ans = torch.from_numpy(recipe.hard_answers(N, case["A"], case["salt"]))
    loss = torch.nn.CrossEntropyLoss()(x, ans)
    loss.backward()
    out = {"x": x.detach().numpy(), "que_att": qa.detach().numpy(), "img_att": ia.detach().numpy(),
           "loss": np.array(loss.item())}
    grad_digest(model, out)
    return out


def run_ibow(case):
    model = ref_net.iBOWIMG(case["img_size"], case["V"], case["E"], case["A"])
    load_recipe(model, case["salt"])
    model.train()
    N = case["N"]
    img = torch.from_numpy(recipe.sym_tensor((N, case["img_size"]), 1.0, recipe.name_seed("ibow_img", case["salt"])))
    q = torch.from_numpy(recipe.question_tokens(N, case["T"], case["V"], case["salt"]))
    with _no_functional_dropout():
        x = model.forward(img, q)
    ans = torch.from_numpy(recipe.hard_answers(N, case["A"], case["salt"]))
    loss = torch.nn.CrossEntropyLoss()(x, ans)
    loss.backward()
    out = {"x": x.detach().numpy(), "loss": np.array(loss.item())}
    grad_digest(model, out)
    return out


def run_att_module(case):
    kind = case["kind"]
    Dm = case["D"]
    if kind == "attention_1":
        model = ref_mod.Attention_1(Dm)
    elif kind == "attention_2":
        model = ref_mod.Attention_2(Dm)
    elif kind == "attention_layer1":
        model = ref_mod.Attention_layer(Dm, 1)
    elif kind == "attention_layer2":
        model = ref_mod.Attention_layer(Dm, 2)
    else:
        model = ref_mod.Nonlinear_layer(Dm)
    load_recipe(model, case["salt"])
    N, L, T = case["N"], case["L"], case["T"]
    f1 = torch.from_numpy(recipe.sym_tensor((N, L, Dm), 1.0, recipe.name_seed("f1", case["salt"]))).requires_grad_()
    f2 = torch.from_numpy(recipe.sym_tensor((N, T, Dm), 1.0, recipe.name_seed("f2", case["salt"]))).requires_grad_()
    out = {}
    if kind == "nonlinear":
        o = model.forward(f1)
        (o * o).sum().backward()
        out["o"] = o.detach().numpy()
    elif kind.startswith("attention_layer"):
        a, b, att = model.forward(f1, f2)
        ((b * b).sum() + (att * att).sum()).backward()
        out.update(a=a.detach().numpy(), b=b.detach().numpy(), att=att.detach().numpy())
        out["df2"] = f2.grad.numpy()
    else:
        fh, att = model.forward(f1, f2)
        ((fh * fh).sum() + (att * att).sum()).backward()
        out.update(f_hat=fh.detach().numpy(), att=att.detach().numpy())
        if f2.grad is not None:
            out["df2"] = f2.grad.numpy()
    out["df1"] = f1.grad.numpy()
    grad_digest(model, out)
    return out


def main():
    jobs = []
    for c in MFB_CASES:
        jobs.append(("mfb_" + c["name"], lambda c=c: run_mfb_like(c, False)))
    for c in MHBCOATT_CASES:
        jobs.append(("mhbcoatt_" + c["name"], lambda c=c: run_mfb_like(c, True)))
    for c in MHB_CASES:
        jobs.append(("mhb_" + c["name"], lambda c=c: run_mhb(c)))
    for c in HIE_CASES:
        jobs.append(("hie_" + c["name"], lambda c=c: run_hie(c)))
    for c in ATTNET_CASES:
        jobs.append(("attnet_" + c["name"], lambda c=c: run_attnet(c)))
    for c in IBOW_CASES:
        jobs.append(("ibow_" + c["name"], lambda c=c: run_ibow(c)))
    for c in ATT_MODULE_CASES:
        jobs.append(("mod_" + c["name"], lambda c=c: run_att_module(c)))
    only = sys.argv[1:]
    for name, fn in jobs:
        if only and not any(o in name for o in only):
            continue
        out = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-40s %8.1f KB" % (name, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
