"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never import this from the product.

A CPU (PyTorch fp32 ops) restatement of the attention-fusion hot path of
klory/vqa-attention-networks, written functionally over a plain state_dict so
that the same weights can be fed to the HIP path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it,
and only as the checker / the timed CPU baseline.

Pinning: every function here is checked by `tests/test_oracle_golden.py`
against vectors captured from the *imported reference* in the build container
(`tests/golden/make_golden.py`, outputs in `tests/golden/*.npz`).  `mhb_forward`
included (round 4): the reference class `MHB` cannot execute as shipped
(mhb_coAtt.py:176 hard `.cuda()`, :214 undefined name), so its goldens come from
that class compiled from its own text with exactly those two edits
(make_golden.py::load_mhb_class; `.cuda()` -> `.to(img_feature.device)`,
`mhb_22` -> `mhb_12`), fp32 and fp64 (`tests/golden/mhb_*.npz`).

Citations are file:line into the reference repository.
Reference quirks are reproduced on purpose:
  * mfb.py:84,118      softmax over an extent-1 axis  -> attention weights == 1
  * mhb_coAtt.py:27-36,72-74  batch_first LSTM fed (T,N,.) -> recurs over the batch
  * hieCoAtten.py:31   fc_Wbv applied to the question (fc_Wbq unused)
  * hieCoAtten.py:52-53 cat((v,q),0).view(N,-1) row pairing
Dropout: `drop` arguments are optional dicts of explicit keep-masks (1=keep);
None means identity (module.eval() for nn.Dropout; functional dropout patched
to identity when the goldens were captured).
"""
import math
import torch
import torch.nn.functional as F

K_POOL = 5          # mfb.py:42-43,100  k = 5
O_POOL = 1000       # mfb.py:100        o = 1000


# --------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------
def _apply_keep(x, keep, p):
    """nn.Dropout in train mode with an explicit keep mask; identity if None."""
    if keep is None:
        return x
    return x * (keep.to(x.dtype).reshape(x.shape) * (1.0 / (1.0 - p)))


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh):
    """Single-layer uni-directional LSTM over dim 1 of x (B,S,I) -> (B,S,H).

    torch.nn.LSTM gate order i,f,g,o; zero initial state (mfb.py:69,
    mhb_coAtt.py:72-74 pass no hx).
    """
    B, S, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    xw = x @ w_ih.t() + b_ih          # (B,S,4H)
    outs = []
    for s in range(S):
        g = xw[:, s] + h @ w_hh.t() + b_hh
        i, f, gg, o = g.chunk(4, dim=1)
        i, f, o = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o)
        gg = torch.tanh(gg)
        c = f * c + i * gg
        h = o * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, dim=1)


def signed_sqrt(s):
    """mfb.py:104,133 / mhb_coAtt.py:106,133,143,199,210: sqrt(relu(x)) - sqrt(relu(-x))."""
    return torch.sqrt(F.relu(s)) - torch.sqrt(F.relu(-s))


def mfb_pool_norm(z, n_rows):
    """z: (n_rows, 5000*?)  -> sum adjacent groups of 5, signed sqrt, L2 over the row.

    Used for the final blocks (mfb.py:131-135): view(N,1,1000,5).sum(3),
    signed sqrt, F.normalize over the 1000 pooled values.
    """
    s = z.reshape(n_rows, O_POOL, K_POOL).sum(2)
    return F.normalize(signed_sqrt(s))


def glimpse_attention(feat, logits, compat_unit_softmax):
    """feat (N,S,C), logits (N,S,2) -> (N,2C) and the weights (N,2,S).

    mfb.py:82-89 / mhb_coAtt.py:84-91 (question side, S=T) and
    mfb.py:116-123 / mhb_coAtt.py:114-121 (image side, S=L).
    compat_unit_softmax=True reproduces mfb.py:84,118 where the softmax runs
    over the trailing singleton axis, i.e. every weight is exactly 1.
    """
    if compat_unit_softmax:
        w = torch.softmax(logits.permute(0, 2, 1).unsqueeze(-1), dim=3).squeeze(-1)
    else:
        w = torch.softmax(logits.permute(0, 2, 1), dim=2)          # over S
    pooled = torch.einsum("ngs,nsc->ngc", w, feat)
    return pooled.reshape(feat.shape[0], -1), w


# --------------------------------------------------------------------------
# MFB-baseline  (mfb.py:61-140) and MHBCoAtt (mhb_coAtt.py:61-151)
# --------------------------------------------------------------------------
def _coatt_trunk(sd, cfg, img, q, glove, drop, *, mhb, live_softmax=False):
    """Shared ladder up to (ques_att_feature, co_att_feature).

    Returns a dict of intermediates; `mhb` selects the MHBCoAtt deltas.
    live_softmax (MFB only, NOT reference behaviour): take both attention softmaxes over the
    token / region axis as mhb_coAtt.py:84,114 do, instead of mfb.py:84,118's singleton axis;
    the checker for the product's `MFB.unit_softmax = False` mode, in which every tensor is live.
    """
    drop = drop or {}
    unit = (not mhb) and not live_softmax
    multilayer = (not mhb) and getattr(cfg, "model_name", "") == "mfb-multilayer"
    N = img.shape[0]
    L, D = img.shape[1], img.shape[2]

    # a2: question encoder                       mfb.py:68-70 / mhb_coAtt.py:69-75
    e = torch.tanh(F.embedding(q, sd["word_embedding.weight"]))       # (N,T,E)
    if mhb and getattr(cfg, "glove", False):
        assert glove is not None, "glove should not be NoneType."     # mhb_coAtt.py:71
        e = torch.cat((e, glove), dim=2)
    lw = [sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"],
          sd["lstm.bias_ih_l0"], sd["lstm.bias_hh_l0"]]
    if mhb:
        # batch_first LSTM fed (T,N,.): "batch"=T, sequence=N  (mhb_coAtt.py:72-74)
        h_tn = lstm_layer(e.permute(1, 0, 2), *lw)                    # (T,N,H)
        h_tn = _apply_keep(h_tn, drop.get("l"), 0.3)                  # :75
        h = h_tn.permute(1, 0, 2)                                     # (N,T,H) view, :78
    else:
        h = lstm_layer(e, *lw)                                        # (N,T,H)
        h = _apply_keep(h, drop.get("l"), 0.3)                        # mfb.py:70

    # a3: question attention MLP + glimpses      mfb.py:73-89 / mhb_coAtt.py:78-91
    w1 = sd["ques_att_conv1.weight"].flatten(1)
    a = F.relu(h @ w1.t() + sd["ques_att_conv1.bias"])
    if multilayer:
        wm = sd["ques_att_multiconv.weight"].flatten(1)
        a = F.relu(a @ wm.t() + sd["ques_att_multiconv.bias"])
    w2 = sd["ques_att_conv2.weight"].flatten(1)
    qlog = a @ w2.t() + sd["ques_att_conv2.bias"]                     # (N,T,2)
    qa, qw = glimpse_attention(h, qlog, compat_unit_softmax=unit)     # (N,2H)

    # a4: ques_proj1                              mfb.py:92-93
    qp = qa @ sd["ques_proj1.weight"].t() + sd["ques_proj1.bias"]     # (N,5000)

    # a5: image projection (1x1 conv == GEMM)     mfb.py:95-96
    wi = sd["img_conv1d.weight"].flatten(1)
    P = img @ wi.t() + sd["img_conv1d.bias"]                          # (N,L,5000)

    # a6: product, dropout, k-pool, signed sqrt, per-sample L2   mfb.py:98-106
    Z = _apply_keep(P * qp[:, None, :], drop.get("m1"), 0.1)
    S = Z.reshape(N, L, O_POOL, K_POOL).sum(3)                        # (N,L,1000)
    R = signed_sqrt(S)
    Y = F.normalize(R.reshape(N, -1)).reshape(N, L, O_POOL)           # norm over 1000*L

    # a7: co-attention MLP                        mfb.py:109-114
    wc1 = sd["co_att_conv1.weight"].flatten(1)
    c = F.relu(Y @ wc1.t() + sd["co_att_conv1.bias"])
    if multilayer:
        wcm = sd["co_att_multiconv.weight"].flatten(1)
        c = F.relu(c @ wcm.t() + sd["co_att_multiconv.bias"])
    wc2 = sd["co_att_conv2.weight"].flatten(1)
    clog = c @ wc2.t() + sd["co_att_conv2.bias"]                      # (N,L,2)

    # a8: softmax over regions + glimpse sums     mfb.py:116-123
    va, vw = glimpse_attention(img, clog, compat_unit_softmax=unit)   # (N,2D)
    return dict(h=h, qlog=qlog, qw=qw, qa=qa, qp=qp, P=P, S=S, R=R, Y=Y,
                clog=clog, vw=vw, va=va)


def _final_block(sd, qa, va, qname, iname, keep):
    """a9: mfb.py:126-135 (and mhb_coAtt.py:124-145 for *_proj2 / *_proj3)."""
    qq = qa @ sd[qname + ".weight"].t() + sd[qname + ".bias"]
    ii = va @ sd[iname + ".weight"].t() + sd[iname + ".bias"]
    z = _apply_keep(qq * ii, keep, 0.1)
    return mfb_pool_norm(z, qa.shape[0])


def mfb_forward(sd, cfg, img, q, drop=None, return_all=False, live_softmax=False):
    """MFB.forward(img_features, questions) -> logits (N,A).   mfb.py:61-140."""
    drop = drop or {}
    t = _coatt_trunk(sd, cfg, img, q, None, drop, mhb=False, live_softmax=live_softmax)
    y = _final_block(sd, t["qa"], t["va"], "ques_proj2", "img_proj2", drop.get("m2"))
    logits = y @ sd["linear_pred.weight"].t() + sd["linear_pred.bias"]   # :137
    if return_all:
        t.update(y=y, logits=logits)
        return t
    return logits                                                        # :140


def mhbcoatt_forward(sd, cfg, img, q, glove=None, drop=None, return_all=False):
    """MHBCoAtt.forward(img, q, glove_matrix=None) -> log-probs (N,A).  mhb_coAtt.py:61-151."""
    drop = drop or {}
    t = _coatt_trunk(sd, cfg, img, q, glove, drop, mhb=True)
    y2 = _final_block(sd, t["qa"], t["va"], "ques_proj2", "img_proj2", drop.get("m2"))
    y3 = _final_block(sd, t["qa"], t["va"], "ques_proj3", "img_proj3", drop.get("m3"))
    y = torch.cat([y2, y3], 1)                                           # :147
    logits = y @ sd["linear_pred.weight"].t() + sd["linear_pred.bias"]   # :148
    out = F.log_softmax(logits, dim=1)                                   # :149 (implicit dim=1 for 2-D)
    if return_all:
        t.update(y=y, logits=logits, out=out)
        return t
    return out


def mhb_forward(sd, cfg, img, q, q_length, drop=None, return_all=False):
    """MHB.forward(img_feature, questions, q_length) -> log-probs.  mhb_coAtt.py:174-217.

    Device-agnostic zeros instead of .cuda() (:176); mhb_22 -> mhb_12 (:214): the two edits
    the goldens' reference class carries too (module docstring).
    """
    drop = drop or {}
    N, T = q.shape
    C = cfg.img_feature_channel
    # :178-180 view (N,14,14,C) -> permute -> AvgPool2d(14,14)  == mean over the L axis
    i_mean = img.reshape(N, -1, C).mean(1)
    e = F.embedding(q, sd["Embedding.weight"])                         # :181 (no tanh here)
    hs = lstm_layer(e, sd["LSTM.weight_ih_l0"], sd["LSTM.weight_hh_l0"],
                    sd["LSTM.bias_ih_l0"], sd["LSTM.bias_hh_l0"])      # correct orientation, :182-183
    idx = (q_length.to(torch.long) - 1)
    last = hs[torch.arange(N), idx]                                    # :185-186
    last = _apply_keep(last, drop.get("l"), 0.3)                       # :188
    q1 = last @ sd["linear_q_1.weight"].t() + sd["linear_q_1.bias"]
    i1 = i_mean @ sd["linear_i_1.weight"].t() + sd["linear_i_1.bias"]
    z1 = _apply_keep(q1 * i1, drop.get("m1"), 0.1)                     # :192-193
    y1 = mfb_pool_norm(z1, N)                                          # :194-199
    q2 = last @ sd["linear_q_2.weight"].t() + sd["linear_q_2.bias"]
    i2 = i_mean @ sd["linear_i_2.weight"].t() + sd["linear_i_2.bias"]
    z2 = _apply_keep((q2 * i2) * z1, drop.get("m2"), 0.1)              # :204-206
    y2 = mfb_pool_norm(z2, N)                                          # :207-211
    y = torch.cat((y1, y2), 1)                                         # :213
    logits = y @ sd["linear_out.weight"].t() + sd["linear_out.bias"]   # :214 (fixed name)
    out = F.log_softmax(logits, dim=1)                                 # :215
    if return_all:
        return dict(out=out, last=last, i_mean=i_mean, y=y)
    return out


# --------------------------------------------------------------------------
# HieCoAtten  (hieCoAtten.py:18-55)
# --------------------------------------------------------------------------
def hiecoatten_forward(sd, img, q, drop=None):
    """HieCoAtten.forward(img_features, que_features) -> (x, av, aq).

    drop: optional keep masks for the five always-on functional dropouts
    (p=0.5): 'img','que','C','Hv','Hq'.
    """
    drop = drop or {}
    N = img.shape[0]
    lin = lambda x, n: x @ sd[n + ".weight"].t() + sd[n + ".bias"]
    im = _apply_keep(F.relu(lin(img, "img_emb")), drop.get("img"), 0.5)      # :25-26
    qu = _apply_keep(F.embedding(q, sd["que_emb.weight"]), drop.get("que"), 0.5)  # :27-28
    Cv = lin(im, "fc_Wbv")                                                   # :30
    Cq = lin(qu, "fc_Wbv")                                                   # :31  (Wbv, not Wbq)
    C = _apply_keep(torch.tanh(Cq @ Cv.transpose(1, 2)), drop.get("C"), 0.5)  # :32-33 (N,T,L)
    im_ = lin(im, "fc_Wv")                                                   # :35
    qu_ = lin(qu, "fc_Wq")                                                   # :36
    Hv = torch.tanh(im_ + (qu_.transpose(1, 2) @ C).transpose(1, 2))         # :38 (N,L,E)
    Hv = _apply_keep(Hv, drop.get("Hv"), 0.5)
    av = torch.softmax(lin(Hv, "fc_Whv"), dim=1)                             # :40 (N,L,1)
    v = (av.transpose(1, 2) @ im).reshape(N, -1)                             # :41-42
    Hq = torch.tanh(qu_ + (im_.transpose(1, 2) @ C.transpose(1, 2)).transpose(1, 2))  # :45
    Hq = _apply_keep(Hq, drop.get("Hq"), 0.5)
    aq = torch.softmax(lin(Hq, "fc_Whq"), dim=1)                             # :47 (N,T,1)
    qv = (aq.transpose(1, 2) @ qu).reshape(N, -1)                            # :48-49
    x = torch.cat((v, qv), 0).reshape(N, -1)                                 # :52-53 row pairing
    x = lin(x, "fc")                                                         # :54
    return x, av.reshape(N, -1), aq.reshape(N, -1)


# --------------------------------------------------------------------------
# modules.py / networks.py
# --------------------------------------------------------------------------
def attention_1(sd, prefix, f1, f2):
    """Attention_1.forward (modules.py:41-77): additive scores over the (N,T,L,D) broadcast."""
    assert f1.shape[2] == f2.shape[2], "dimension of feature_1 and feature_2 not match"
    hsum = f1[:, None, :, :] + f2[:, :, None, :]                       # :51-57 (N,T,L,D)
    att = (hsum @ sd[prefix + "fc.weight"].t() + sd[prefix + "fc.bias"]).squeeze(-1)  # :60-61
    att = torch.softmax(att, dim=2)                                    # :64
    return att @ f1, att                                               # :65


def attention_2(sd, prefix, f1, f2):
    """Attention_2.forward (modules.py:85-95): bilinear scores."""
    assert f1.shape[2] == f2.shape[2], "dimension of img_feature and q_feature not match"
    g = f1 @ sd[prefix + "fc1.weight"].t()                             # :90
    att = torch.softmax(f2 @ g.transpose(1, 2), dim=2)                 # :91-92
    return att @ f1, att                                               # :94


def attention_layer(sd, prefix, f1, f2, att_type=1):
    """Attention_layer.forward (modules.py:26-33)."""
    a, b = F.relu(f1), F.relu(f2)
    fn = attention_1 if att_type == 1 else attention_2
    f_hat, att = fn(sd, prefix + "att_layer.", a, b)
    return a, F.relu(b + f_hat), att


def nonlinear_layer(sd, prefix, x):
    """Nonlinear_layer.forward (modules.py:103-109): tanh(W1 x) * sigmoid(W2 x)."""
    o1 = x @ sd[prefix + "fc1.weight"].t() + sd[prefix + "fc1.bias"]
    o2 = x @ sd[prefix + "fc2.weight"].t() + sd[prefix + "fc2.bias"]
    return torch.tanh(o1) * torch.sigmoid(o2)


def _batchnorm_train(x, w, b, eps=1e-5):
    m = x.mean(0)
    v = x.var(0, unbiased=False)
    return (x - m) / torch.sqrt(v + eps) * w + b


def attentionnet_forward(sd, img, q, att_num=6, drop=None, bn_training=True):
    """AttentionNet.forward (networks.py:47-69) -> (x, que_att, img_att)."""
    drop = drop or {}
    N = img.shape[0]
    im = _apply_keep(F.relu(img @ sd["img_emb.weight"].t() + sd["img_emb.bias"]),
                     drop.get("img"), 0.5)                             # :54-55
    qu = _apply_keep(F.embedding(q, sd["que_emb.weight"]), drop.get("que"), 0.5)  # :56-57
    que_att = img_att = None
    for i in range(att_num):                                           # :58-62
        if i % 2 == 0:
            im, qu, que_att = attention_layer(sd, "att%d." % i, im, qu)
        else:
            qu, im, img_att = attention_layer(sd, "att%d." % i, qu, im)
    x = torch.cat((que_att, img_att.transpose(1, 2)), 0).reshape(N, -1)  # :64-65
    x = x @ sd["fc.weight"].t() + sd["fc.bias"]                        # :66
    if bn_training:
        x = _batchnorm_train(x, sd["batchnorm.weight"], sd["batchnorm.bias"])  # :68
    else:
        x = (x - sd["batchnorm.running_mean"]) / torch.sqrt(sd["batchnorm.running_var"] + 1e-5) \
            * sd["batchnorm.weight"] + sd["batchnorm.bias"]
    return x, que_att, img_att


def ibowimg_forward(sd, img, q, drop=None, bn_training=True):
    """iBOWIMG.forward (networks.py:15-28); img is (N, img_size)."""
    drop = drop or {}
    x = img @ sd["img_emb.weight"].t() + sd["img_emb.bias"]
    if bn_training:
        x = _batchnorm_train(x, sd["img_bn.weight"], sd["img_bn.bias"])
    else:
        x = (x - sd["img_bn.running_mean"]) / torch.sqrt(sd["img_bn.running_var"] + 1e-5) \
            * sd["img_bn.weight"] + sd["img_bn.bias"]
    im = _apply_keep(F.relu(x), drop.get("img"), 0.5)
    qu = _apply_keep(F.embedding(q, sd["que_emb.weight"]), drop.get("que"), 0.5).sum(1)
    return torch.cat((im, qu), 1) @ sd["fc.weight"].t() + sd["fc.bias"]


# --------------------------------------------------------------------------
# losses as the caller computes them (solver.py:26-30,91)
# --------------------------------------------------------------------------
def ce_loss(logits, a):
    return F.cross_entropy(logits, a)                                  # solver.py:29


def kldiv_loss(logp, soft):
    return F.kl_div(logp, soft, reduction="mean")                      # solver.py:27 default 'mean'


def adam_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """One torch.optim.Adam step (solver.py:29 constructs it with defaults but lr; :93 steps it).
    The optimizer is a third-party dependency of the reference (torch, version unpinned; 2.10.0
    here); this restates its published single-tensor algorithm (amsgrad/maximize off) and
    tests/test_oracle_golden.py pins it against torch.optim.Adam itself.
    `state` is a list of dicts {'step', 'exp_avg', 'exp_avg_sq'} updated in place."""
    b1, b2 = betas
    for p, g, st in zip(params, grads, state):
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        st["step"] += 1
        t = st["step"]
        if weight_decay != 0:
            g = g + weight_decay * p
        st["exp_avg"] += (g - st["exp_avg"]) * (1 - b1)
        st["exp_avg_sq"].mul_(b2).add_(g * g * (1 - b2))
        denom = st["exp_avg_sq"].sqrt() / math.sqrt(1 - b2 ** t) + eps
        p -= (lr / (1 - b1 ** t)) * (st["exp_avg"] / denom)


# --------------------------------------------------------------------------
# state_dict shape tables (what the reference constructors allocate)
# --------------------------------------------------------------------------
def mfb_shapes(cfg, mhb=False):
    """mfb.py:25-59 / mhb_coAtt.py:25-59."""
    H, E, D = cfg.hidden_dim, cfg.emb_dim, cfg.img_feature_channel
    att_h = 512 if mhb else 1024
    lstm_in = E * 2 if (mhb and getattr(cfg, "glove", False)) else E
    multilayer = (not mhb) and getattr(cfg, "model_name", "") == "mfb-multilayer"
    s = {
        "word_embedding.weight": (cfg.q_vocab_size, E),
        "lstm.weight_ih_l0": (4 * H, lstm_in), "lstm.weight_hh_l0": (4 * H, H),
        "lstm.bias_ih_l0": (4 * H,), "lstm.bias_hh_l0": (4 * H,),
        "ques_att_conv1.weight": (att_h, H, 1, 1), "ques_att_conv1.bias": (att_h,),
    }
    last = att_h
    if multilayer:
        s["ques_att_multiconv.weight"] = (512, 1024, 1, 1)
        s["ques_att_multiconv.bias"] = (512,)
        last = 512
    s.update({
        "ques_att_conv2.weight": (2, last, 1, 1), "ques_att_conv2.bias": (2,),
        "ques_proj1.weight": (5000, 2 * H), "ques_proj1.bias": (5000,),
        "img_conv1d.weight": (5000, D, 1, 1), "img_conv1d.bias": (5000,),
        "co_att_conv1.weight": (att_h, 1000, 1, 1), "co_att_conv1.bias": (att_h,),
    })
    if multilayer:
        s["co_att_multiconv.weight"] = (512, 1024, 1, 1)
        s["co_att_multiconv.bias"] = (512,)
    s.update({
        "co_att_conv2.weight": (2, last, 1, 1), "co_att_conv2.bias": (2,),
        "ques_proj2.weight": (5000, 2 * H), "ques_proj2.bias": (5000,),
        "img_proj2.weight": (5000, 2 * D), "img_proj2.bias": (5000,),
    })
    if mhb:
        s.update({
            "ques_proj3.weight": (5000, 2 * H), "ques_proj3.bias": (5000,),
            "img_proj3.weight": (5000, 2 * D), "img_proj3.bias": (5000,),
            "linear_pred.weight": (cfg.a_vocab_size, 2000), "linear_pred.bias": (cfg.a_vocab_size,),
        })
    else:
        s.update({"linear_pred.weight": (cfg.a_vocab_size, 1000),
                  "linear_pred.bias": (cfg.a_vocab_size,)})
    return s


def mhb_shapes(cfg):
    """mhb_coAtt.py:154-172."""
    H, E, D = cfg.hidden_dim, cfg.emb_dim, cfg.img_feature_channel
    return {
        "Embedding.weight": (cfg.q_vocab_size, E),
        "LSTM.weight_ih_l0": (4 * H, E), "LSTM.weight_hh_l0": (4 * H, H),
        "LSTM.bias_ih_l0": (4 * H,), "LSTM.bias_hh_l0": (4 * H,),
        "linear_q_1.weight": (5000, H), "linear_q_1.bias": (5000,),
        "linear_q_2.weight": (5000, H), "linear_q_2.bias": (5000,),
        "linear_i_1.weight": (5000, D), "linear_i_1.bias": (5000,),
        "linear_i_2.weight": (5000, D), "linear_i_2.bias": (5000,),
        "linear_out.weight": (cfg.a_vocab_size, 2000), "linear_out.bias": (cfg.a_vocab_size,),
    }


def hiecoatten_shapes(img_size, vocab_size, embed_size, output_size):
    """hieCoAtten.py:6-16."""
    E = embed_size
    s = {"img_emb.weight": (E, img_size), "img_emb.bias": (E,),
         "que_emb.weight": (vocab_size, E)}
    for n in ("fc_Wbv", "fc_Wbq", "fc_Wv", "fc_Wq"):
        s[n + ".weight"] = (E, E)
        s[n + ".bias"] = (E,)
    for n in ("fc_Whv", "fc_Whq"):
        s[n + ".weight"] = (1, E)
        s[n + ".bias"] = (1,)
    s["fc.weight"] = (output_size, 2 * E)
    s["fc.bias"] = (output_size,)
    return s


def attentionnet_shapes(block_num, word_num, img_size, vocab_size, embed_size, att_num, output_size):
    """networks.py:31-45."""
    E = embed_size
    s = {"img_emb.weight": (E, img_size), "img_emb.bias": (E,),
         "que_emb.weight": (vocab_size, E)}
    for i in range(att_num):
        s["att%d.att_layer.fc.weight" % i] = (1, E)
        s["att%d.att_layer.fc.bias" % i] = (1,)
    s["fc.weight"] = (output_size, 2 * block_num * word_num)
    s["fc.bias"] = (output_size,)
    s["batchnorm.weight"] = (output_size,)
    s["batchnorm.bias"] = (output_size,)
    return s
