"""Node-by-node checks of a real training step (test infrastructure; the oracle here is torch in fp64 / fp32 on the GPU).

Why node by node.  At the BASELINE batches the model-level gradient criterion (golden_util.grad_parity) stops discriminating
for the tensors behind the signed square root (mfb.py:104,133: derivative 0.5*|s|^-1/2): the CPU fp32 path itself sits 5-300 %
from the fp64 gradient there (profiles/r04_grad_parity.log), so a bound of 8x that noise accepts anything.  Node-locally no
such amplification exists: every autograd node of the step is replayed alone and compared with an fp64 evaluation of THE SAME
node from THE SAME operands (the kernel's own stored projection P is taken as given, so no rounding decision is re-made
upstream of a singular derivative).

The step is run ONCE through the product modules with every `Function.apply` recorded (inputs, outputs, incoming gradients).
  (1) wiring: each node replayed alone on its recorded inputs gives BIT-identical outputs; the gradients it hands to
      parameters are BIT-identical to what the model's backward left in p.grad; the gradients the consumers hand to an
      intermediate tensor add up to the gradient recorded at its producer (1e-6);
  (2) numerics, per node kind, norm-relative to the fp64 evaluation of the node:
      fp32 products (LinearFn, ImgProj*Fn, the projections inside FinalMfbFn): 2e-6 * max(1, sqrt(K)/8), K = the reduction
          length of that product (fp32 accumulation in a fixed k order: u * sqrt(K)-scale; the bound of the fp64 launch tests);
      bf16-operand products: 2e-3 vs fp64 on the same bf16-rounded operands;
      AttHeadFn (two products, softmax, pooling; K <= 2048): 2e-5 fp32 / 2e-3 bf16, the fp64 evaluation taking the kernel's
          own ReLU decisions (hidden layer > 0) as given -- one pre-activation within fp32 rounding of 0 decided the other way
          moves the downstream gradients by 1e-4 of their norm, and the image-side head has 5e7 of them; its dead forms
          (mfb.py:84,118 singleton softmax) must hand EXACT zeros to the MLP;
      LstmBatchFn (14 dependent steps) 5e-5, LstmSeqFn (512 dependent steps, mhb_coAtt.py:72-74) 2e-4: rounding compounds
          along the recursion; EmbedTanhFn 2e-6 (the fast tanh: 1e-7 absolute); DropoutBTFn, LogSoftmaxRowsFn 1e-5;
      HieCoreFn (hieCoAtten.py:25-53 as ONE node of a dozen chained stages: products with K <= 2048, tanh, two softmaxes):
          outputs and gradients 5e-4 at most, and every gradient within 4x the distance a torch-fp32 evaluation of the same node
          keeps from fp64 (measured at B = 256: 2e-6 ... 8e-5 for torch, 1.0-2.3x that for the kernels; both printed);
      the MFB fusion nodes (ImgFuseFn, MfbFuseFn, FinalMfbFn): outputs 1e-5; gradients by the node-local conditioning bound --
          as close to fp64 as an fp32 torch evaluation of the same formulas from the same P is (x8, floor 5e-4 fp32 / 2e-3
          bf16): the pooled sums s = sum_5 P q are formed inside the node, and where |s| is at rounding level the gradient
          is not determined to better than that by ANY fp32 evaluation.
"""
import os

import torch

BENIGN_TOL_BF16 = 2e-3      # nodes fed bf16-rounded operands, fp32 accumulate, vs fp64 on the same rounded operands
ATT_TOL_F32 = 2e-5
LSTM_BATCH_TOL_F32 = 5e-5
LSTM_SEQ_TOL_F32 = 2e-4
HIE_TOL_F32 = 5e-4       # cap; each gradient must also be within 4x the torch-fp32 evaluation's own distance from fp64
FUSE_OUT_TOL = 1e-5
FUSE_FLOOR_F32, FUSE_FLOOR_BF16 = 5e-4, 2e-3


def gemm_tol(K):
    """norm-relative bound of an fp32 product with reduction length K against fp64"""
    return 2e-6 * max(1.0, float(K) ** 0.5 / 8.0)


# ---------------------------------------------------------------------------------------------------------------
# rounding helpers of the emulation
def bf(x):
    """round-to-nearest-even to bf16, back in x's dtype"""
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


class _RoundSTE(torch.autograd.Function):          # forward: bf16 rounding; backward: identity
    @staticmethod
    def forward(ctx, x):
        return bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _GradRound(torch.autograd.Function):         # forward: identity; backward: the gradient is rounded to bf16
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return bf(g)


def _nrel(a, ref):
    a, ref = a.detach().double(), ref.detach().double()
    return float((a - ref).norm() / max(float(ref.norm()), 1e-30))


def _cond_check(name, got, r64, r32, k=8.0, floor=2e-3):
    """node-local grad_parity: |got - r64| <= max(k |r32 - r64|, floor |r64|)"""
    r64 = r64.detach().double()
    err = float((got.detach().double() - r64).norm())
    noise = float((r32.detach().double() - r64).norm())
    bound = max(k * noise, floor * float(r64.norm()))
    assert err <= bound, (name, "err %.3e bound %.3e noise %.3e norm %.3e" % (err, bound, noise, float(r64.norm())))
    return err / bound


# ---------------------------------------------------------------------------------------------------------------
# reference formulas (dtype-generic torch)
def ref_fuse(P, q, N, L, normalise=True):
    """mfb.py:98-106 / mhb_coAtt.py:100-108 without dropout: product, k=5 sum-pool, signed sqrt, per-sample L2
    (normalise=False: the signed square roots R themselves, the output of a fusion node with a NormLink)."""
    z = P.view(N, L, -1) * q[:, None, :]
    s = z.view(N, L, -1, 5).sum(-1)
    r = torch.sqrt(torch.relu(s)) - torch.sqrt(torch.relu(-s))
    if not normalise:
        return r.reshape(N * L, -1)
    nrm = r.reshape(N, -1).norm(dim=1).clamp_min(1e-12)
    return (r / nrm[:, None, None]).reshape(N * L, -1)


def linked_cotangent(R, dYs, N):
    """A fusion node with a NormLink outputs R and is handed dYs = dL/dR with 1/||R_n|| held CONSTANT (its consumer applies
    that factor in its GEMM epilogue).  The total gradient adds the dependence of 1/||R_n|| on R:
    dR = dYs - R_n (sum_n R dYs) / ||R_n||^2   (F.normalize's backward written for the un-normalised tensor)."""
    Rn, d = R.detach().reshape(N, -1), dYs.reshape(N, -1)
    s = (Rn * d).sum(1, keepdim=True)
    n2 = (Rn * Rn).sum(1, keepdim=True).clamp_min(1e-24)
    return (d - Rn * s / n2).reshape(R.shape)


def ref_att_head(x, feat, w1, b1, w2, b2, bf16, inv_rows=None, unit=False, mask=None):
    """AttHeadFn without the multilayer conv: conv1 + ReLU -> 2 logits -> softmax over S -> glimpse sums.
    unit: mfb.py:84,118 -- the softmax runs over a singleton axis, every weight is 1 and the MLP is dead.
    mask (the kernel's own hidden layer > 0): the ReLU's derivative is discontinuous, and of the 5e7 pre-activations of the
    image-side head a handful sit within fp32 rounding of 0 -- ONE sign decided the other way moves the gradients downstream
    by 1e-4 of their norm (tools/att_head_probe.py).  The node check takes the kernel's decisions as given, like the stored
    projection of the fusion nodes: relu(pre) is evaluated as pre * mask.
    inv_rows (N*S,): the NormLink form -- x is the un-normalised fusion output, the per-sample 1/norm (a CONSTANT of this
    node) multiplies the conv's accumulator; the gradient that enters the bf16 products is rounded AFTER that factor."""
    N, S, C = feat.shape
    sc = 1.0 if inv_rows is None else inv_rows[:, None]
    if bf16:
        hid = _GradRound.apply(_RoundSTE.apply(x) @ _RoundSTE.apply(w1).t()) * sc + b1
    else:
        hid = (x @ w1.t()) * sc + b1
    hid = torch.relu(hid) if mask is None else hid * mask.to(hid.dtype)
    logits = (hid @ w2.t() + b2).view(N, S, -1)
    if unit:
        wts = torch.softmax(logits.unsqueeze(-1), dim=-1).squeeze(-1)      # softmax over an axis of extent 1 == 1.0
    else:
        wts = torch.softmax(logits, dim=1)                   # over the S positions, per glimpse
    return torch.einsum("nsg,nsc->ngc", wts, feat).reshape(N, -1)


class _RecProd(torch.autograd.Function):
    """the recurrent product h W_hh^T with bf16 operands: forward bf(h) bf(W)^T; backward dh = bf(g) bf(W); dW from the rounded
    operands too (wgrad_bf16: LstmSeqFn in both bf16 modes, LstmBatchFn in "bf16-all") or from the unrounded g, h (LstmBatchFn
    in "bf16": its recurrent weight gradient is an fp32 product)"""
    @staticmethod
    def forward(ctx, h, w, wgrad_bf16):
        ctx.save_for_backward(h, w)
        ctx.wgrad_bf16 = wgrad_bf16
        return bf(h) @ bf(w).t()

    @staticmethod
    def backward(ctx, g):
        h, w = ctx.saved_tensors
        gb = bf(g)
        return gb @ bf(w), (gb.t() @ bf(h)) if ctx.wgrad_bf16 else (g.t() @ h), None


def ref_lstm_seq(x, w_ih, w_hh, b_ih, b_hh, bf16, wgrad_bf16=True):
    """LstmSeqFn / LstmBatchFn: recursion over dim 0 of x (S,B,I); bf16 = True: bf16 operands in the recurrent product (forward:
    h and W_hh, backward: dG and W_hh); "all": also in the input projection and its two gradients."""
    S, B, I = x.shape
    H = w_hh.shape[1]
    if bf16 == "all":
        xw = _GradRound.apply(_RoundSTE.apply(x.reshape(S * B, I)) @ _RoundSTE.apply(w_ih).t()) + (b_ih + b_hh)
    else:
        xw = x.reshape(S * B, I) @ w_ih.t() + (b_ih + b_hh)
    xw = xw.view(S, B, 4 * H)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = []
    for s in range(S):
        rec = (_RecProd.apply(h, w_hh, wgrad_bf16) if bf16 else h @ w_hh.t()) if s else 0.0
        g = xw[s] + rec
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 0)


def ref_hie_core(imgf, ids, w_emb, b_emb, w_que, wbv, bbv, wv, bv, wq, bq, whv, bhv, whq, bhq, mask=None):
    """hieCoAtten.py:25-53 with the five functional dropouts at rate 0 -> (cat((v, q), 0).view(N, -1), av (N,1,L), aq (N,1,T)).
    mask: the kernel's own ReLU decisions of :25 (see ref_att_head)."""
    N = imgf.shape[0]
    im = imgf @ w_emb.t() + b_emb                                             # :25-26
    im = torch.relu(im) if mask is None else im * mask.to(im.dtype)
    qu = w_que[ids]                                                           # :27-28
    Cv, Cq = im @ wbv.t() + bbv, qu @ wbv.t() + bbv                           # :30-31 (fc_Wbv on both sides)
    C = torch.tanh(Cq @ Cv.transpose(1, 2))                                   # :32-33  (N,T,L)
    im_, qu_ = im @ wv.t() + bv, qu @ wq.t() + bq                             # :35-36
    Hv = torch.tanh(im_ + (qu_.transpose(1, 2) @ C).transpose(1, 2))          # :38
    av = torch.softmax(Hv @ whv.t() + bhv, dim=1)                             # :40  (N,L,1)
    v = (av.transpose(1, 2) @ im).reshape(N, -1)                              # :41-42
    Hq = torch.tanh(qu_ + (im_.transpose(1, 2) @ C.transpose(1, 2)).transpose(1, 2))   # :45
    aq = torch.softmax(Hq @ whq.t() + bhq, dim=1)                             # :47
    qv = (aq.transpose(1, 2) @ qu).reshape(N, -1)                             # :48-49
    return torch.cat((v, qv), 0).reshape(N, -1), av.transpose(1, 2), aq.transpose(1, 2)     # :52-53


# ---------------------------------------------------------------------------------------------------------------
class Recorder:
    def __init__(self, monkeypatch, fns, names):
        self.records, self.orig = [], {}
        for name in names:
            cls = getattr(fns, name)
            self.orig[cls] = cls.apply
            monkeypatch.setattr(cls, "apply", staticmethod(self._wrap(cls)))

    def _wrap(self, cls):
        def call(*args):
            out = self.orig[cls](*args)
            outs = out if isinstance(out, tuple) else (out,)
            rec = dict(cls=cls, args=args, out=out, dout=None, douts=[None] * len(outs))
            for j, o in enumerate(outs):
                if torch.is_tensor(o) and o.requires_grad:
                    def hook(g, rec=rec, j=j):
                        if g is None:           # an output nobody differentiates (HieCoAtten's av / aq under a loss on x)
                            return
                        rec["douts"][j] = g.detach().clone()
                        if j == 0:
                            rec["dout"] = rec["douts"][0]
                    o.register_hook(hook)
            self.records.append(rec)
            return out
        return call

    def replay(self, rec, between=None):
        """the node alone: fresh leaves from the recorded inputs -> (output, {arg index: gradient}, saved tensors); `between(out)`
        runs after the forward (a linked fusion node gets its consumer's (dlogits, lin) back there; a deferred projection is
        filled there)"""
        args2 = [a.detach().clone().requires_grad_(a.requires_grad) if (torch.is_tensor(a) and a.is_floating_point()) else a
                 for a in rec["args"]]
        out = self.orig[rec["cls"]](*args2)
        outs = out if isinstance(out, tuple) else (out,)
        saved = tuple(outs[0].grad_fn.saved_tensors)          # before the backward frees them (the kernel's own P, qq, vv)
        if between is not None:
            between(out)
        idx = [i for i, a in enumerate(args2) if torch.is_tensor(a) and a.requires_grad]
        live = [j for j, d in enumerate(rec["douts"]) if d is not None]
        grads = torch.autograd.grad([outs[j] for j in live], [args2[i] for i in idx],
                                    grad_outputs=[rec["douts"][j] for j in live], allow_unused=True)
        return out, dict(zip(idx, grads)), saved


def _dbl(t, dt):
    return None if t is None else t.detach().to(dt)


def _check_linear(rec, out, grads, rep):
    x, w, b, relu, bf16 = (list(rec["args"]) + [False, False])[:5]
    assert not relu
    w2 = w.reshape(w.shape[0], -1)
    use_bf16 = bool(bf16) and all(d % 8 == 0 for d in (x.shape[0], x.shape[1], w2.shape[0]))
    xo, wo = (bf(x.detach()), bf(w2.detach())) if use_bf16 else (x.detach(), w2.detach())
    M, K = x.shape
    Nn = w2.shape[0]
    tols = {"y": gemm_tol(K), 0: gemm_tol(Nn), 1: gemm_tol(M), 2: gemm_tol(M)}
    if use_bf16:
        tols = {k: BENIGN_TOL_BF16 for k in tols}
    dy = rec["dout"]
    y64 = xo.double() @ wo.double().t() + (b.detach().double() if b is not None else 0.0)
    dyo = bf(dy) if use_bf16 else dy
    worst = _nrel(out, y64)
    assert worst <= tols["y"], ("LinearFn y", worst, tols["y"])
    ratio = worst / tols["y"]
    for i, ref in ((0, dyo.double() @ wo.double()), (1, (dyo.double().t() @ xo.double()).view_as(w)), (2, dy.double().sum(0))):
        if i in grads and grads[i] is not None:
            e = _nrel(grads[i], ref)
            assert e <= tols[i], ("LinearFn grad of arg %d" % i, e, tols[i], tuple(x.shape), tuple(w2.shape))
            worst, ratio = max(worst, e), max(ratio, e / tols[i])
    rep.append("LinearFn%s %s: %.1e (err/bound %.2f)" % ("[bf16]" if use_bf16 else "", tuple(w2.shape), worst, ratio))


def _check_linear2(rec, out, grads, rep):
    """Linear2Fn: y = [x1 | x2] W^T + b without the concatenated tensor (mhb_coAtt.py:147-148,213-214)"""
    x1, x2, w, b = rec["args"][:4]
    w2 = w.reshape(w.shape[0], -1)
    M, K1 = x1.shape
    K = K1 + x2.shape[1]
    dy = rec["dout"].double()
    xc = torch.cat((x1.detach(), x2.detach()), 1).double()
    y64 = xc @ w2.detach().double().t() + (b.detach().double() if b is not None else 0.0)
    worst = _nrel(out, y64)
    assert worst <= gemm_tol(K), ("Linear2Fn y", worst)
    dx = dy @ w2.detach().double()
    refs = {0: (dx[:, :K1], gemm_tol(w2.shape[0])), 1: (dx[:, K1:], gemm_tol(w2.shape[0])), 2: ((dy.t() @ xc).view_as(w), gemm_tol(M)),
            3: (dy.sum(0), gemm_tol(M))}
    ratio = worst / gemm_tol(K)
    for i, (ref, tol) in refs.items():
        if i in grads and grads[i] is not None:
            e = _nrel(grads[i], ref)
            assert e <= tol, ("Linear2Fn grad of arg %d" % i, e, tol)
            worst, ratio = max(worst, e), max(ratio, e / tol)
    rep.append("Linear2Fn %s: %.1e (err/bound %.2f)" % (tuple(w2.shape), worst, ratio))


def _check_logsoftmax(rec, out, grads, rep):
    x = rec["args"][0].detach().double().requires_grad_(True)
    y = torch.log_softmax(x, dim=1)
    (dx,) = torch.autograd.grad(y, x, rec["dout"].double())
    e = max(_nrel(out, y), _nrel(grads[0], dx))
    assert e <= 1e-5, ("LogSoftmaxRowsFn", e)
    rep.append("LogSoftmaxRowsFn: %.1e" % e)


def _check_att_head(rec, out, grads, rep, saved):
    x, feat, w1, b1, wm, bm, w2, b2, unit, bf16 = rec["args"][:10]
    link = rec["args"][10] if len(rec["args"]) > 10 else None
    same_src = bool(rec["args"][11]) if len(rec["args"]) > 11 else False     # x = feat.view(N*S, C): ONE gradient, handed to feat
    assert wm is None
    unit = bool(unit)
    inv_rows = link.inv.detach().double().repeat_interleave(link.L) if link is not None else None
    bf16 = bool(bf16)
    dt = torch.float64
    mask = saved[5] > 0                          # saved: (x, feat, w1, wm, w2, hid1, hid2, wts, lin): the kernel's ReLU decisions
    leaves = {0: _dbl(x, dt), 2: _dbl(w1.reshape(w1.shape[0], -1), dt), 3: _dbl(b1, dt),
              6: _dbl(w2.reshape(w2.shape[0], -1), dt), 7: _dbl(b2, dt)}
    if feat.requires_grad:                       # question side: the LSTM states are pooled AND feed the MLP
        leaves[1] = _dbl(feat, dt)
    for t in leaves.values():
        t.requires_grad_(True)
    if same_src and grads.get(0) is None:        # the MLP reads the pooled tensor itself: its input gradient landed in feat's
        assert 1 in leaves                       # (the bf16 head returns the two gradients separately)
        leaves[0] = leaves[1].view(x.shape)
    y = ref_att_head(leaves[0], leaves.get(1, _dbl(feat, dt)), leaves[2], leaves[3], leaves[6], leaves[7], bf16, inv_rows, unit, mask)
    tol = BENIGN_TOL_BF16 if bf16 else ATT_TOL_F32
    worst = _nrel(out, y)
    assert worst <= tol, ("AttHeadFn pooled", worst)
    if unit:
        # the dead MLP: exact zeros for conv1 / conv2 and for the MLP's input; the pooling's gradient into feat is the
        # broadcast of dpooled (exact: one value per element)
        for i in (0, 2, 3, 6, 7):
            if i in grads and grads[i] is not None:
                assert float(grads[i].abs().max()) == 0.0, ("AttHeadFn (singleton softmax): arg %d must get exact zeros" % i)
        if 1 in grads and grads[1] is not None:
            (g,) = torch.autograd.grad(y, [leaves[1]], rec["dout"].double())
            e = _nrel(grads[1], g)
            assert e <= 1e-6, ("AttHeadFn (singleton softmax) dfeat", e)
            worst = max(worst, e)
        rep.append("AttHeadFn[unit] x%s: %.1e, MLP gradients exactly 0" % (tuple(x.shape), worst))
        return
    want = [i for i in leaves if i in grads and grads[i] is not None]
    g64 = torch.autograd.grad(y, [leaves[i] for i in want], rec["dout"].double())
    for i, g in zip(want, g64):
        e = _nrel(grads[i].reshape(g.shape), g)
        # the two biases in front of the softmax (b2) carry a mathematically-zero gradient: compare against the scale of dw2
        if i == 7:
            scale = float(g64[want.index(6)].norm()) if 6 in want else 1.0
            assert float((grads[i].double() - g).norm()) <= tol * max(scale, 1e-30), ("AttHeadFn db2", e)
            continue
        assert e <= tol, ("AttHeadFn grad of arg %d" % i, e, tuple(x.shape))
        worst = max(worst, e)
    rep.append("AttHeadFn%s%s x%s: %.1e (bound %.0e)" % ("[bf16]" if bf16 else "", "[linked]" if link is not None else "",
                                                          tuple(x.shape), worst, tol))


def _sampled_product_check(name, P_k, X, W, bias, K, bf16):
    """the stored projection against the fp64 product of its operands on every 7th row"""
    rows = torch.arange(0, X.shape[0], 7, device=X.device)
    P64 = X[rows].double() @ W.double().t() + (bias.detach().double() if bias is not None else 0.0)
    if bf16:
        assert float(((P_k[rows].double() - P64).abs() - (2.0 ** -8) * P64.abs()).max()) <= 1e-6, name + ": stored P is not a bf16 rounding of the product"
        eq = float((P_k[rows] == P64.to(torch.float32).to(torch.bfloat16)).float().mean())
        assert eq >= 0.98, (name + ": stored P vs RNE(fp64 product)", eq)
        return "P bit-equal %.4f" % eq
    e = _nrel(P_k[rows], P64)
    assert e <= gemm_tol(K), (name + ": projection vs fp64", e, gemm_tol(K))
    return "P %.1e (bound %.1e)" % (e, gemm_tol(K))


def _check_img_fuse(rec, out, grads, rep, saved):
    img, wi, bi, q, keep, seed, p_drop, bf16 = rec["args"][:8]
    assert keep is None and p_drop == 0.0
    bf16 = bool(bf16)
    linked = len(rec["args"]) > 8 and rec["args"][8] is not None       # NormLink: the node outputs R, receives dYs
    N, L, D = img.shape
    img_k, P_k = saved[0], saved[3]                               # saved: (img2d, wi, q, P, Y, norm, inv, keep)
    if bf16:
        assert img_k.dtype == torch.bfloat16 and P_k.dtype == torch.bfloat16, "config 3 stores the image grid and P in bf16"
    rnd = bf if bf16 else (lambda t: t)
    wb = rnd(wi.detach().reshape(wi.shape[0], -1))
    dY = rec["dout"]
    img2 = img_k.reshape(N * L, D)
    pmsg = _sampled_product_check("ImgFuseFn", P_k, img2, wb, bi, D, bf16)
    # fusion forward / backward from the kernel's own P, fp64 and fp32 (the noise of the formulas themselves)
    res = {}
    for dt in (torch.float64, torch.float32):
        Pl, ql = P_k.to(dt).requires_grad_(True), q.detach().to(dt).requires_grad_(True)
        Y = ref_fuse(Pl, ql, N, L, normalise=not linked)
        dP, dq = torch.autograd.grad(Y, [Pl, ql], linked_cotangent(Y, dY.to(dt), N) if linked else dY.to(dt))
        dwi = (rnd(dP).double().t() @ img2.double()).view_as(wi)         # bf16 mode: the bf16 hand-off dW = RNE(dP)^T X
        res[dt] = (Y.detach(), dq, dP.sum(0), dwi)
        del Pl, ql, Y, dP
    y_err = _nrel(out, res[torch.float64][0])
    assert y_err <= FUSE_OUT_TOL, ("ImgFuseFn Y", y_err)
    worst = 0.0
    floor = FUSE_FLOOR_BF16 if bf16 else FUSE_FLOOR_F32
    for name, i, j in (("dq", 3, 1), ("dbias", 2, 2), ("dW", 1, 3)):
        worst = max(worst, _cond_check("ImgFuseFn " + name, grads[i], res[torch.float64][j], res[torch.float32][j], floor=floor))
    rep.append("ImgFuseFn%s%s N=%d: Y %.1e, %s, grads err/bound %.2f" % ("[bf16]" if bf16 else "", "[linked]" if linked else "", N,
                                                                       y_err, pmsg, worst))


def _check_img_proj(rec, out, grads, rep):
    """The projection as its own node.  ImgProjLateFn(P0, img2, wi, cu_limit): P0 was computed early without a node, the node
    owns the weight gradient dW = dP^T X.  ImgProjDeferFn(img, wi) / ImgProjFn(img, wi, bf16): the node's output is P0."""
    name = rec["cls"].__name__
    if name == "ImgProjLateFn":
        P0, img2, wi = rec["args"][:3]
        widx = 2
    else:
        img, wi = rec["args"][:2]
        img2, P0, widx = img.reshape(-1, img.shape[-1]), out, 1
        if name == "ImgProjFn" and len(rec["args"]) > 2 and rec["args"][2]:
            img2 = bf(img2.detach()).to(torch.bfloat16)
    bf16 = img2.dtype == torch.bfloat16
    rnd = bf if bf16 else (lambda t: t)
    wb = rnd(wi.detach().reshape(wi.shape[0], -1))
    dout = rec["dout"]
    pmsg = _sampled_product_check(name, P0.detach(), img2, wb, None, img2.shape[1], bf16)
    g = grads[widx]
    if float(dout.abs().max()) == 0.0:
        # faithful MFB (mfb.py:84,118): the fusion is dead, dP == 0 exactly, and so must be the weight gradient
        assert float(g.abs().max()) == 0.0, (name, "dW of an all-zero dP must be exactly zero")
        rep.append("%s%s: %s, dP == 0 -> dW exactly 0" % (name, "[bf16]" if bf16 else "", pmsg))
        return
    ref = (dout.double().t() @ img2.double()).view_as(wi)
    e = _nrel(g, ref)
    tol = BENIGN_TOL_BF16 if bf16 else gemm_tol(img2.shape[0])
    assert e <= tol, (name + " dW", e, tol)
    rep.append("%s%s: %s, dW %.1e (bound %.1e)" % (name, "[bf16]" if bf16 else "", pmsg, e, tol))


def _check_mfb_fuse(rec, out, grads, rep):
    """the image fusion on a projection handed in: P0 (+ bias inside the kernel) in, dP out, fp32 or bf16 storage of both"""
    P0, bi, q, keep, seed, p_drop, N, L = rec["args"][:8]
    assert keep is None and p_drop == 0.0
    bf16 = P0.dtype == torch.bfloat16
    linked = len(rec["args"]) > 8 and rec["args"][8] is not None
    dead = float(rec["dout"].abs().max()) == 0.0
    res = {}
    for dt in (torch.float64, torch.float32):
        Pl = P0.detach().to(dt).requires_grad_(True)
        bl, ql = bi.detach().to(dt).requires_grad_(True), q.detach().to(dt).requires_grad_(True)
        Y = ref_fuse(Pl + bl, ql, N, L, normalise=not linked)
        if dead:
            res[dt] = (Y.detach(),)
            del Pl, Y
            break
        dP, db, dq = torch.autograd.grad(Y, [Pl, bl, ql], linked_cotangent(Y, rec["dout"].to(dt), N) if linked else rec["dout"].to(dt))
        res[dt] = (Y.detach(), dP, db, dq)
        del Pl, Y
    y_err = _nrel(out, res[torch.float64][0])
    assert y_err <= FUSE_OUT_TOL, ("MfbFuseFn Y", y_err)
    assert grads[0].dtype == P0.dtype
    tag = "MfbFuseFn%s%s N=%d" % ("[bf16 P/dP]" if bf16 else "", "[linked]" if linked else "", N)
    if dead:
        for i in (0, 1, 2):
            assert float(grads[i].float().abs().max()) == 0.0, ("MfbFuseFn: a zero cotangent must give exact zeros", i)
        rep.append("%s: Y %.1e, dY == 0 -> dP, dbias, dq exactly 0" % (tag, y_err))
        return
    floor = FUSE_FLOOR_BF16 if bf16 else FUSE_FLOOR_F32
    worst = _cond_check("MfbFuseFn dP", grads[0].float(), res[torch.float64][1], res[torch.float32][1], floor=5e-3 if bf16 else floor)
    worst = max(worst, _cond_check("MfbFuseFn dbias", grads[1], res[torch.float64][2], res[torch.float32][2], floor=floor))
    worst = max(worst, _cond_check("MfbFuseFn dq", grads[2], res[torch.float64][3], res[torch.float32][3], floor=floor))
    rep.append("%s: Y %.1e, grads err/bound %.2f" % (tag, y_err, worst))


def _check_final_mfb(rec, out, grads, rep, saved):
    qa, va, wq, bq, wv, bv, keep, seed, p_drop, cascade, want_zdrop, bf16 = (list(rec["args"]) + [None, False, False])[:12]
    assert keep is None and p_drop == 0.0 and cascade is None and not want_zdrop
    N = qa.shape[0]
    qq_k, vv_k = saved[4], saved[5]               # saved: (qa_s, va_s, wq, wv, qq, vv, y, norm, inv, keep, cascade, wqb, wvb)
    bf16 = bool(bf16) and all(d % 8 == 0 for d in (N, qa.shape[1], va.shape[1], wq.shape[0]))
    rnd = bf if bf16 else (lambda t: t)
    qa_o, va_o, wq_o, wv_o = rnd(qa.detach()), rnd(va.detach()), rnd(wq.detach()), rnd(wv.detach())
    e_q = _nrel(qq_k, qa_o.double() @ wq_o.double().t() + bq.detach().double())
    e_v = _nrel(vv_k, va_o.double() @ wv_o.double().t() + bv.detach().double())
    assert e_q <= (BENIGN_TOL_BF16 if bf16 else gemm_tol(qa.shape[1])), ("FinalMfbFn question projection", e_q)
    assert e_v <= (BENIGN_TOL_BF16 if bf16 else gemm_tol(va.shape[1])), ("FinalMfbFn image projection", e_v)
    res = {}
    for dt in (torch.float64, torch.float32):
        ql, vl = qq_k.to(dt).requires_grad_(True), vv_k.to(dt).requires_grad_(True)
        y = ref_fuse(vl, ql, N, 1)
        dvv, dqq = torch.autograd.grad(y, [vl, ql], rec["dout"].to(dt))
        dqo, dvo = rnd(dqq).double(), rnd(dvv).double()
        res[dt] = (y.detach(), dqo @ wq_o.double(), dvo @ wv_o.double(), dqo.t() @ qa_o.double(), dqq.double().sum(0),
                   dvo.t() @ va_o.double(), dvv.double().sum(0))
    y_err = _nrel(out, res[torch.float64][0])
    assert y_err <= FUSE_OUT_TOL, ("FinalMfbFn y", y_err)
    worst = 0.0
    floor = FUSE_FLOOR_BF16 if bf16 else FUSE_FLOOR_F32
    for i in range(6):
        if i in grads and grads[i] is not None:
            worst = max(worst, _cond_check("FinalMfbFn grad of arg %d" % i, grads[i], res[torch.float64][1 + i], res[torch.float32][1 + i],
                                           floor=floor))
    rep.append("FinalMfbFn%s: proj %.1e, y %.1e, grads err/bound %.2f" % ("[bf16]" if bf16 else "", max(e_q, e_v), y_err, worst))


def _check_lstm_seq(rec, out, grads, rep):
    x, w_ih, w_hh, b_ih, b_hh, bf16 = (list(rec["args"]) + [False])[:6]
    dt = torch.float64
    leaves = [t.detach().to(dt).requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
    batch_form = rec["cls"].__name__ == "LstmBatchFn"
    if batch_form and bf16 and w_hh.shape[1] % 8:
        bf16 = False                                           # functions.LstmBatchFn: bf16 needs H % 8 == 0
    if batch_form and bf16 == "all" and x.shape[1] % 8:
        wg = False                                             # ... and its bf16 recurrent weight gradient B % 8 == 0
    else:
        wg = (bf16 == "all") if batch_form else True
    hs = ref_lstm_seq(*leaves, bf16, wg)
    want = [i for i in range(5) if i in grads and grads[i] is not None]
    g64 = torch.autograd.grad(hs, [leaves[i] for i in want], rec["dout"].double())
    tol = BENIGN_TOL_BF16 if bf16 else (LSTM_BATCH_TOL_F32 if batch_form else LSTM_SEQ_TOL_F32)
    worst = _nrel(out, hs)
    assert worst <= tol, ("%s hs" % rec["cls"].__name__, worst)
    for i, g in zip(want, g64):
        e = _nrel(grads[i], g)
        assert e <= tol, ("%s grad of arg %d (bf16=%r)" % (rec["cls"].__name__, i, bf16), e)
        worst = max(worst, e)
    rep.append("%s[bf16=%r] S=%d: %.1e (bound %.0e)" % (rec["cls"].__name__, bf16, x.shape[0], worst, tol))


def _check_embed_tanh(rec, out, grads, rep):
    ids, weight, tanh, time_major = (list(rec["args"]) + [True, False])[:4]
    w = weight.detach().double().requires_grad_(True)
    y = w[ids]
    if tanh:
        y = torch.tanh(y)
    if time_major:
        y = y.transpose(0, 1)
    (dw,) = torch.autograd.grad(y, w, rec["dout"].double())
    e = max(_nrel(out, y), _nrel(grads[1], dw))
    assert e <= 2e-6, ("EmbedTanhFn", e)
    rep.append("EmbedTanhFn: %.1e" % e)


def _check_dropout_bt(rec, out, grads, rep):
    x, keep, seed, p = rec["args"][:4]
    assert keep is None and p == 0.0
    assert torch.equal(out, x.detach()) and torch.equal(grads[0], rec["dout"]), "DropoutBTFn at rate 0 is a re-layout"
    rep.append("DropoutBTFn[p=0]: exact")


def _check_hie_core(rec, out, grads, rep, saved):
    args = rec["args"]
    drops = args[15]
    assert all(k is None and p == 0.0 for (k, s, p) in drops.values()), "node check runs with the functional dropouts at rate 0"
    fidx = [i for i in range(15) if torch.is_tensor(args[i]) and args[i].is_floating_point()]
    mask = (saved[2] > 0).view(args[0].shape[0], args[0].shape[1], -1)      # saved[2]: img = relu(img_emb(.)) as the kernel stored it
    live = [j for j, d in enumerate(rec["douts"]) if d is not None]
    res = {}
    for dt in (torch.float64, torch.float32):            # fp32: what a torch evaluation of the same node is from fp64 (printed)
        leaves = {i: args[i].detach().to(dt).requires_grad_(args[i].requires_grad) for i in fidx}
        ref = ref_hie_core(leaves[0], args[1], *[leaves[i] for i in range(2, 15)], mask=mask)
        want = [i for i in fidx if leaves[i].requires_grad and i in grads and grads[i] is not None]
        g = torch.autograd.grad([ref[j] for j in live], [leaves[i] for i in want],
                                [rec["douts"][j].to(dt).reshape(ref[j].shape) for j in live])
        res[dt] = ([r.detach() for r in ref], dict(zip(want, g)))
        del leaves, ref
    ref, g64 = res[torch.float64]
    worst = 0.0
    for j, (o, r) in enumerate(zip(out, ref)):
        e = _nrel(o.reshape(r.shape), r)
        assert e <= HIE_TOL_F32, ("HieCoreFn output %d" % j, e)
        worst = max(worst, e)
    gmax = max(float(g.norm()) for g in g64.values())
    names = {2: "img_emb.w", 3: "img_emb.b", 4: "que_emb", 5: "Wbv.w", 6: "Wbv.b", 7: "Wv.w", 8: "Wv.b", 9: "Wq.w", 10: "Wq.b",
             11: "Whv.w", 12: "Whv.b", 13: "Whq.w", 14: "Whq.b"}
    detail = []
    for i, g in g64.items():
        # the biases in front of a softmax (fc_Whv / fc_Whq bias) carry a mathematically-zero gradient: absolute scale
        err = float((grads[i].double().reshape(g.shape) - g).norm())
        noise = float((res[torch.float32][1][i].double() - g).norm())
        assert err <= HIE_TOL_F32 * float(g.norm()) + 1e-7 * gmax, ("HieCoreFn grad of arg %d (%s)" % (i, names.get(i)), err, float(g.norm()))
        assert err <= max(4.0 * noise, 2e-5 * float(g.norm())) + 1e-7 * gmax, ("HieCoreFn grad of arg %d (%s) vs the fp32 noise of the node"
                                                                             % (i, names.get(i)), err, noise, float(g.norm()))
        if float(g.norm()) > 1e-6 * gmax:
            worst = max(worst, err / float(g.norm()))
            detail.append("%s %.1e/%.1e" % (names.get(i, i), err / float(g.norm()), noise / float(g.norm())))
    rep.append("HieCoreFn N=%d: %.1e (bound %.0e); per tensor err / torch-fp32 noise: %s" % (args[0].shape[0], worst, HIE_TOL_F32, ", ".join(detail)))


def _to_base(g, a, base):
    """gradient w.r.t. a view `a` expressed in the layout of its base (views met here: same-shape, reshape, transpose(0,1))"""
    if a.shape == base.shape:
        return g
    if a.dim() == 3 and base.dim() == 3 and a.transpose(0, 1).shape == base.shape and a.stride(0) == base.stride(1) \
            and a.stride(1) == base.stride(0):
        return g.transpose(0, 1)
    return g.reshape(base.shape)


def check_every_node(model, recd, label, min_links=3, skip_params=("word_embedding.weight",)):
    """Replays every recorded node of the step that `model` has just run (forward + backward) alone and checks wiring and
    numerics as the module docstring says.  -> (report line, names of the parameters whose gradient a checked node produced)."""
    params = {id(p): (k, p) for k, p in model.named_parameters()}
    produced = {}
    for r in recd.records:
        for o in (r["out"] if isinstance(r["out"], tuple) else (r["out"],)):
            if torch.is_tensor(o):
                produced[id(o)] = r
    into = {}                                    # id(intermediate tensor) -> list of node-local gradients handed to it
    seen_params, rep = set(), []
    # a fusion node with a NormLink and the attention head that consumes its output share the link object: the head's backward
    # leaves (dlogits, lin) in it for the fusion node's backward.  Replayed alone, the head goes first and the fusion node gets
    # that pair back between its forward (which re-arms the link) and its backward.
    cache = {}
    fns = None
    for rec in recd.records:
        name = rec["cls"].__name__
        link = (rec["args"][8] if len(rec["args"]) > 8 else None) if name in ("ImgFuseFn", "MfbFuseFn") else None
        if link is not None:
            head = [r for r in recd.records if r["cls"].__name__ == "AttHeadFn" and len(r["args"]) > 10 and r["args"][10] is link]
            assert len(head) == 1
            cache[id(head[0])] = recd.replay(head[0])
            lin = link.lin
            assert lin is not None
            out2, grads, saved = recd.replay(rec, between=lambda o: setattr(link, "lin", lin))
        elif id(rec) in cache:
            out2, grads, saved = cache.pop(id(rec))
        elif name == "ImgProjDeferFn":
            # the node's forward only allocates P0; the model fills it behind the question encoder (mfb._SideStream.join)
            out2, grads, saved = recd.replay(rec, between=lambda o, rec=rec: rec["cls"].fill(o, rec["args"][0], rec["args"][1]))
        else:
            out2, grads, saved = recd.replay(rec)
        torch.cuda.synchronize()
        outs2 = out2 if isinstance(out2, tuple) else (out2,)
        outs = rec["out"] if isinstance(rec["out"], tuple) else (rec["out"],)
        for o2, o in zip(outs2, outs):
            assert torch.equal(o2, o), (name, "a node replayed alone is not bit-identical")
        for i, g in grads.items():
            a = rec["args"][i]
            if id(a) in params and g is not None:
                k, p = params[id(a)]
                assert torch.equal(g, p.grad), (k, "model-level gradient is not this node's gradient, bit for bit")
                seen_params.add(k)
            elif g is not None:
                # a consumer may take a VIEW of a producer's output (AttHeadFn gets hs and hs.view(N*T, H)): credit the base
                base = a._base if (a._base is not None and id(a._base) in produced) else a
                into.setdefault(id(base), []).append(_to_base(g, a, base))
        if name == "LinearFn":
            _check_linear(rec, out2, grads, rep)
        elif name == "Linear2Fn":
            _check_linear2(rec, out2, grads, rep)
        elif name == "LogSoftmaxRowsFn":
            _check_logsoftmax(rec, out2, grads, rep)
        elif name == "AttHeadFn":
            _check_att_head(rec, out2, grads, rep, saved)
        elif name == "ImgFuseFn":
            _check_img_fuse(rec, out2, grads, rep, saved)
        elif name in ("ImgProjLateFn", "ImgProjDeferFn", "ImgProjFn"):
            _check_img_proj(rec, out2, grads, rep)
        elif name == "MfbFuseFn":
            _check_mfb_fuse(rec, out2, grads, rep)
        elif name == "FinalMfbFn":
            _check_final_mfb(rec, out2, grads, rep, saved)
        elif name in ("LstmSeqFn", "LstmBatchFn"):
            _check_lstm_seq(rec, out2, grads, rep)
        elif name == "EmbedTanhFn":
            _check_embed_tanh(rec, out2, grads, rep)
        elif name == "DropoutBTFn":
            _check_dropout_bt(rec, out2, grads, rep)
        elif name == "HieCoreFn":
            _check_hie_core(rec, out2, grads, rep, saved)
        else:
            raise AssertionError("no node check for %s" % name)
        del out2, grads, saved
        torch.cuda.empty_cache()
    # every parameter with a gradient was produced by exactly one checked node (skip_params: produced by a torch op upstream)
    expect = {k for k, p in model.named_parameters() if p.grad is not None} - set(skip_params)
    assert seen_params == expect, (sorted(expect - seen_params), sorted(seen_params - expect))
    # chain wiring: what the consumers hand to an intermediate adds up to the gradient recorded at its producer
    n_links = 0
    for tid, gs in into.items():
        if tid in produced:
            r = produced[tid]
            outs = r["out"] if isinstance(r["out"], tuple) else (r["out"],)
            j = [id(o) for o in outs].index(tid)
            tot = gs[0].double()
            for g in gs[1:]:
                tot = tot + g.double()
            e = _nrel(tot, r["douts"][j])
            assert e <= 1e-6, (r["cls"].__name__, "consumers' gradients do not add up to the producer's", e)
            n_links += 1
    assert n_links >= min_links                  # qa (3 consumers), qp, Y, va (2), logits
    line = label + ": " + " | ".join(rep)
    print(line)
    root = os.environ.get("GRAFT_REPO_ROOT")
    if root and os.path.isdir(os.path.join(root, "gpurun_out")):
        with open(os.path.join(root, "gpurun_out", "grad_parity.log"), "a") as f:
            f.write(line + "\n")
    return line, seen_params


ALL_NODES = ["EmbedTanhFn", "LstmSeqFn", "LstmBatchFn", "DropoutBTFn", "AttHeadFn", "LinearFn", "Linear2Fn", "ImgFuseFn", "ImgProjFn",
             "ImgProjLateFn", "ImgProjDeferFn", "MfbFuseFn", "FinalMfbFn", "LogSoftmaxRowsFn", "HieCoreFn"]
