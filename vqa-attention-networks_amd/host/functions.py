"""torch.autograd.Function per fused stage of the hot path.

Each Function's forward/backward is a short sequence of C-ABI launches on the
current HIP stream; no torch arithmetic op touches the big tensors.

  LinearFn     a4, a10      nn.Linear / 1x1 conv          (mfb.py:92,137 ...)
  AttHeadFn    a3, a7+a8    attention MLP -> 2 logits -> softmax -> glimpse sums
  ImgFuseFn    a5+a6        image projection + MFB fusion over the 196 regions
  FinalMfbFn   a9           the (N,5000) x (N,5000) MFB block(s)
"""
import torch

from . import ops


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _w2d(w):
    """(out,in,1,1) conv weight or (out,in) linear weight -> contiguous (out,in) view."""
    return _c(w.reshape(w.shape[0], -1))


def _bf16_ok(*dims):
    """bf16 GEMM entry point: every extent that becomes a K or a K-major row extent must be a multiple of 8."""
    return all(d % 8 == 0 for d in dims)


class EmbedTanhFn(torch.autograd.Function):
    """que_embedded = tanh(word_embedding(questions))   (mfb.py:68, mhb_coAtt.py:69): gather + tanh in one launch, and a
    deterministic one-launch weight gradient (csrc/embed.hip) instead of torch's sort-based embedding backward."""

    @staticmethod
    def forward(ctx, ids, weight, tanh=True, time_major=False):
        ids = ids.contiguous()
        out = ops.embed_tanh_fwd(_c(weight), ids, tanh, time_major)
        ctx.save_for_backward(ids, out if tanh else None)
        ctx.V, ctx.tm = weight.shape[0], bool(time_major)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, out = ctx.saved_tensors
        return None, ops.embed_tanh_bwd(_c(dout), out, ids, ctx.V, ctx.tm), None, None


def _plain_embedding(embedding, ids):
    w = embedding.weight
    return (w.is_cuda and w.dtype == torch.float32 and ids.is_cuda and ids.dtype == torch.int64 and embedding.padding_idx is None
            and embedding.max_norm is None and not embedding.sparse and not embedding.scale_grad_by_freq and w.shape[1] <= 1024)


def embed_tanh(embedding, ids, time_major=False):
    """tanh(embedding(ids)) on the HIP path when the nn.Embedding is a plain lookup (no padding_idx / max_norm / sparse grads /
    frequency scaling) with an fp32 GPU weight of width <= 1024; torch otherwise.  time_major: ids (N,Tq) -> (Tq,N,E), the layout
    the batch-major LSTM consumes (the lookup writes it directly: no transposing copy)."""
    if _plain_embedding(embedding, ids):
        return EmbedTanhFn.apply(ids, embedding.weight, True, bool(time_major) and ids.dim() == 2)
    y = torch.tanh(embedding(ids))
    return y.transpose(0, 1).contiguous() if time_major and ids.dim() == 2 else y


def embed(embedding, ids):
    """embedding(ids) (hieCoAtten.py:27, networks.py:23,56, mhb_coAtt.py:181) on the HIP path under the same conditions."""
    if _plain_embedding(embedding, ids):
        return EmbedTanhFn.apply(ids, embedding.weight, False, False)
    return embedding(ids)


class NormLink:
    """F.normalize folded into its consumer (mfb.py:105-109: fusion_normed is consumed by co_att_conv1 alone).

    The producer (ImgFuseFn / MfbFuseFn) hands on R, the signed square roots WITHOUT the per-sample 1/norm, and leaves
    inv = 1 / max(||R_n||, eps) and L here; the consumer (the co-attention AttHeadFn) applies inv in the epilogue of its
    conv GEMM (vqf_gemm_f32_rowscale), scales its stored hidden-layer gradient by it so that dW1 and the gradient it
    returns (dYs = dY / norm) need no further pass, and leaves (dlogits, lin) here, from which the producer's backward gets
    sum(Y * dY) per sample.  Saves the scale_rows pass (forward) and the rowdot pass (backward) over the (N*L, 1000)
    tensors.  One link per forward call; only valid while the producer's output has exactly this one consumer."""
    __slots__ = ("inv", "L", "lin", "xb")

    def __init__(self):
        self.inv, self.L, self.lin, self.xb = None, 0, None, None    # xb: the producer's bf16 copy of R (K padded to 32), bf16 modes


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b  (optionally relu).  x (M,K), W (N,K).
    bf16 (gemm_dtype "bf16-all"): bf16 operands / fp32 accumulate in the forward, the dgrad and the weight gradient
    (operands cast once per use by vqf_cast_f32_bf16; bias, ReLU and every reduction stay fp32)."""

    @staticmethod
    def forward(ctx, x, w, b, relu=False, bf16=False):
        x = _c(x)
        w2 = _w2d(w)
        bf16 = bool(bf16) and _bf16_ok(x.shape[0], x.shape[1], w2.shape[0])
        if bf16:
            xb, wb = ops.cast_bf16(x), ops.cast_bf16(w2)
            y = ops.gemm_bf16(xb, wb, bias=b, relu=relu)
            ctx.save_for_backward(xb, w, y if relu else None, wb)
        else:
            y = ops.gemm(x, w2, bias=b, relu=relu)
            ctx.save_for_backward(x, w, y if relu else None, None)
        ctx.has_bias = b is not None
        ctx.relu = relu
        ctx.bf16 = bf16
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, wb = ctx.saved_tensors
        w2 = _w2d(w)
        dy = _c(dy)
        db = None
        if ctx.relu:
            dy, db = ops.relu_bwd(dy, y, want_bias=ctx.has_bias)
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dy)
        dx = dw = None
        if ctx.bf16:
            dyb = ops.cast_bf16(dy)
            if ctx.needs_input_grad[0]:
                dx = ops.gemm_bf16(dyb, wb, tb=True)                         # dX = dY W      (M,K)
            if ctx.needs_input_grad[1]:
                dw = ops.gemm_bf16(dyb, x, ta=True, tb=True).view_as(w)      # dW = dY^T X    (N,K)
            return dx, dw, db, None, None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dy, w2, tb=True)                       # dX = dY W      (M,K)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm(dy, x, ta=True, tb=True).view_as(w)    # dW = dY^T X    (N,K)
        return dx, dw, db, None, None


class Linear2Fn(torch.autograd.Function):
    """y = [x1 | x2] @ W^T + b without materialising the concatenation (mhb_coAtt.py:147-148,213-214: the two final MFB blocks
    are concatenated along the feature axis and fed to the classifier).  x1 (M,K1), x2 (M,K2), W (N, K1+K2): two products into
    one output (the second accumulates), the weight's column blocks read in place as row-strided operands; the backward hands
    each block its own contiguous gradient and writes the weight gradient's column blocks in place."""

    @staticmethod
    def forward(ctx, x1, x2, w, b):
        x1, x2 = _c(x1), _c(x2)
        w2 = _w2d(w)
        K1 = x1.shape[1]
        y = ops.gemm(x1, w2[:, :K1], bias=b)
        ops.gemm(x2, w2[:, K1:], out=y, accumulate=True)
        ctx.save_for_backward(x1, x2, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, w = ctx.saved_tensors
        w2 = _w2d(w)
        dy = _c(dy)
        K1 = x1.shape[1]
        dx1 = ops.gemm(dy, w2[:, :K1], tb=True) if ctx.needs_input_grad[0] else None
        dx2 = ops.gemm(dy, w2[:, K1:], tb=True) if ctx.needs_input_grad[1] else None
        dw = None
        if ctx.needs_input_grad[2]:
            dw = torch.empty_like(w2)
            ops.gemm(dy, x1, ta=True, tb=True, out=dw[:, :K1])
            ops.gemm(dy, x2, ta=True, tb=True, out=dw[:, K1:])
            dw = dw.view_as(w)
        db = ops.colsum(dy) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        return dx1, dx2, dw, db


class JoinRowsFn(torch.autograd.Function):
    """cat((v, q), 0) where v and q already ARE the two row blocks of `buf` (their producers wrote them there): returns buf,
    hands each block its contiguous half of the gradient (hieCoAtten.py:52)."""

    @staticmethod
    def forward(ctx, v, q, buf):
        n = v.shape[0]
        if v.data_ptr() != buf.data_ptr() or q.data_ptr() != buf[n:].data_ptr() or buf.shape[0] != n + q.shape[0]:
            raise ops._l.VqfError("JoinRowsFn: v and q must be the row blocks of buf")
        ctx.n = n
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        return g[:ctx.n], g[ctx.n:], None


class AttHeadFn(torch.autograd.Function):
    """Attention MLP + glimpse pooling.

    x    (N*S, Cin)  MLP input  (question side: the LSTM states; image side: fusion_normed)
    feat (N, S, C)   what the glimpses pool (LSTM states / the image tensor)
    w1,b1 [wm,bm] w2,b2: conv1 (+ "multilayer" conv) + conv2 -> 2 logits
    unit_softmax: reproduce mfb.py:84,118 (softmax over the singleton axis == 1)
    same_src: the CALLER's statement that x is feat.view(N*S, C) of the same autograd tensor (the question head pools the very
        tensor its MLP reads, mfb.py:73-89): the backward then adds the MLP's input gradient onto the pooling's in the GEMM
        epilogue and returns ONE gradient (for feat; None for x) instead of leaving two for autograd to add.  Never inferred
        from pointers: an alias with a different autograd history must keep its own gradient.
    returns pooled (N, 2C); the attention weights (N,2,S) are kept on ctx.
    """

    @staticmethod
    def forward(ctx, x, feat, w1, b1, wm, bm, w2, b2, unit_softmax, bf16=False, link=None, same_src=False):
        x = _c(x)
        feat = _c(feat)
        ctx.bf16 = bool(bf16)
        ctx.same_src = bool(same_src)
        if ctx.same_src and (x.numel() != feat.numel() or x.shape[-1] != feat.shape[-1] or x.dtype != feat.dtype):
            raise ops._l.VqfError("AttHeadFn: same_src needs x = feat.view(N*S, C)")
        ctx.link = link if (link is not None and link.inv is not None) else None
        if ctx.link is not None:
            # x is the UN-NORMALISED fusion output: 1/norm of the sample goes into the conv GEMM's epilogue, and the logit
            # kernel also returns the part of each logit that is linear in x (NormLink)
            if wm is not None or b1 is None:
                raise ops._l.VqfError("AttHeadFn: a NormLink needs the single-hidden-layer head with a bias")
            ctx.bf16 = ctx.bf16 and _bf16_ok(_w2d(w1).shape[0], _w2d(w1).shape[1])
            if ctx.bf16:       # bf16 operands (BASELINE config 3): the un-normalised R is cast, 1/norm rides in the bf16 GEMM's epilogue
                w1b = ops.cast_bf16(_w2d(w1), 32)
                xb = link.xb if (link.xb is not None and link.xb.shape == (x.shape[0], w1b.shape[1])) else ops.cast_bf16(x, 32)
                link.xb = None
                hid1 = ops.gemm_bf16_rowscale(xb, w1b, link.inv, link.L, bias=b1, relu=True, K=xb.shape[1])
                x = xb
            else:
                hid1 = ops.gemm_rowscale(x, _w2d(w1), link.inv, link.L, bias=b1, relu=True)
            logits, lin = ops.att_logits_fwd_lin(hid1, _w2d(w2), b2, b1)
            wts, pooled = ops.glimpse_pool_fwd(feat, logits, unit_softmax)
            # (bf16 mode: slot 3 -- no "multilayer" conv with a NormLink -- carries the bf16 copy of w1 to the backward: one cast per step)
            ctx.save_for_backward(x, feat, w1, w1b if ctx.bf16 else None, w2, hid1, None, wts, lin)
            ctx.unit = bool(unit_softmax)
            return pooled
        # (the bf16 GEMM entry point wants K and every K-major operand's row extent to be multiples of 8: here the hidden
        #  width and, for the dgrad's N = C_in columns, the input width; else the head stays fp32, like LinearFn / FinalMfbFn)
        ctx.bf16 = ctx.bf16 and _bf16_ok(_w2d(w1).shape[0], _w2d(w1).shape[1])
        if ctx.bf16:
            # bf16 operands, fp32 accumulate (BASELINE config 3); K padded to a multiple of 32
            xb, w1b = ops.cast_bf16(x, 32), ops.cast_bf16(_w2d(w1), 32)
            hid1 = ops.gemm_bf16(xb, w1b, K=xb.shape[1], bias=b1, relu=True)
            x = xb                                      # the backward only needs the bf16 copy
        else:
            hid1 = ops.gemm(x, _w2d(w1), bias=b1, relu=True)
        hid2 = ops.gemm(hid1, _w2d(wm), bias=bm, relu=True) if wm is not None else None
        last = hid2 if hid2 is not None else hid1
        logits = ops.att_logits_fwd(last, _w2d(w2), b2)
        wts, pooled = ops.glimpse_pool_fwd(feat, logits, unit_softmax)
        ctx.save_for_backward(x, feat, w1, wm, w2, hid1, hid2, wts, w1b if ctx.bf16 else None)      # (last slot: the bf16 copy of w1)
        ctx.unit = bool(unit_softmax)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x, feat, w1, wm, w2, hid1, hid2, wts, lin = ctx.saved_tensors
        dpooled = _c(dpooled)
        need_dfeat = ctx.needs_input_grad[1]
        dlogits, dfeat = ops.glimpse_pool_bwd(dpooled, feat, wts, ctx.unit, need_dfeat)
        if ctx.link is not None:
            link = ctx.link
            w1b = wm                                       # (slot 3, see forward)
            obf = ctx.bf16 and hid1.shape[1] % 8 == 0 and w2.shape[0] == 2          # the kernel stores bf16 rows itself: no cast launch
            d1s, dw2, db2, db1 = ops.att_logits_bwd(dlogits, hid1, _w2d(w2), relu_mask=True, rowscale=link.inv,
                                                    rows_per_scale=link.L, out_bf16=obf)   # stored rows already times 1/norm
            link.lin = (dlogits, lin)                                               # -> sum(Y * dY) in the producer's backward
            if ctx.bf16:
                cin = _w2d(w1).shape[1]
                d1b = d1s if obf else ops.cast_bf16(d1s)
                dw1 = ops.gemm_bf16(d1b, x, ta=True, tb=True)[:, :cin].contiguous().view_as(w1)
                dx = ops.gemm_bf16(d1b, w1b, tb=True, N=cin) if ctx.needs_input_grad[0] else None
                return dx, dfeat, dw1, db1, None, None, dw2.view_as(w2), db2, None, None, None, None
            dw1 = ops.gemm(d1s, x, ta=True, tb=True).view_as(w1)                    # = dpre^T Y
            dx = ops.gemm(d1s, _w2d(w1), tb=True) if ctx.needs_input_grad[0] else None   # dYs = dY / norm
            return dx, dfeat, dw1, db1, None, None, dw2.view_as(w2), db2, None, None, None, None
        last = hid2 if hid2 is not None else hid1
        dlast_pre, dw2, db2, dblast = ops.att_logits_bwd(dlogits, last, _w2d(w2), relu_mask=True)
        dwm = dbm = None
        if hid2 is not None:
            dwm = ops.gemm(dlast_pre, hid1, ta=True, tb=True).view_as(wm)
            dbm = dblast
            dhid1 = ops.gemm(dlast_pre, _w2d(wm), tb=True)
            d1_pre, db1 = ops.relu_bwd(dhid1, hid1, want_bias=True)
        else:
            d1_pre, db1 = dlast_pre, dblast
        if ctx.bf16:
            cin = _w2d(w1).shape[1]
            d1b = ops.cast_bf16(d1_pre)
            dw1 = ops.gemm_bf16(d1b, x, ta=True, tb=True)[:, :cin].contiguous().view_as(w1)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = ops.gemm_bf16(d1b, lin, tb=True, N=cin)                  # lin: slot 8 = w1's bf16 copy in this form
        else:
            dw1 = ops.gemm(d1_pre, x, ta=True, tb=True).view_as(w1)
            if ctx.same_src and dfeat is not None:
                ops.gemm(d1_pre, _w2d(w1), tb=True, out=dfeat.view(x.shape), accumulate=True)     # dfeat += dx: one gradient for the shared source
                dx = None
            else:
                dx = ops.gemm(d1_pre, _w2d(w1), tb=True) if ctx.needs_input_grad[0] else None
        return dx, dfeat, dw1, db1, dwm, dbm, dw2.view_as(w2), db2, None, None, None, None


def _arm_link(link, inv, L, xb=None):
    """producer side of a NormLink: publish 1/norm and the row-group size for the consumer's GEMM epilogue (and, in the bf16 modes,
    the bf16 copy of R the fusion kernel wrote beside the fp32 one: the consumer's GEMM operand without a cast pass)"""
    if link is not None:
        link.inv, link.L, link.lin, link.xb = inv, L, None, (xb[0] if xb else None)
    return link


def _take_lin(link):
    """producer's backward: the consumer's (dlogits, lin), exactly once"""
    if link is None:
        return None
    if link.lin is None:
        raise ops._l.VqfError("NormLink: the fusion output's consumer has not run its backward (the un-normalised output "
                              "of ImgFuseFn / MfbFuseFn may only feed the co-attention AttHeadFn it was linked to)")
    lin, link.lin = link.lin, None
    return lin


class ImgFuseFn(torch.autograd.Function):
    """a5+a6: P = img W^T + b;  Y = L2norm_n(ssqrt(pool5(dropout(P * q[n])))).

    img (N,L,D) data (no gradient), wi (5000,D[,1,1]), bi (5000), q (N,5000)
    -> Y (N*L, 1000), row m = n*L + l   (the reference's fusion_normed is the
    (N,1000,L,1) permutation of the same values, mfb.py:103-106).
    """

    BF16_P = True      # bf16 mode: store P in bf16 when the large-tile GEMM applies (A/B switch)

    @staticmethod
    def forward(ctx, img, wi, bi, q, keep, seed, p_drop, bf16=False, link=None):
        img = _c(img)
        q = _c(q)
        N, L, D = img.shape
        wi2 = _w2d(wi)
        O = wi2.shape[0] // ops.POOL_K
        ctx.bf16 = bool(bf16)
        if ctx.bf16:
            # bf16 storage of the image tensor and the projection weight, fp32 accumulation
            img = img.view(N * L, D) if img.dtype == torch.bfloat16 else ops.cast_bf16(img.view(N * L, D))
            wb = ops.cast_bf16(wi2)
            # the projection is STORED in bf16 when the large-tile kernel applies (half the bytes of the largest
            # tensor of the step in the GEMM epilogue, the fusion forward and the fusion backward)
            P = ops.gemm_bf16(img, wb, bias=bi, out_bf16=True) if ImgFuseFn.BF16_P else None
            if P is None:
                P = ops.gemm_bf16(img, wb, bias=bi)
        else:
            P = ops.gemm(img.view(N * L, D), wi2, bias=bi)
        rb = [] if (link is not None and P.dtype == torch.bfloat16) else None
        Y, norm, inv, _ = ops.mfb_fuse_fwd(P, q, N, L, O, keep=keep, seed=seed, p_drop=p_drop, normalise=link is None, r_bf16=rb)
        ctx.link = _arm_link(link, inv, L, rb)
        ctx.save_for_backward(img, wi, q, P, Y, norm, inv, keep)
        ctx.seed, ctx.p_drop, ctx.dims = seed, p_drop, (N, L, D, O)
        return Y

    @staticmethod
    def backward(ctx, dY):
        img, wi, q, P, Y, norm, inv, keep = ctx.saved_tensors
        N, L, D, O = ctx.dims
        dP, dq, _, dbi = ops.mfb_fuse_bwd(_c(dY), Y, norm, inv, P, q, N, L, O, keep=keep, seed=ctx.seed,
                                          p_drop=ctx.p_drop, want_dbias=True, dp_bf16=ctx.bf16, lin=_take_lin(ctx.link))
        if ctx.bf16:                  # dP already is the bf16 A operand of the weight-gradient GEMM
            dwi = ops.gemm_bf16(dP, img, ta=True, tb=True).view_as(wi)
        else:
            dwi = ops.gemm(dP, img.view(N * L, D), ta=True, tb=True).view_as(wi)   # wgrad, K = N*L
        return None, dwi, dbi, dq, None, None, None, None, None


class ImgProjFn(torch.autograd.Function):
    """a5 alone: P0 = img W^T (NO bias: the bias is added inside the fusion kernel).  Split from the
    fusion so that the module can run this GEMM -- 86 % of the forward FLOPs, independent of the
    question path -- on a side stream; autograd then runs its weight-gradient GEMM on that same
    stream, concurrently with the question-side backward and the gradient all-reduce."""

    @staticmethod
    def forward(ctx, img, wi, bf16=False):
        img = _c(img)
        N, L, D = img.shape
        wi2 = _w2d(wi)
        ctx.bf16 = bool(bf16)
        if ctx.bf16:
            # bf16 feature storage (FeatureStager(bf16=True)): the batch is used as it is, no per-step cast
            img2 = img.view(N * L, D) if img.dtype == torch.bfloat16 else ops.cast_bf16(img.view(N * L, D))
            P0 = ops.gemm_bf16(img2, ops.cast_bf16(wi2))
        else:
            img2 = img.view(N * L, D)
            P0 = ops.gemm(img2, wi2)
        ctx.save_for_backward(img2, wi)
        return P0

    @staticmethod
    def backward(ctx, dP):
        img2, wi = ctx.saved_tensors
        dP = _c(dP)
        if ctx.bf16:
            dwi = ops.gemm_bf16(ops.cast_bf16(dP), img2, ta=True, tb=True).view_as(wi)
        else:
            dwi = ops.gemm(dP, img2, ta=True, tb=True).view_as(wi)            # wgrad, K = N*L
        return None, dwi, None


class ImgProjDeferFn(torch.autograd.Function):
    """fp32 image projection whose NODE exists before its PRODUCT: forward() only allocates P0 and returns it; fill() issues the
    GEMM into that buffer later (a raw write the autograd graph does not see).  The one-stream form wants the node created
    FIRST (a node created first runs its backward last: the 14 ms weight gradient then hides the other gradients' all-reduce)
    but the product issued AFTER the question encoder: the projection is then not the kernel right behind the previous step's
    weight-gradient GEMM and Adam -- two back-to-back 14 ms MFMA-bound launches made the second one run 1.4 % slower (same box:
    14.67 vs 14.47 ms, `profiles/r03_all_configs.log`; forward-only steps reach 14.23)."""

    @staticmethod
    def forward(ctx, img, wi):
        img = _c(img)
        N, L, D = img.shape
        ctx.save_for_backward(img.view(N * L, D), wi)
        return torch.empty((N * L, wi.shape[0]), dtype=torch.float32, device=img.device)

    @staticmethod
    def fill(P0, img, wi):
        img = _c(img)
        ops.gemm(img.view(-1, img.shape[-1]), _w2d(wi), out=P0.detach())

    @staticmethod
    def backward(ctx, dP):
        img2, wi = ctx.saved_tensors
        return None, ops.gemm(_c(dP), img2, ta=True, tb=True).view_as(wi)            # wgrad, K = N*L


def img_project(img, wi, bf16, cu_limit=0):
    """a5 without an autograd node: P0 = img W^T (no bias), fp32 or -- bf16 operands -- STORED in bf16 when the large-tile
    kernel applies.  Returns (P0, img2d as the GEMM consumed it).  cu_limit > 0: the persistent large-tile GEMM leaves
    the other CUs to kernels of other streams (library option gemm_cu_limit)."""
    img = _c(img)
    N, L, D = img.shape
    wi2 = _w2d(wi)
    # (no limit asked for: the option is left alone, so a limit the user set with ops.set_option / VQF_GEMM_CU_LIMIT applies)
    with (ops.options(gemm_cu_limit=cu_limit) if cu_limit else ops.options()):
        if bf16:
            img2 = img.view(N * L, D) if img.dtype == torch.bfloat16 else ops.cast_bf16(img.view(N * L, D))
            wb = ops.cast_bf16(wi2)
            P0 = ops.gemm_bf16(img2, wb, out_bf16=True) if ImgFuseFn.BF16_P else None
            if P0 is None:
                P0 = ops.gemm_bf16(img2, wb)
        else:
            img2 = img.view(N * L, D)
            P0 = ops.gemm(img2, wi2)
    return P0, img2


class ImgProjLateFn(torch.autograd.Function):
    """The autograd node of an image projection whose product was ALREADY computed (img_project, issued at the very start
    of the forward on a side stream).  Created late -- right before the fusion node that consumes P0 -- so that autograd runs
    its backward (the weight-gradient GEMM, 14 ms in fp32) right after the fusion's backward and BEFORE the question-side
    backward is issued: on its own stream it then overlaps the LSTM backward.  (A node created first runs last.)
    dP arrives in P0's dtype: fp32, or bf16 straight from vqf_mfb_fuse_bwd_pbf16 (no cast pass)."""

    @staticmethod
    def forward(ctx, P0, img2, wi, cu_limit=0):
        ctx.save_for_backward(img2, wi)
        ctx.cu_limit = cu_limit
        return P0.view_as(P0)

    @staticmethod
    def backward(ctx, dP):
        img2, wi = ctx.saved_tensors
        dP = _c(dP)
        with (ops.options(gemm_cu_limit=ctx.cu_limit) if ctx.cu_limit else ops.options()):
            if img2.dtype == torch.bfloat16:
                dPb = dP if dP.dtype == torch.bfloat16 else ops.cast_bf16(dP)
                dwi = ops.gemm_bf16(dPb, img2, ta=True, tb=True).view_as(wi)
            else:
                dwi = ops.gemm(dP, img2, ta=True, tb=True).view_as(wi)            # wgrad, K = N*L
        return None, None, dwi, None


class MfbFuseFn(torch.autograd.Function):
    """a6: Y = L2norm_n(ssqrt(pool5(dropout((P0 + bias) * q[n])))) for the L regions of each sample.
    P0 fp32, or bf16 (then dP is handed back in bf16 too: the bf16 mode's projection storage)."""

    @staticmethod
    def forward(ctx, P0, bi, q, keep, seed, p_drop, N, L, link=None):
        P0, q = _c(P0), _c(q)
        O = P0.shape[1] // ops.POOL_K
        rb = [] if (link is not None and P0.dtype == torch.bfloat16) else None
        Y, norm, inv, _ = ops.mfb_fuse_fwd(P0, q, N, L, O, keep=keep, seed=seed, p_drop=p_drop, pbias=bi,
                                           normalise=link is None, r_bf16=rb)
        ctx.link = _arm_link(link, inv, L, rb)
        ctx.save_for_backward(P0, bi, q, Y, norm, inv, keep)
        ctx.seed, ctx.p_drop, ctx.dims = seed, p_drop, (N, L, O)
        return Y

    @staticmethod
    def backward(ctx, dY):
        P0, bi, q, Y, norm, inv, keep = ctx.saved_tensors
        N, L, O = ctx.dims
        dP, dq, _, dbi = ops.mfb_fuse_bwd(_c(dY), Y, norm, inv, P0, q, N, L, O, keep=keep, seed=ctx.seed,
                                          p_drop=ctx.p_drop, want_dbias=True, pbias=bi, lin=_take_lin(ctx.link),
                                          dp_bf16=P0.dtype == torch.bfloat16)
        return dP, dbi, dq, None, None, None, None, None, None


class _Fork:
    """Run a block of launches on a second stream beside the caller's (two independent ~100-us products of a final MFB block:
    each leaves a third of the chip's issue slots idle in its prologue, tail and slab reduce).  with _Fork(dev) as f: ... launches
    on the side stream ...; f.join(*tensors_made_there) makes the caller's stream wait and tells the allocator who reads them."""
    _streams = {}

    def __init__(self, device):
        self.cur = torch.cuda.current_stream(device)
        self.side = _Fork._streams.get(device)
        if self.side is None:
            self.side = _Fork._streams[device] = torch.cuda.Stream(device=device)
        self.side.wait_stream(self.cur)
        self._ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)

    def join(self, *tensors):
        self.cur.wait_stream(self.side)
        for t in tensors:
            if t is not None:
                t.record_stream(self.cur)


class FinalMfbFn(torch.autograd.Function):
    """a9: y = L2norm_row(ssqrt(pool5(dropout((qa Wq^T + bq) * (va Wv^T + bv))))), (N,1000).

    cascade (N,5000) optional third factor and want_zdrop: MHB's high-order block
    (mhb_coAtt.py:201-211) reuses this stage.
    """

    TWO_STREAMS = False    # A/B: the question-side and image-side products (and their gradients) on two streams

    @staticmethod
    def forward(ctx, qa, va, wq, bq, wv, bv, keep, seed, p_drop, cascade=None, want_zdrop=False, bf16=False):
        qa, va = _c(qa), _c(va)
        N = qa.shape[0]
        O = wq.shape[0] // ops.POOL_K
        wq2, wv2 = _w2d(wq), _w2d(wv)
        bf16 = bool(bf16) and _bf16_ok(N, qa.shape[1], va.shape[1], wq2.shape[0])
        if bf16:            # "bf16-all": both projections with bf16 operands, fp32 accumulate
            qa_s, va_s, wqb, wvb = ops.cast_bf16(qa), ops.cast_bf16(va), ops.cast_bf16(wq2), ops.cast_bf16(wv2)
            qq = ops.gemm_bf16(qa_s, wqb, bias=bq)
            vv = ops.gemm_bf16(va_s, wvb, bias=bv)
        elif FinalMfbFn.TWO_STREAMS:
            qa_s, va_s, wqb, wvb = qa, va, None, None
            with _Fork(qa.device) as f:
                vv = ops.gemm(va, wv2, bias=bv)
            qq = ops.gemm(qa, wq2, bias=bq)
            f.join(vv)
        else:
            qa_s, va_s, wqb, wvb = qa, va, None, None
            qq = ops.gemm(qa, wq2, bias=bq)
            vv = ops.gemm(va, wv2, bias=bv)
        if cascade is not None:
            cascade = _c(cascade)
        y, norm, inv, zdrop = ops.mfb_fuse_fwd(vv, qq, N, 1, O, keep=keep, seed=seed, p_drop=p_drop,
                                               cascade=cascade, want_zdrop=want_zdrop)
        ctx.save_for_backward(qa_s, va_s, wq, wv, qq, vv, y, norm, inv, keep, cascade, wqb, wvb)
        ctx.seed, ctx.p_drop, ctx.dims, ctx.bf16 = seed, p_drop, (N, O), bf16
        if want_zdrop:
            return y, zdrop
        return y

    @staticmethod
    def backward(ctx, dy, dz=None):
        qa, va, wq, wv, qq, vv, y, norm, inv, keep, cascade, wqb, wvb = ctx.saved_tensors
        N, O = ctx.dims
        dvv, dqq, dcasc, _ = ops.mfb_fuse_bwd(_c(dy), y, norm, inv, vv, qq, N, 1, O, keep=keep,
                                              seed=ctx.seed, p_drop=ctx.p_drop, cascade=cascade,
                                              dzdrop=None if dz is None else _c(dz))
        if ctx.bf16:
            dqb, dvb = ops.cast_bf16(dqq), ops.cast_bf16(dvv)
            dqa = ops.gemm_bf16(dqb, wqb, tb=True) if ctx.needs_input_grad[0] else None
            dva = ops.gemm_bf16(dvb, wvb, tb=True) if ctx.needs_input_grad[1] else None
            dwq = ops.gemm_bf16(dqb, qa, ta=True, tb=True).view_as(wq)
            dwv = ops.gemm_bf16(dvb, va, ta=True, tb=True).view_as(wv)
        elif FinalMfbFn.TWO_STREAMS:
            wq2, wv2 = _w2d(wq), _w2d(wv)
            with _Fork(dqq.device) as f:
                dva = ops.gemm(dvv, wv2, tb=True) if ctx.needs_input_grad[1] else None
                dwv = ops.gemm(dvv, va, ta=True, tb=True).view_as(wv)
                dbv = ops.colsum(dvv)
            dqa = ops.gemm(dqq, wq2, tb=True) if ctx.needs_input_grad[0] else None
            dwq = ops.gemm(dqq, qa, ta=True, tb=True).view_as(wq)
            dbq = ops.colsum(dqq)
            f.join(dva, dwv, dbv)
            return dqa, dva, dwq, dbq, dwv, dbv, None, None, None, dcasc, None, None
        else:
            wq2, wv2 = _w2d(wq), _w2d(wv)
            dqa = ops.gemm(dqq, wq2, tb=True) if ctx.needs_input_grad[0] else None
            dva = ops.gemm(dvv, wv2, tb=True) if ctx.needs_input_grad[1] else None
            dwq = ops.gemm(dqq, qa, ta=True, tb=True).view_as(wq)
            dwv = ops.gemm(dvv, va, ta=True, tb=True).view_as(wv)
        dbq = ops.colsum(dqq)
        dbv = ops.colsum(dvv)
        return dqa, dva, dwq, dbq, dwv, dbv, None, None, None, dcasc, None, None


# ---------------------------------------------------------------------------------------------
# stages used by HieCoAtten (hieCoAtten.py) and modules.py / networks.py
# ---------------------------------------------------------------------------------------------
class DropoutFn(torch.autograd.Function):
    """F.dropout with an in-kernel Philox mask (or an explicit uint8 keep-mask)."""

    @staticmethod
    def forward(ctx, x, keep, seed, p_drop):
        x = _c(x)
        ctx.keep, ctx.seed, ctx.p = keep, seed, p_drop
        return ops.dropout(x, keep=keep, seed=seed, p_drop=p_drop)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(_c(dy), keep=ctx.keep, seed=ctx.seed, p_drop=ctx.p), None, None, None


class DropoutBTFn(torch.autograd.Function):
    """The LSTM-output dropout (mfb.py:70, mhb_coAtt.py:75): x (B, T, H), any strides on the first two axes (MFB hands in the
    transposed view of its time-major LSTM states) -> CONTIGUOUS dropout(x); the gradient goes back in x's own layout, so the
    (T, B, H) <-> (B, T, H) re-layouts of both directions ride in the two dropout passes instead of torch copy kernels."""

    @staticmethod
    def forward(ctx, x, keep, seed, p_drop):
        ctx.keep, ctx.seed, ctx.p = keep, seed, p_drop
        ctx.in_strides = (x.stride(0), x.stride(1))
        return ops.dropout_bt(x, torch.empty(x.shape, dtype=torch.float32, device=x.device), keep, seed, p_drop)

    @staticmethod
    def backward(ctx, dy):
        B, T, H = dy.shape
        if dy.stride(2) != 1 or dy.stride(0) % 4 or dy.stride(1) % 4 or dy.data_ptr() % 16:
            dy = dy.contiguous()           # an offset view or odd strides: the kernel wants 16-byte rows
        sb, st = ctx.in_strides
        if st > sb and sb == H and st == B * H:                     # x was the transposed view of a contiguous (T, B, H) tensor
            dx = torch.empty((T, B, H), dtype=torch.float32, device=dy.device).transpose(0, 1)
        else:
            dx = torch.empty((B, T, H), dtype=torch.float32, device=dy.device)
        ops.dropout_bt(dy, dx, ctx.keep, ctx.seed, ctx.p)
        return dx, None, None, None


def lstm_out_dropout(module, x, seeds, tag="l"):
    """dropout_l / lstm_dropout of the reference modules on the HIP path: rate = the nn.Dropout's p in train mode, 0 in eval
    (then the pass only makes x contiguous); an explicit (B*T, H) keep-mask under `tag` replaces the in-kernel Philox draw."""
    p = float(module.p) if module.training else 0.0
    keep = seeds.keep.get(tag)
    if x.dim() == 3 and x.is_cuda and x.dtype == torch.float32 and x.stride(2) == 1 and x.shape[2] % 4 == 0 \
            and all(s % 4 == 0 for s in x.stride()[:2]) and x.data_ptr() % 16 == 0:
        if p <= 0.0 and keep is None and x.is_contiguous():
            return x
        seed, pp = seeds.next(module.training, p)
        return DropoutBTFn.apply(x, keep, seed, p if keep is not None else pp)
    return module(x).contiguous()


class TanhDropFn(torch.autograd.Function):
    """y = dropout(tanh(a [+ b]))   (hieCoAtten.py:32-33,38-39,45-46)."""

    @staticmethod
    def forward(ctx, a, b, keep, seed, p_drop):
        a = _c(a)
        b = None if b is None else _c(b)
        y = ops.tanh_dropout_fwd(a, b, keep=keep, seed=seed, p_drop=p_drop)
        ctx.save_for_backward(y)
        ctx.keep, ctx.seed, ctx.p, ctx.has_b = keep, seed, p_drop, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = ops.tanh_dropout_bwd(_c(dy), y, keep=ctx.keep, seed=ctx.seed, p_drop=ctx.p)
        return dx, (dx if ctx.has_b else None), None, None, None


class GateFn(torch.autograd.Function):
    """y = tanh(a) * sigmoid(b): the gate of Nonlinear_layer (modules.py:103-109), one launch each way."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        return ops.gate_tanh_sigmoid_fwd(a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        return ops.gate_tanh_sigmoid_bwd(_c(dy), a, b)


class BmmFn(torch.autograd.Function):
    """Batched C_b = Aop_b @ Bop_b^T on the fp32 MFMA GEMM.

    ta=False: a is (B,M,K), ta=True: a is (B,K,M); tb=False: b is (B,N,K), tb=True: b is (B,K,N).
    """

    @staticmethod
    def forward(ctx, a, b, ta, tb):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        ctx.ta, ctx.tb = ta, tb
        return ops.bgemm(a, b, ta=ta, tb=tb)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = _c(dc)
        ta, tb = ctx.ta, ctx.tb
        da = db = None
        if ctx.needs_input_grad[0]:
            if not ta:      # dA (B,M,K) = dC (M,N) x Bop (N,K)
                da = ops.bgemm(dc, b, ta=False, tb=not tb)
            else:           # dA (B,K,M) = Bop^T (K,N) x dC^T (N,M)
                da = ops.bgemm(b, dc, ta=not tb, tb=False)
        if ctx.needs_input_grad[1]:
            if not tb:      # dB (B,N,K) = dC^T (N,M) x Aop (M,K)
                db = ops.bgemm(dc, a, ta=True, tb=not ta)
            else:           # dB (B,K,N) = Aop^T (K,M) x dC (M,N)
                db = ops.bgemm(a, dc, ta=not ta, tb=True)
        return da, db, None, None


class AttPoolFn(torch.autograd.Function):
    """logits = x W^T + b (G rows, no hidden layer), softmax over the S positions of a sample,
    pooled[n] = sum_s wts[n,g,s] feat[n,s,:].  Returns (pooled (N,G*C), wts (N,G,S)), both
    differentiable (hieCoAtten.py:40-42,47-49 and :55; modules.py:60-65; networks.py:64-66).
    x (N*S, Cin), feat (N,S,C), w (G,Cin), b (G)."""

    @staticmethod
    def forward(ctx, x, feat, w, b, pooled_out=None):
        """pooled_out: a contiguous (N, G*C) buffer (a row block of a larger one) the pooled vectors are written into"""
        x, feat = _c(x), _c(feat)
        w2 = _w2d(w)
        logits = ops.att_logits_fwd(x, w2, b)
        wts, pooled = ops.glimpse_pool_fwd(feat, logits, False, pooled_out=pooled_out)
        ctx.save_for_backward(x, feat, w, wts)
        # (pooled_out is written as raw memory, like ImgProjDeferFn.fill: the output is a fresh view of it, autograd sees no
        #  in-place operation on the buffer)
        return (pooled.view_as(pooled) if pooled_out is not None else pooled), wts

    @staticmethod
    def backward(ctx, dpooled, dwts):
        x, feat, w, wts = ctx.saved_tensors
        dlogits, dfeat = ops.glimpse_pool_bwd(_c(dpooled), feat, wts, False, ctx.needs_input_grad[1],
                                              dwts=None if dwts is None else _c(dwts))
        dx, dw, db, _ = ops.att_logits_bwd(dlogits, x, _w2d(w), relu_mask=False)
        return (dx if ctx.needs_input_grad[0] else None), dfeat, dw.view_as(w), db, None


class HieCoreFn(torch.autograd.Function):
    """HieCoAtten's ladder from the raw inputs to cat((v, q), 0).view(N, -1) (hieCoAtten.py:25-53) as ONE autograd node with a
    hand-ordered backward, so that
      * fc_Wbv and fc_Wv, both applied to `img` (:30,35), are ONE product with the concatenated (2E, E) weight -- [Cv | img_] =
        img [Wbv; Wv]^T, a (N*L, 2E) buffer whose halves the later stages consume in place (row-strided operands) -- and the
        question side likewise ([Cq | que_] = que [Wbv; Wq]^T: :31 applies fc_Wbv to the question too);
      * their input gradients are ONE product with K = 2E over the gradient buffer [dCv | dimg_], which the batched products
        and the tanh backward fill in place (VQF_GEMM_ACCUM where two consumers meet): no gradient-accumulation adds of the
        (N*L, E) tensors, one weight-gradient product per side;
      * the attention pool's gradient into `img` -- the rank-1 term av[n,l] * dv[n,:] -- is never materialised: the backward of
        dropout(relu(img_emb(.))) adds it on the fly (vqf_relu_bwd_rank1_f32);
      * v and q land in the two halves of one (2N, E) buffer: the reference's cat is a view.
    drops: {tag: (keep mask | None, seed, p)} for 'img', 'que', 'C', 'Hv', 'Hq' (the always-on functional dropouts).
    Returns (x (N, 2E), av (N,1,L), aq (N,1,T)), all differentiable."""

    STREAM = True      # the tiny-T stages as streaming passes (csrc/hie.hip) where supported; False: batched GEMMs + element-wise (A/B)
    AFFINITY = True    # Cq Cv^T and its gradient on vqf_hie_affinity (needs STREAM); False: batched GEMMs + element-wise (A/B)

    @staticmethod
    def forward(ctx, imgf, ids, w_emb, b_emb, w_que, wbv, bbv, wv, bv, wq, bq, whv, bhv, whq, bhq, drops):
        imgf, ids = _c(imgf), ids.contiguous()
        N, L, D = imgf.shape
        T = ids.shape[1]
        E = w_emb.shape[0]
        M, MT = N * L, N * T
        dev = imgf.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        # :25-26  img = dropout(relu(img_emb(img_features)))   (ReLU in the GEMM epilogue, the dropout in place)
        # (per-sample tiles where the library takes the shape: 256 samples x 196 regions = one workgroup per CU, csrc/gemm_f32_sample.hip)
        img = ops.gemm_rows(imgf.view(M, D), _c(w_emb), L, bias=b_emb, relu=True)
        ops.dropout(img, *drops["img"], out=img)
        # :27-28  que = dropout(que_emb(que_features))
        if E % 4 == 0 and E <= 1024:
            que = ops.embed_dropout_fwd(_c(w_que), ids, *drops["que"])           # lookup + dropout in one launch
        else:
            que = ops.embed_tanh_fwd(_c(w_que), ids, False).view(MT, E)
            ops.dropout(que, *drops["que"], out=que)
        # :30-31,35-36  the four E x E layers as two products with concatenated weights
        Wi, bi, Wq2, bq2 = new(2 * E, E), new(2 * E), new(2 * E, E), new(2 * E)
        ops.multi_copy([(_c(wbv), Wi[:E]), (_c(wv), Wi[E:]), (bbv, bi[:E]), (bv, bi[E:]),
                        (_c(wbv), Wq2[:E]), (_c(wq), Wq2[E:]), (bbv, bq2[:E]), (bq, bq2[E:])])
        CI = ops.gemm_rows(img, Wi, L, bias=bi)               # (M, 2E)  = [Cv | img_]
        CQ = ops.gemm(que, Wq2, bias=bq2)                     # (MT, 2E) = [Cq | que_]
        Cv3, img_3 = CI[:, :E].view(N, L, E), CI[:, E:].view(N, L, E)
        Cq3, que_3 = CQ[:, :E].view(N, T, E), CQ[:, E:].view(N, T, E)
        # :32-33  C = dropout(tanh(Cq Cv^T))   (N,T,L)
        stream = HieCoreFn.STREAM and ops.hie_stream_supported(N, L, E, T)
        aff = stream and HieCoreFn.AFFINITY and ops.hie_affinity_supported(N, L, E, T, 2)
        if aff:                  # one pass over Cv on 16x16x4 MFMAs, tanh + dropout in its epilogue (csrc/hie.hip)
            C3 = ops.hie_affinity(CQ[:, :E], CI[:, :E], N, L, T, epi=1, drop=drops["C"])
        else:
            C3 = ops.bgemm(Cq3, Cv3)
            ops.tanh_dropout_fwd(C3.view(MT, L), None, *drops["C"], out=C3.view(MT, L))
        # :38-42  Hv = dropout(tanh(img_ + C^T que_)), av = softmax_L(Whv Hv), v = av^T img;  :45  ti = C img_
        if stream:
            # ONE pass over img_: the rank-T update, tanh, dropout, and the T-row sums of ti in registers (csrc/hie.hip)
            S = ops.hie_chunks(N, L)
            if S == 1:           # one workgroup per sample: the T-row sums are final, no partial slabs, no slab-sum launch
                ti = new(MT, E)
                Hv = ops.hie_hv_fwd(CI[:, E:], C3, CQ[:, E:], drops["Hv"], N, L, T, new(M, E), ti)
            else:
                part = new(S, MT, E)
                Hv = ops.hie_hv_fwd(CI[:, E:], C3, CQ[:, E:], drops["Hv"], N, L, T, new(M, E), part)
                ti = ops.hie_slab_sum(part, new(MT, E))
        else:
            tq = ops.bgemm(C3, que_3, ta=True, tb=True).view(M, E)
            Hv = ops.tanh_dropout_fwd2d(CI[:, E:], tq, *drops["Hv"], out=tq)
            ti = ops.bgemm(C3, img_3, ta=False, tb=True).view(MT, E)
        xcat = new(2 * N, E)
        av, _ = ops.glimpse_pool_fwd(img.view(N, L, E), ops.att_logits_fwd(Hv, _w2d(whv), bhv), False, pooled_out=xcat[:N])
        # :45-49  Hq = dropout(tanh(que_ + C img_)), aq = softmax_T(Whq Hq), q = aq^T que
        Hq = ops.tanh_dropout_fwd2d(CQ[:, E:], ti, *drops["Hq"], out=ti)
        aq, _ = ops.glimpse_pool_fwd(que.view(N, T, E), ops.att_logits_fwd(Hq, _w2d(whq), bhq), False, pooled_out=xcat[N:])
        ctx.save_for_backward(imgf, ids, img, que, Wi, Wq2, CI, CQ, C3, Hv, Hq, av, aq, whv, whq)
        ctx.drops, ctx.dims, ctx.V, ctx.stream, ctx.aff = drops, (N, L, T, D, E), w_que.shape[0], stream, aff
        # an output nobody differentiates (av / aq under a loss on x: solver.py:84-91) arrives as None in the backward, not as a
        # zero tensor torch has to fill and the pooling kernels have to read
        ctx.set_materialize_grads(False)
        return xcat.view(N, 2 * E), av, aq                    # :52-53: cat((v, q), 0).view(N, -1) is a view of xcat

    @staticmethod
    def backward(ctx, dx, dav, daq):
        imgf, ids, img, que, Wi, Wq2, CI, CQ, C3, Hv, Hq, av, aq, whv, whq = ctx.saved_tensors
        N, L, T, D, E = ctx.dims
        M, MT = N * L, N * T
        drops = ctx.drops
        dev = imgf.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        if dx is None:
            dx = torch.zeros((N, 2 * E), dtype=torch.float32, device=dev)
        dxcat = _c(dx).view(2 * N, E)
        dv, dq = dxcat[:N], dxcat[N:]
        Cv3, img_3 = CI[:, :E].view(N, L, E), CI[:, E:].view(N, L, E)
        Cq3, que_3 = CQ[:, :E].view(N, T, E), CQ[:, E:].view(N, T, E)
        dCI, dCQ = new(M, 2 * E), new(MT, 2 * E)              # [dCv | dimg_], [dCq | dque_]
        # question-side head: q = aq^T que, aq = softmax(Whq Hq), Hq = dropout(tanh(que_ + ti))
        dlq, dque = ops.glimpse_pool_bwd(dq, que.view(N, T, E), aq, False, True, dwts=None if daq is None else _c(daq))
        dHq, dwhq, dbhq, _ = ops.att_logits_bwd(dlq, Hq, _w2d(whq), relu_mask=False)
        if not ctx.stream:
            ops.tanh_dropout_bwd2d(dHq, Hq, *drops["Hq"], out=dCQ[:, E:])      # d(que_ + ti): first term of dque_ ...
        dti = ops.tanh_dropout_bwd2d(dHq, Hq, *drops["Hq"], out=dHq)           # ... and, untouched by the sums below, dti (7 MB)
        # image-side head
        dlv, _ = ops.glimpse_pool_bwd(dv, img.view(N, L, E), av, False, False, dwts=None if dav is None else _c(dav))
        dti3, dtq3 = dti.view(N, T, E), dCI[:, E:].view(N, L, E)
        if ctx.stream:
            # dtq = d(img_ + tq) straight from the logit gradient (dHv = dlv (x) whv is never written), C dtq and dlv^T Hv on the way
            S = ops.hie_chunks(N, L)
            # one buffer of partial rows, one per workgroup: [colsum dCv | colsum dimg_ | dl^T Hv | sum dl | 0 0 0]; its column
            # sums are the bias gradient of [fc_Wbv; fc_Wv] (no column-sum pass over the 205 MB gradient buffer) and d fc_Whv
            cpart = new(S * N, 3 * E + 4)
            part, wpart = (None if S == 1 else new(S, MT, E)), cpart[:, 2 * E:]
            if S == 1:           # dque_ = dti + C dtq written by the pass itself (one workgroup per sample)
                ops.hie_head_bwd(Hv, dlv.view(M), whv.view(E), C3, drops["Hv"], N, L, T, dCI[:, E:], dCQ[:, E:], wpart, part_add=dti)
            else:
                ops.hie_head_bwd(Hv, dlv.view(M), whv.view(E), C3, drops["Hv"], N, L, T, dCI[:, E:], part, wpart)
            if ctx.aff:          # dC = dti img_^T + que_ dtq^T and the backward of C = dropout(tanh(.)) in ONE pass over img_ and dtq
                dC3 = ops.hie_affinity(dti, CI[:, E:], N, L, T, x2=CQ[:, E:], y2=dCI[:, E:], epi=2, yprev=C3, drop=drops["C"])
            else:
                dC3 = ops.bgemm(dti3, img_3)                                       # dC = dti img_^T + que_ dtq^T
                ops.bgemm(que_3, dtq3, out=dC3, accumulate=True)
            if S > 1:
                ops.hie_slab_sum(part, dCQ[:, E:], add=dti)                        # dque_ = dti + C dtq
            ops.hie_rank_add(dCI[:, E:], C3, dti, N, L, T, dCI[:, E:], colpart=cpart[:, E:2 * E])      # dimg_ = dtq + C^T dti (dtq's own uses are above)
            # C = dropout(tanh(Cq Cv^T)):  dCv = daff^T Cq,  dCq = daff Cv   (one pass over Cv)
            if not ctx.aff:
                ops.tanh_dropout_bwd(dC3.view(MT, L), C3.view(MT, L), *drops["C"], out=dC3.view(MT, L))
            if S == 1:
                ops.hie_rank_left(dC3, CQ[:, :E], CI[:, :E], N, L, T, dCI[:, :E], dCQ[:, :E], colpart=cpart[:, :E])      # dCq straight into its column block
            else:
                ops.hie_rank_left(dC3, CQ[:, :E], CI[:, :E], N, L, T, dCI[:, :E], part, colpart=cpart[:, :E])
                ops.hie_slab_sum(part, dCQ[:, :E])
        else:
            dHv, dwhv, dbhv, _ = ops.att_logits_bwd(dlv, Hv, _w2d(whv), relu_mask=False)
            ops.tanh_dropout_bwd2d(dHv, Hv, *drops["Hv"], out=dCI[:, E:])          # d(img_ + tq), first term of dimg_
            del dHv
            # tq = C^T que_, ti = C img_:  dC = dti img_^T + que_ dtq^T;  dque_ += C dtq;  dimg_ += C^T dti
            dC3 = ops.bgemm(dti3, img_3)
            ops.bgemm(que_3, dtq3, out=dC3, accumulate=True)
            ops.bgemm(C3, dtq3, ta=False, tb=True, out=dCQ[:, E:].view(N, T, E), accumulate=True)
            ops.bgemm(C3, dti3, ta=True, tb=True, out=dtq3, accumulate=True)       # (dtq's own uses are above this line)
            # C = dropout(tanh(Cq Cv^T)):  dCq = daff Cv,  dCv = daff^T Cq
            ops.tanh_dropout_bwd(dC3.view(MT, L), C3.view(MT, L), *drops["C"], out=dC3.view(MT, L))
            ops.bgemm(dC3, Cv3, ta=False, tb=True, out=dCQ[:, :E].view(N, T, E))
            ops.bgemm(dC3, Cq3, ta=True, tb=True, out=dCI[:, :E].view(N, L, E))
        # the concatenated layers: one input-gradient product (K = 2E), one weight-gradient product and one column sum per side
        dimg = ops.gemm_rows(dCI, Wi, L, tb=True)                              # (M, E); the pool's rank-1 term is added below
        ops.gemm(dCQ, Wq2, tb=True, out=dque.view(MT, E), accumulate=True)     # on top of the pool's gradient into que
        dWi, dWq2 = ops.gemm(dCI, img, ta=True, tb=True), ops.gemm(dCQ, que, ta=True, tb=True)
        if ctx.stream:
            csum = ops.colsum(cpart)
            dbi, dwhv, dbhv = csum[:2 * E], csum[2 * E:3 * E], csum[3 * E:3 * E + 1]
        else:
            dbi = ops.colsum(dCI)
        dbq2 = ops.colsum(dCQ)
        dwbv, dbbv = new(E, E), new(E)
        ops.multi_add([(dWi[:E], dWq2[:E], dwbv), (dbi[:E], dbq2[:E], dbbv)])  # fc_Wbv serves both sides (:30-31)
        # img = dropout(relu(img_emb(.))): mask, 1 / (1 - p) and the pool's av[n,l] * dv[n,:] in one pass, in place
        keep_i, _, p_i = drops["img"]
        scale = 1.0 / (1.0 - p_i) if (keep_i is not None or p_i > 0.0) else 1.0
        dpre, db_emb = ops.relu_bwd_rank1(dimg, img, av.view(M), dv, L, scale, want_bias=True, out=dimg)
        dw_emb = ops.gemm(dpre, imgf.view(M, D), ta=True, tb=True)
        dque2 = dque.view(MT, E)
        if E % 4 == 0 and E <= 1024:
            dw_que = ops.embed_dropout_bwd(dque2, ids, ctx.V, *drops["que"])     # dropout's backward inside the segment sums
        else:
            ops.dropout(dque2, *drops["que"], out=dque2)
            dw_que = ops.embed_tanh_bwd(dque2, None, ids, ctx.V)
        return (None, None, dw_emb, db_emb, dw_que, dwbv, dbbv, dWi[E:], dbi[E:], dWq2[E:], dbq2[E:],
                dwhv.view_as(whv), dbhv, dwhq.view_as(whq), dbhq, None)


class SoftmaxRowsFn(torch.autograd.Function):
    """softmax over the last axis of a 2-D tensor (modules.py:91-92)."""

    @staticmethod
    def forward(ctx, x):
        y = ops.softmax_rows_fwd(_c(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.softmax_rows_bwd(_c(dy), y)


class LogSoftmaxRowsFn(torch.autograd.Function):
    """log_softmax over the last axis of the 2-D logits: the classifier tail of MHBCoAtt / MHB (mhb_coAtt.py:149-151,215-217)."""

    @staticmethod
    def forward(ctx, x):
        y = ops.log_softmax_rows_fwd(_c(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.log_softmax_rows_bwd(_c(dy), y)


def _bias_sum(b_ih, b_hh):
    """b_ih + b_hh of an nn.LSTM (one library launch; torch for anything the launch does not take)"""
    if b_ih is None:
        return None
    if b_ih.is_cuda and b_ih.dtype == torch.float32 and b_ih.is_contiguous() and b_hh.is_contiguous():
        out = torch.empty_like(b_ih)
        ops.multi_add([(b_ih.detach(), b_hh.detach(), out)])
        return out
    return b_ih + b_hh


def _copy_of(t):
    """a second tensor with t's values (b_hh's gradient beside b_ih's), one library launch"""
    if t is None:
        return None
    if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
        out = torch.empty_like(t)
        ops.multi_copy([(t, out)])
        return out
    return t.clone()


def _lstm_in_proj(x2, w_ih, bias, bf16_proj):
    """Input projection of a whole sequence, x2 (S*B, I) @ w_ih^T (+ b_ih + b_hh).  bf16_proj (gemm_dtype "bf16-all"):
    bf16 operands, columns zero-padded to a multiple of 8 by the cast (I = 300 -> 304); returns the casts for the
    backward."""
    if not bf16_proj:
        return ops.gemm(x2, _c(w_ih), bias=bias), None, None
    xb, wb = ops.cast_bf16(x2), ops.cast_bf16(_c(w_ih))
    return ops.gemm_bf16(xb, wb, bias=bias), xb, wb


def _lstm_in_proj_bwd(dgb, xb, wb, I, need_dx):
    """dX = dG W_ih and dW_ih = dG^T X with the bf16 operands of _lstm_in_proj (padded columns come back as zeros)."""
    dx = ops.gemm_bf16(dgb, wb, tb=True)[:, :I].contiguous() if need_dx else None
    dw = ops.gemm_bf16(dgb, xb, ta=True, tb=True)[:, :I].contiguous()
    return dx, dw


class LstmSeqFn(torch.autograd.Function):
    """Single-layer LSTM over dim 0 of x (S,B,I) -> hs (S,B,H), zero initial state.

    For small per-step batches B (MHBCoAtt's batch-axis recursion, mhb_coAtt.py:72-74: S = N, B = T):
    input projection and all weight gradients are MFMA GEMMs over the whole sequence; the recursion
    itself is one fused kernel launch per step (vqf_lstm_seq_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, bf16=False):
        x = _c(x)
        S, B, I = x.shape
        H = w_hh.shape[1]
        bias = _bias_sum(b_ih, b_hh)
        ctx.bf16 = bool(bf16)          # bf16 operands in the recurrent product (bf16 modes)
        ctx.bf16_proj = bf16 == "all" and H % 8 == 0       # "bf16-all": also in the input projection and its gradients
        xw, xb, wb = _lstm_in_proj(x.view(S * B, I), w_ih, bias, ctx.bf16_proj)
        xw = xw.view(S, B, 4 * H)
        hs, cs, gates = ops.lstm_seq_fwd(xw, _c(w_hh), bf16=ctx.bf16)
        ctx.save_for_backward(x, w_ih, w_hh, hs, cs, gates, xb, wb)
        ctx.has_bias = b_ih is not None
        return hs

    @staticmethod
    def backward(ctx, dhs):
        x, w_ih, w_hh, hs, cs, gates, xb, wb = ctx.saved_tensors
        S, B, I = x.shape
        H = w_hh.shape[1]
        dg = ops.lstm_seq_bwd(_c(dhs), gates, cs, _c(w_hh), bf16=ctx.bf16)                # (S,B,4H)
        dg2 = dg.view(S * B, 4 * H)
        dgb = ops.cast_bf16(dg2) if ctx.bf16 and H % 8 == 0 else None      # one cast serves every bf16 gradient product
        if ctx.bf16_proj:
            dx, dw_ih = _lstm_in_proj_bwd(dgb, xb, wb, I, ctx.needs_input_grad[0])
            dx = dx.view(S, B, I) if dx is not None else None
        else:
            dx = ops.gemm(dg2, _c(w_ih), tb=True).view(S, B, I) if ctx.needs_input_grad[0] else None
            dw_ih = ops.gemm(dg2, x.view(S * B, I), ta=True, tb=True)
        if S > 1:
            if dgb is not None:              # bf16 modes: the 60-GFLOP recurrent weight gradient takes bf16 operands too
                dw_hh = ops.gemm_bf16(dgb[B:], ops.cast_bf16(hs[:-1].reshape((S - 1) * B, H)), ta=True, tb=True)
            else:
                dw_hh = ops.gemm(dg[1:].reshape((S - 1) * B, 4 * H), hs[:-1].reshape((S - 1) * B, H), ta=True, tb=True)
        else:
            dw_hh = torch.zeros_like(w_hh)
        db = ops.colsum(dg2) if ctx.has_bias else None
        return dx, dw_ih, dw_hh, db, _copy_of(db), None


class LstmBatchFn(torch.autograd.Function):
    """Single-layer LSTM over dim 0 of x (T,B,I) -> hs (T,B,H), zero initial state, for LARGE per-step
    batches (the question encoder in its regular orientation, mfb.py:68-70: T = 14 steps of the N-row
    minibatch).  Input projection, the T recurrent products (accumulating into the projection) and all weight
    gradients are MFMA GEMMs; one point-wise kernel per step does the rest (vqf_lstm_cell_fwd / _bwd).
    bf16=True (bf16 mode): the recurrent products take bf16 operands (W_hh cast once, h_t / dG_t per step),
    fp32 accumulation; gates, cell state and every stored tensor stay fp32."""

    FUSED_STEP = True      # fp32: a step's recurrent product and its cell stage in one launch where supported (A/B switch)

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, bf16=False):
        x = _c(x)
        T, B, I = x.shape
        H = w_hh.shape[1]
        bias = _bias_sum(b_ih, b_hh)
        ctx.bf16 = bool(bf16) and H % 8 == 0
        ctx.bf16_proj = ctx.bf16 and bf16 == "all"          # "bf16-all": input projection and every weight gradient too
        whh = ops.cast_bf16(_c(w_hh)) if ctx.bf16 else _c(w_hh)
        gates, xb, wb = _lstm_in_proj(x.view(T * B, I), w_ih, bias, ctx.bf16_proj)
        gates = gates.view(T, B, 4 * H)                     # pre-activations -> activated in place
        hs = torch.empty((T, B, H), dtype=torch.float32, device=x.device)
        cs = torch.empty_like(hs)
        fused = LstmBatchFn.FUSED_STEP and not ctx.bf16 and ops.lstm_step_supported(B, H)
        for t in range(T):
            if t > 0 and fused:                 # product + cell in ONE launch (bit-identical to the two below; csrc/gemm_f32_wave.hip)
                ops.lstm_step_fwd(hs[t - 1], whh, gates[t], cs[t - 1], cs[t], hs[t])
                continue
            if t > 0:                                                               # += h_{t-1} W_hh^T
                if ctx.bf16:
                    ops.gemm_bf16(ops.cast_bf16(hs[t - 1]), whh, out=gates[t], accumulate=True)
                else:
                    ops.gemm(hs[t - 1], whh, out=gates[t], accumulate=True)
            ops.lstm_cell_fwd(gates[t], cs[t - 1] if t else None, cs[t], hs[t])
        ctx.save_for_backward(x, w_ih, w_hh, hs, cs, gates, xb, wb)
        ctx.has_bias = b_ih is not None
        return hs

    @staticmethod
    def backward(ctx, dhs):
        x, w_ih, w_hh, hs, cs, gates, xb, wb = ctx.saved_tensors
        T, B, I = x.shape
        H = w_hh.shape[1]
        dhs = _c(dhs)
        whh = ops.cast_bf16(_c(w_hh)) if ctx.bf16 else _c(w_hh)
        dG = torch.empty_like(gates)
        dc = torch.empty((B, H), dtype=torch.float32, device=x.device)
        dh = None
        for t in range(T - 1, -1, -1):
            ops.lstm_cell_bwd(dhs[t], dh, gates[t], cs[t], cs[t - 1] if t else None, t == T - 1, dc, dG[t])
            if t > 0:                                                               # dG_t W_hh  (B,H)
                dh = ops.gemm_bf16(ops.cast_bf16(dG[t]), whh, tb=True) if ctx.bf16 else ops.gemm(dG[t], whh, tb=True)
        dG2 = dG.view(T * B, 4 * H)
        if ctx.bf16_proj:
            dGb = ops.cast_bf16(dG2)
            dx, dw_ih = _lstm_in_proj_bwd(dGb, xb, wb, I, ctx.needs_input_grad[0])
            dx = dx.view(T, B, I) if dx is not None else None
        else:
            dx = ops.gemm(dG2, _c(w_ih), tb=True).view(T, B, I) if ctx.needs_input_grad[0] else None
            dw_ih = ops.gemm(dG2, x.view(T * B, I), ta=True, tb=True)
        if T > 1 and ctx.bf16_proj and B % 8 == 0:
            dw_hh = ops.gemm_bf16(dGb[B:], ops.cast_bf16(hs[:-1].reshape((T - 1) * B, H)), ta=True, tb=True)
        elif T > 1:
            dw_hh = ops.gemm(dG[1:].reshape((T - 1) * B, 4 * H), hs[:-1].reshape((T - 1) * B, H), ta=True, tb=True)
        else:
            dw_hh = torch.zeros_like(w_hh)
        db = ops.colsum(dG2) if ctx.has_bias else None
        return dx, dw_ih, dw_hh, db, _copy_of(db), None


class UnitPoolFn(torch.autograd.Function):
    """pooled[n, g*C + c] = sum_s feat[n, s, c] for g < G: the glimpse sums under mfb.py:84,118's
    singleton-axis softmax (weights == 1), without the attention MLP in front (MFB's `pruned` mode)."""

    @staticmethod
    def forward(ctx, feat, G):
        feat = _c(feat)
        N, S, C = feat.shape
        logits = torch.zeros((N * S, G), dtype=torch.float32, device=feat.device)     # ignored under unit weights
        wts, pooled = ops.glimpse_pool_fwd(feat, logits, True)
        ctx.save_for_backward(feat, wts)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        feat, wts = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None
        _, dfeat = ops.glimpse_pool_bwd(_c(dpooled), feat, wts, True, True)
        return dfeat, None


class DeadParamsFn(torch.autograd.Function):
    """Identity on `out` that hands EXACT-ZERO gradients to parameters whose contribution is provably
    dead (what the reference's autograd computes for them, the long way round)."""

    @staticmethod
    def forward(ctx, out, *params):
        ctx.meta = [(p.shape, p.dtype, p.device) for p in params]
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        return (g,) + tuple(torch.zeros(sh, dtype=dt, device=dv) for sh, dt, dv in ctx.meta)
