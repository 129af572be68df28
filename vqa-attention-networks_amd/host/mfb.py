"""MFB-baseline model on the HIP fusion path.

Mirrors the reference interface (mfb.py:6-140): `MFB(cfg)`,
`forward(img_features, questions, is_training=True) -> logits (N, a_vocab_size)`,
identical state_dict keys/shapes, so it drops into solver.py / train_models.py.
The question encoder's embedding lookup + tanh (mfb.py:68, csrc/embed.hip), its LSTM
recursion and everything from the question attention to the logits run in
libvqa_fusion.so (`use_hip_lstm = False` puts the LSTM back on nn.LSTM / MIOpen), dropout_l
(mfb.py:70) included (`vqf_dropout_bt`: the nn.Dropout module only carries the rate).
"""
import torch
import torch.nn as nn

from .functions import (LinearFn, AttHeadFn, ImgFuseFn, ImgProjFn, ImgProjLateFn, ImgProjDeferFn, MfbFuseFn, FinalMfbFn,
                        LstmBatchFn, UnitPoolFn, DeadParamsFn, NormLink, img_project, embed_tanh, lstm_out_dropout)


def _image_is_data(img, gemm_dtype="fp32"):
    """The image grid features are input data on this path (no d/d-image kernels: SURVEY 8a, a5),
    and they must live on the GPU: there is no CPU fallback.  A bf16 feature tensor (bf16 storage,
    data_loader.FeatureStager(bf16=True)) is accepted when the image projection runs in bf16."""
    from .lib import VqfError
    if not img.is_cuda:
        raise VqfError("vqa fusion modules need GPU tensors (HIP extension is the only path; no CPU fallback)")
    if img.dtype == torch.bfloat16:
        if gemm_dtype not in ("bf16", "bf16-img", "bf16-all"):
            raise VqfError("bf16 img_features need model.gemm_dtype = 'bf16', 'bf16-img' or 'bf16-all'; "
                           "the fp32 path takes fp32 features")
        if img.shape[-1] % 8:
            raise VqfError("bf16 img_features: the channel count must be a multiple of 8")
    elif img.dtype != torch.float32:
        raise VqfError("img_features must be fp32 or bf16, got %s" % img.dtype)
    if img.requires_grad:
        raise VqfError("img_features.requires_grad=True: the HIP fusion path treats the image tensor as "
                       "data and does not produce its gradient")


_warned = set()


def warn_once(key, msg):
    """One log line per process and reason when a module leaves the HIP path for a torch/MIOpen op
    (the answer stays the same, the speed does not: MHBCoAtt's batch-axis recursion costs ~35 ms per step there)."""
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn("vqa fusion path: " + msg, RuntimeWarning, stacklevel=3)


def _lstm_bf16(gemm_dtype):
    """LSTM precision flag of the functions.Lstm*Fn: False (fp32), True ("bf16": bf16 operands in the recurrent
    products) or "all" ("bf16-all": also in the input projection and the weight gradients)."""
    return "all" if gemm_dtype == "bf16-all" else gemm_dtype == "bf16"


def hip_batch_lstm_ok(lstm, x_is_cuda_fp32, use_hip=True):
    """whether batch_first_lstm runs the recursion on the HIP path (functions.LstmBatchFn)"""
    return bool(use_hip and x_is_cuda_fp32 and lstm.num_layers == 1 and not lstm.bidirectional and lstm.proj_size == 0
                and lstm.hidden_size % 4 == 0)


def batch_first_lstm(lstm, x, use_hip=True, bf16=False, time_major_in=False):
    """nn.LSTM(batch_first=True) forward of x (N,T,E) -> (N,T,H) with zero initial state (mfb.py:69).  The
    recursion runs on the HIP path (MFMA GEMMs + one point-wise kernel per step, functions.LstmBatchFn) with
    the nn.LSTM module's own parameters; anything it does not cover (several layers, bidirectional,
    projections) stays on nn.LSTM.  time_major_in: x arrives as (T,N,E) (embed_tanh(time_major=True))."""
    if hip_batch_lstm_ok(lstm, x.is_cuda and x.dtype == torch.float32, use_hip):
        hs = LstmBatchFn.apply(x if time_major_in else x.transpose(0, 1).contiguous(), lstm.weight_ih_l0, lstm.weight_hh_l0,
                               lstm.bias_ih_l0 if lstm.bias else None, lstm.bias_hh_l0 if lstm.bias else None, bf16)
        return hs.transpose(0, 1)
    if time_major_in:
        x = x.transpose(0, 1).contiguous()
    if use_hip:
        warn_once("lstm_batch", "question-encoder LSTM runs on nn.LSTM (MIOpen), not on the HIP LstmBatchFn: it needs "
                  "a GPU fp32 input, one layer, unidirectional, no projection, hidden_size %% 4 == 0 (got layers=%d, "
                  "bidirectional=%s, proj=%d, hidden=%d, dtype=%s)" % (lstm.num_layers, lstm.bidirectional,
                                                                      lstm.proj_size, lstm.hidden_size, x.dtype))
    out, _ = lstm(x)
    return out


class _SideStream:
    """Runs the image projection on a second HIP stream (one per device, created lazily)."""

    def __init__(self):
        self.streams = {}

    def project(self, img, conv, bf16, same_stream=False, cu_limit=0):
        dev = img.device
        if same_stream:
            # the projection stays its own autograd node but runs on the caller's stream: created first, its
            # backward (the weight-gradient GEMM) is the LAST node autograd runs, so every other gradient bucket
            # is already being all-reduced (on RCCL's stream) while that 15 ms GEMM computes
            if not bf16 and _SideStream.DEFER and img.dtype == torch.float32:
                # ... and its PRODUCT is issued by join(), behind the question encoder (functions.ImgProjDeferFn)
                return ("defer", ImgProjDeferFn.apply(img, conv.weight), img, conv.weight), None
            return ImgProjFn.apply(img, conv.weight, bf16), None
        # a second stream: the product is computed NOW, without an autograd node; join() creates the node late, so that
        # its backward (the weight gradient) is issued early in the backward pass (functions.ImgProjLateFn) and overlaps
        # the question-side backward the way the product overlaps the question-side forward
        side = self.streams.get(dev)
        if side is None:
            side = self.streams[dev] = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)                    # inputs / weights produced on the caller's stream
        with torch.cuda.stream(side), torch.no_grad():
            P0, img2 = img_project(img, conv.weight, bf16, cu_limit)
        return (P0, img2, conv.weight, cu_limit), side

    DEFER = True       # one-stream form: issue the projection's product behind the question encoder (A/B switch)

    @staticmethod
    def join(P0, side):
        if side is None:
            if isinstance(P0, tuple) and P0[0] == "defer":
                _, P, img, w = P0
                with torch.no_grad():
                    ImgProjDeferFn.fill(P, img, w)
                return P
            return P0
        P0, img2, w, cu_limit = P0
        with torch.cuda.stream(side):            # the node's stream = the stream its backward will run on
            P = ImgProjLateFn.apply(P0, img2, w, cu_limit)
        cur = torch.cuda.current_stream(P.device)
        cur.wait_stream(side)
        P0.record_stream(cur)                    # allocated on the side stream, consumed here
        img2.record_stream(cur)
        return P


class _DropSeeds:
    """Per-call dropout seeds for the in-kernel Philox masks (train mode)."""

    def __init__(self):
        self.keep = {}      # optional externally supplied uint8 keep-masks (parity tests)

    def next(self, training, p):
        if not training or p <= 0.0:
            return 0, 0.0
        # one 63-bit seed per call from torch's CPU generator: reproducible under manual_seed
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()), p


class MFB(nn.Module):
    def __init__(self, cfg):
        super(MFB, self).__init__()
        self.cfg = cfg
        self.word_embedding = nn.Embedding(cfg.q_vocab_size, cfg.emb_dim)
        self.lstm = nn.LSTM(input_size=cfg.emb_dim, hidden_size=cfg.hidden_dim,
                            num_layers=cfg.num_layers, batch_first=True)
        self.dropout_l = nn.Dropout(p=0.3)
        self.multilayer = cfg.model_name == 'mfb-multilayer'
        self.ques_att_conv1 = nn.Conv2d(cfg.hidden_dim, 1024, [1, 1])
        if self.multilayer:
            self.ques_att_multiconv = nn.Conv2d(1024, 512, [1, 1])
            self.ques_att_conv2 = nn.Conv2d(512, 2, [1, 1])
        else:
            self.ques_att_conv2 = nn.Conv2d(1024, 2, [1, 1])
        self.ques_proj1 = nn.Linear(2 * cfg.hidden_dim, 5000)
        self.img_conv1d = nn.Conv2d(cfg.img_feature_channel, 5000, [1, 1])
        self.dropout_m = nn.Dropout(p=0.1)      # its .p is the rate of the in-kernel Philox masks
        self.co_att_conv1 = nn.Conv2d(1000, 1024, [1, 1])
        if self.multilayer:
            self.co_att_multiconv = nn.Conv2d(1024, 512, [1, 1])
            self.co_att_conv2 = nn.Conv2d(512, 2, [1, 1])
        else:
            self.co_att_conv2 = nn.Conv2d(1024, 2, [1, 1])
        self.ques_proj2 = nn.Linear(2 * cfg.hidden_dim, 5000)
        self.img_proj2 = nn.Linear(2 * cfg.img_feature_channel, 5000)
        self.linear_pred = nn.Linear(1000, cfg.a_vocab_size)
        # reference_compat: mfb.py:84,118 run both softmaxes over a singleton axis (weights == 1)
        self.unit_softmax = True
        # Execution mode under unit_softmax.  False ("faithful", default and the benchmarked one): every op
        # the reference's autograd executes is executed.  True ("pruned"): with attention weights == 1 the
        # glimpses are plain sums, so both attention MLPs, ques_proj1, img_conv1d and the fusion over the
        # regions cannot influence the logits and their 12 parameter tensors get exactly-zero gradients;
        # the pruned mode skips that work (96 % of the FLOPs) and returns bit-identical logits and gradients.
        self.pruned = False
        # "fp32" (default, parity 1e-4) or "bf16": bf16 operands / fp32 accumulate for the two large
        # GEMM families (img_conv1d and co_att_conv1, 96 % of the FLOPs); everything else stays fp32.
        # "bf16-all": additionally ques_proj1, the final blocks' ques_proj* / img_proj* and the question-attention
        # conv take bf16 operands (forward, dgrad and weight gradient), and so do the LSTM's input projection (K = 300
        # zero-padded to 304 by the cast) with its two gradients and the recurrent weight gradient; the classifier, the
        # fusion / attention / normalisation arithmetic, the LSTM's gates and state and every reduction stay fp32
        self.gemm_dtype = "fp32"
        # True: run img_conv1d (and, through autograd, its weight gradient) on a side stream, concurrently
        # with the question encoder / question attention (and their backward + gradient all-reduce).
        # "same-stream": the projection is its own autograd node on the caller's stream (its weight gradient
        # then runs last in the backward, behind which the other buckets' all-reduce hides); one compute stream.
        # False: projection + fusion as one node (ImgFuseFn).
        self.overlap_streams = True
        # question encoder's LSTM recursion on the HIP path instead of nn.LSTM / MIOpen (same parameters)
        self.use_hip_lstm = True
        # bf16 mode: one autograd node for projection + fusion, whose backward writes dP in bf16 for the
        # weight-gradient GEMM (no 2 GB fp32 round trip, no cast pass); takes precedence over the overlap
        self.fuse_bf16_dp = True
        # F.normalize (mfb.py:105) folded into co_att_conv1's GEMM epilogue: fusion_normed is never written and neither the
        # scale pass nor the sum(Y * dY) pass of its backward runs (functions.NormLink).  fp32 co-attention, single hidden
        # layer only; False materialises fusion_normed as round 2 did
        self.fold_norm = True
        # overlap_streams = True only: the persistent image-projection GEMMs use at most this many CUs (a multiple of 8; 0 =
        # all), which leaves the others to the question-side kernels of the main stream (library option gemm_cu_limit)
        self.side_cu_limit = 0
        # bf16 modes keep projection + fusion in ONE node on the caller's stream by default (fuse_bf16_dp); side_bf16 = True
        # lets overlap_streams = True put the bf16 projection on the second stream as well (bf16 P in, bf16 dP out of
        # MfbFuseFn): worth it where the question side is long and idle, i.e. MHBCoAtt's 512-step LSTM recursion
        self.side_bf16 = False
        self._side = _SideStream()
        self._seeds = _DropSeeds()

    # -- helpers -----------------------------------------------------------
    def _mc(self, name):
        m = getattr(self, name, None) if self.multilayer else None
        return (m.weight, m.bias) if m is not None else (None, None)

    def set_keep_masks(self, **masks):
        """Test hook: explicit uint8 keep-masks 'm1' (N*L,5000), 'm2' (N,5000) instead of Philox."""
        self._seeds.keep = masks

    def _forward_pruned(self, img_features, ques_feature, keep):
        """The live part of the reference graph when both softmaxes are over a singleton axis."""
        pm = self.dropout_m.p
        qa = UnitPoolFn.apply(ques_feature, 2)                             # (N, 2H)   mfb.py:85-89 with weights 1
        self._seeds.next(self.training, pm)                                # the regions' dropout draw (unused, keeps the stream)
        va = UnitPoolFn.apply(img_features, 2)                             # (N, 2D)   mfb.py:119-123 with weights 1
        seed, p = self._seeds.next(self.training, pm)
        k2 = keep.get('m2')
        y = FinalMfbFn.apply(qa, va, self.ques_proj2.weight, self.ques_proj2.bias,
                             self.img_proj2.weight, self.img_proj2.bias, k2, seed, pm if k2 is not None else p)
        out = LinearFn.apply(y, self.linear_pred.weight, self.linear_pred.bias)
        dead = [self.ques_att_conv1, self.ques_att_conv2, self.ques_proj1, self.img_conv1d, self.co_att_conv1,
                self.co_att_conv2]
        if self.multilayer:
            dead += [self.ques_att_multiconv, self.co_att_multiconv]
        params = [t for m in dead for t in (m.weight, m.bias) if t.requires_grad]
        return DeadParamsFn.apply(out, *params) if params and torch.is_grad_enabled() else out

    def forward(self, img_features, questions, is_training=True):
        _image_is_data(img_features, self.gemm_dtype)
        bf16_img = self.gemm_dtype in ("bf16", "bf16-img", "bf16-all")
        bf16_all = self.gemm_dtype == "bf16-all"          # also ques_proj*, img_proj*, the question-attention conv
        # a5 starts first, on the side stream: it only needs the image and its weights
        # bf16 mode keeps projection + fusion in one autograd node (ImgFuseFn): its backward hands dP to the
        # weight-gradient GEMM in bf16 without an fp32 round trip, which is worth more than the stream overlap
        # (a real second stream -- overlap_streams is True -- keeps the bf16 hand-off too: MfbFuseFn takes / returns bf16)
        side = self.overlap_streams and not (bf16_img and self.fuse_bf16_dp and not (self.overlap_streams is True and self.side_bf16)) \
            and not (self.pruned and self.unit_softmax)
        proj = self._side.project(img_features, self.img_conv1d, bf16_img,
                                  self.overlap_streams == "same-stream", self.side_cu_limit) if side else None
        # a2: question encoder                                               mfb.py:68-70
        # (the lookup writes (T,N,E) directly when the recursion runs on the HIP path: no transposing copy in between)
        tm = questions.dim() == 2 and hip_batch_lstm_ok(self.lstm, questions.is_cuda and self.word_embedding.weight.dtype == torch.float32,
                                                        self.use_hip_lstm)
        que_embedded = embed_tanh(self.word_embedding, questions, time_major=tm)
        lstm_o = batch_first_lstm(self.lstm, que_embedded, self.use_hip_lstm, _lstm_bf16(self.gemm_dtype), time_major_in=tm)
        ques_feature = lstm_out_dropout(self.dropout_l, lstm_o, self._seeds)   # (N,T,H) contiguous; mfb.py:70
        N, T, H = ques_feature.shape
        L = img_features.shape[1]
        keep = self._seeds.keep
        if self.pruned and self.unit_softmax:
            return self._forward_pruned(img_features, ques_feature, keep)

        # a3: question attention                                             mfb.py:73-89
        wm, bm = self._mc('ques_att_multiconv')
        qa = AttHeadFn.apply(ques_feature.view(N * T, H), ques_feature,
                             self.ques_att_conv1.weight, self.ques_att_conv1.bias, wm, bm,
                             self.ques_att_conv2.weight, self.ques_att_conv2.bias, self.unit_softmax, bf16_all, None, True)
        # a4: ques_proj1                                                     mfb.py:92-93
        qp = LinearFn.apply(qa, self.ques_proj1.weight, self.ques_proj1.bias, False, bf16_all)
        # a5+a6: image projection + MFB fusion over the regions             mfb.py:95-106
        pm = self.dropout_m.p
        seed, p = self._seeds.next(self.training, pm)
        k1 = keep.get('m1')
        coatt_bf16 = self.gemm_dtype in ("bf16", "bf16-att", "bf16-all")
        link = NormLink() if (self.fold_norm and not self.multilayer) else None
        if proj is not None:
            P0 = self._side.join(*proj)
            Y = MfbFuseFn.apply(P0, self.img_conv1d.bias, qp, k1, seed, pm if k1 is not None else p, N, L, link)
        else:
            Y = ImgFuseFn.apply(img_features, self.img_conv1d.weight, self.img_conv1d.bias, qp,
                                k1, seed, pm if k1 is not None else p, bf16_img, link)
        # a7+a8: co-attention over the regions                               mfb.py:109-123
        # (with a link Y is the un-normalised R and 1/norm rides in co_att_conv1's GEMM epilogue)
        wm, bm = self._mc('co_att_multiconv')
        va = AttHeadFn.apply(Y, img_features, self.co_att_conv1.weight, self.co_att_conv1.bias, wm, bm,
                             self.co_att_conv2.weight, self.co_att_conv2.bias, self.unit_softmax, coatt_bf16, link)
        # a9: final MFB block                                                mfb.py:126-135
        seed, p = self._seeds.next(self.training, pm)
        k2 = keep.get('m2')
        y = FinalMfbFn.apply(qa, va, self.ques_proj2.weight, self.ques_proj2.bias,
                             self.img_proj2.weight, self.img_proj2.bias, k2, seed,
                             pm if k2 is not None else p, None, False, bf16_all)
        # a10: classifier; the reference computes a softmax and discards it  mfb.py:137-140
        return LinearFn.apply(y, self.linear_pred.weight, self.linear_pred.bias)
