"""MHBCoAtt and MHB on the HIP fusion path (reference interface: mhb_coAtt.py).

MHBCoAtt keeps the reference's behaviour, including the LSTM that is built
with batch_first=True but fed (T,N,.) and therefore recurs over the minibatch
axis (mhb_coAtt.py:27-36,72-74); set `fix_lstm_orientation=True` to opt out.
"""
import torch
import torch.nn as nn

from . import ops
from .functions import (LinearFn, Linear2Fn, AttHeadFn, ImgFuseFn, MfbFuseFn, FinalMfbFn, LstmSeqFn, LstmBatchFn, LogSoftmaxRowsFn,
                        NormLink, embed_tanh, embed, lstm_out_dropout)
from .mfb import _DropSeeds, _image_is_data, _SideStream, _lstm_bf16, batch_first_lstm, warn_once


class MHBCoAtt(nn.Module):
    def __init__(self, cfg):
        super(MHBCoAtt, self).__init__()
        self.cfg = cfg
        self.word_embedding = nn.Embedding(cfg.q_vocab_size, cfg.emb_dim)
        lstm_in = cfg.emb_dim * 2 if cfg.glove else cfg.emb_dim
        self.lstm = nn.LSTM(input_size=lstm_in, hidden_size=cfg.hidden_dim,
                            num_layers=cfg.num_layers, batch_first=True)
        self.dropout_l = nn.Dropout(p=0.3)
        self.ques_att_conv1 = nn.Conv2d(cfg.hidden_dim, 512, [1, 1])
        self.ques_att_conv2 = nn.Conv2d(512, 2, [1, 1])
        self.ques_proj1 = nn.Linear(2 * cfg.hidden_dim, 5000)
        self.img_conv1d = nn.Conv2d(cfg.img_feature_channel, 5000, [1, 1])
        self.dropout_m = nn.Dropout(p=0.1)
        self.co_att_conv1 = nn.Conv2d(1000, 512, [1, 1])
        self.co_att_conv2 = nn.Conv2d(512, 2, [1, 1])
        self.ques_proj2 = nn.Linear(2 * cfg.hidden_dim, 5000)
        self.ques_proj3 = nn.Linear(2 * cfg.hidden_dim, 5000)
        self.img_proj2 = nn.Linear(2 * cfg.img_feature_channel, 5000)
        self.img_proj3 = nn.Linear(2 * cfg.img_feature_channel, 5000)
        self.linear_pred = nn.Linear(2000, cfg.a_vocab_size)
        self.fix_lstm_orientation = False
        self.gemm_dtype = "fp32"          # or "bf16" (BASELINE config 3), see MFB.gemm_dtype
        # the batch-axis recursion as one fused HIP kernel per step instead of nn.LSTM (MIOpen spends
        # ~35 ms per step on 512 sequential tiny steps); same arithmetic, see csrc/lstm.hip
        self.use_hip_lstm = True
        self.overlap_streams = True       # img_conv1d on a side stream, see MFB.overlap_streams
        self.fuse_bf16_dp = True          # see MFB.fuse_bf16_dp
        self.fold_norm = True             # see MFB.fold_norm
        self.side_cu_limit = 0            # see MFB.side_cu_limit
        self.side_bf16 = False            # see MFB.side_bf16
        self._side = _SideStream()
        self._seeds = _DropSeeds()

    def set_keep_masks(self, **masks):
        self._seeds.keep = masks

    def forward(self, img_features, questions, glove_matrix=None, is_training=True):
        _image_is_data(img_features, self.gemm_dtype)
        N, L, D = img_features.shape
        keep = self._seeds.keep
        bf16_img = self.gemm_dtype in ("bf16", "bf16-img", "bf16-all")
        bf16_all = self.gemm_dtype == "bf16-all"          # also ques_proj*, img_proj*, the question-attention conv
        # bf16 mode keeps projection + fusion in one autograd node (ImgFuseFn): its backward hands dP to the
        # weight-gradient GEMM in bf16 without an fp32 round trip, which is worth more than the stream overlap
        # (a real second stream -- overlap_streams is True -- keeps the bf16 hand-off too: MfbFuseFn takes / returns bf16)
        side = self.overlap_streams and not (bf16_img and self.fuse_bf16_dp and not (self.overlap_streams is True and self.side_bf16))
        proj = self._side.project(img_features, self.img_conv1d, bf16_img,
                                  self.overlap_streams == "same-stream", self.side_cu_limit) if side else None
        que_embedded = embed_tanh(self.word_embedding, questions)            # (N,T,E)
        if self.cfg.glove:
            assert glove_matrix is not None, 'glove should not be NoneType.'
            que_embedded = torch.cat((que_embedded, glove_matrix), dim=2)
        if self.fix_lstm_orientation:
            lstm_o = batch_first_lstm(self.lstm, que_embedded, self.use_hip_lstm, _lstm_bf16(self.gemm_dtype))   # (N,T,H)
            ques_feature = lstm_out_dropout(self.dropout_l, lstm_o, self._seeds)
        elif (self.use_hip_lstm and self.lstm.num_layers == 1 and que_embedded.is_cuda
              and ops.lstm_seq_supported(que_embedded.shape[1], self.cfg.hidden_dim)):
            # batch_first LSTM fed (T,N,.): sequence axis = N, per-step batch = T  (mhb_coAtt.py:72-74).
            # (N,T,.) is already (S,B,.) for the sequence kernel and its output (N,T,H) is the
            # reference's lstm_o.permute(1,0,2).
            hs = LstmSeqFn.apply(que_embedded, self.lstm.weight_ih_l0, self.lstm.weight_hh_l0,
                                 self.lstm.bias_ih_l0, self.lstm.bias_hh_l0,
                                 _lstm_bf16(self.gemm_dtype))     # bf16 modes: bf16 operands in the recurrent product
            ques_feature = lstm_out_dropout(self.dropout_l, hs, self._seeds)        # mhb_coAtt.py:75
        else:
            if self.use_hip_lstm:
                warn_once("lstm_seq", "MHBCoAtt's batch-axis LSTM recursion runs on nn.LSTM (MIOpen: one tiny step per "
                          "sample, ~35 ms per 512-sample step), not on the HIP LstmSeqFn, which covers one layer, "
                          "tokens per question <= 32 and hidden_dim in {256, 512, 768, 1024} (got T=%d, H=%d, layers=%d)"
                          % (que_embedded.shape[1], self.cfg.hidden_dim, self.lstm.num_layers))
            lstm_o, _ = self.lstm(que_embedded.permute(1, 0, 2))             # (T,N,H), recurs over N
            ques_feature = lstm_out_dropout(self.dropout_l, lstm_o.permute(1, 0, 2), self._seeds)   # (N,T,H) contiguous
        T, H = ques_feature.shape[1], ques_feature.shape[2]

        qa = AttHeadFn.apply(ques_feature.view(N * T, H), ques_feature,
                             self.ques_att_conv1.weight, self.ques_att_conv1.bias, None, None,
                             self.ques_att_conv2.weight, self.ques_att_conv2.bias, False, bf16_all, None, True)
        qp = LinearFn.apply(qa, self.ques_proj1.weight, self.ques_proj1.bias, False, bf16_all)
        pm = self.dropout_m.p
        seed, p = self._seeds.next(self.training, pm)
        k1 = keep.get('m1')
        coatt_bf16 = self.gemm_dtype in ("bf16", "bf16-att", "bf16-all")
        link = NormLink() if self.fold_norm else None
        if proj is not None:
            P0 = self._side.join(*proj)
            Y = MfbFuseFn.apply(P0, self.img_conv1d.bias, qp, k1, seed, pm if k1 is not None else p, N, L, link)
        else:
            Y = ImgFuseFn.apply(img_features, self.img_conv1d.weight, self.img_conv1d.bias, qp,
                                k1, seed, pm if k1 is not None else p, bf16_img, link)
        va = AttHeadFn.apply(Y, img_features, self.co_att_conv1.weight, self.co_att_conv1.bias, None, None,
                             self.co_att_conv2.weight, self.co_att_conv2.bias, False, coatt_bf16, link)
        ys = []
        for tag, qpj, ipj in (('m2', self.ques_proj2, self.img_proj2), ('m3', self.ques_proj3, self.img_proj3)):
            seed, p = self._seeds.next(self.training, pm)
            kk = keep.get(tag)
            ys.append(FinalMfbFn.apply(qa, va, qpj.weight, qpj.bias, ipj.weight, ipj.bias, kk, seed,
                                       pm if kk is not None else p, None, False, bf16_all))
        # :147-148  linear_pred(cat((att_normed_2, att_normed_3), 1)): the concatenation is never materialised
        logits = Linear2Fn.apply(ys[0], ys[1], self.linear_pred.weight, self.linear_pred.bias)
        return LogSoftmaxRowsFn.apply(logits)                                # :149-151 (implicit dim = 1 on the 2-D logits)


class MHB(nn.Module):
    """Mean-pooled image x last valid LSTM state, two cascaded MFB blocks (mhb_coAtt.py:153-217).

    The reference class cannot run as shipped (hard .cuda() at :176 is harmless
    here, but :214 names an undefined `mhb_22`); this follows the evident
    intent `mhb_12` (:213).
    """

    def __init__(self, cfg):
        super(MHB, self).__init__()
        self.model_name = cfg.model_name
        self.cfg = cfg
        self.mean_pool = nn.AvgPool2d((14, 14))
        self.Embedding = nn.Embedding(cfg.q_vocab_size, cfg.emb_dim)
        self.LSTM = nn.LSTM(input_size=cfg.emb_dim, hidden_size=cfg.hidden_dim, num_layers=1,
                            batch_first=False)
        self.linear_q_1 = nn.Linear(cfg.hidden_dim, 5000)
        self.linear_q_2 = nn.Linear(cfg.hidden_dim, 5000)
        self.linear_i_1 = nn.Linear(cfg.img_feature_channel, 5000)
        self.linear_i_2 = nn.Linear(cfg.img_feature_channel, 5000)
        self.lstm_dropout = nn.Dropout(0.3)
        self.mfb_dropout = nn.Dropout(0.1)
        self.linear_out = nn.Linear(2000, cfg.a_vocab_size)
        self.use_hip_lstm = True          # LSTM recursion on the HIP path (functions.LstmBatchFn) instead of MIOpen
        self._seeds = _DropSeeds()

    def set_keep_masks(self, **masks):
        self._seeds.keep = masks

    def forward(self, img_feature, questions, q_length):
        batch_size, max_len = questions.size()
        keep = self._seeds.keep
        # AvgPool2d(14,14) over the (N,C,14,14) view == mean over the 196 regions   :178-180
        # (the glimpse kernel with unit weights sums the regions; fp32 or bf16 feature storage)
        img3 = img_feature.reshape(batch_size, -1, self.cfg.img_feature_channel)
        _image_is_data(img3, "bf16")
        img3 = img3 if img3.is_contiguous() else img3.contiguous()
        L = img3.shape[1]
        _, i_sum = ops.glimpse_pool_fwd(img3, torch.zeros((batch_size * L, 1), device=img3.device), True)
        i_mean = i_sum * (1.0 / L)
        q_embedded = embed(self.Embedding, questions).permute(1, 0, 2)              # (T,N,E)  :181-182
        if (self.use_hip_lstm and q_embedded.is_cuda and self.LSTM.num_layers == 1 and not self.LSTM.bidirectional
                and self.LSTM.hidden_size % 4 == 0):
            lstm_outs = LstmBatchFn.apply(q_embedded.contiguous(), self.LSTM.weight_ih_l0, self.LSTM.weight_hh_l0,
                                          self.LSTM.bias_ih_l0, self.LSTM.bias_hh_l0)       # (T,N,H)
        else:
            if self.use_hip_lstm:
                warn_once("lstm_mhb", "MHB's LSTM runs on nn.LSTM (MIOpen), not on the HIP LstmBatchFn (one layer, "
                          "unidirectional, hidden_size % 4 == 0, GPU input)")
            lstm_outs, _ = self.LSTM(q_embedded)                             # (T,N,H)
        idx = (q_length.to(torch.long) - 1).to(lstm_outs.device)
        lstm_out = lstm_outs[idx, torch.arange(batch_size, device=lstm_outs.device)]   # :185-186
        lstm_out = lstm_out_dropout(self.lstm_dropout, lstm_out.unsqueeze(1), self._seeds).squeeze(1)      # :188
        pm = self.mfb_dropout.p
        seed, p = self._seeds.next(self.training, pm)
        k1 = keep.get('m1')
        mhb_1, z1 = FinalMfbFn.apply(lstm_out, i_mean, self.linear_q_1.weight, self.linear_q_1.bias,
                                     self.linear_i_1.weight, self.linear_i_1.bias, k1, seed,
                                     pm if k1 is not None else p, None, True)       # :190-199
        seed, p = self._seeds.next(self.training, pm)
        k2 = keep.get('m2')
        mhb_2 = FinalMfbFn.apply(lstm_out, i_mean, self.linear_q_2.weight, self.linear_q_2.bias,
                                 self.linear_i_2.weight, self.linear_i_2.bias, k2, seed,
                                 pm if k2 is not None else p, z1, False)             # :201-211
        # :213-214  linear_out(cat((mhb_1, mhb_2), 1)) without the concatenated tensor (:214 names mhb_22: see the class docstring)
        logits = Linear2Fn.apply(mhb_1, mhb_2, self.linear_out.weight, self.linear_out.bias)
        return LogSoftmaxRowsFn.apply(logits)                                # :215-217
