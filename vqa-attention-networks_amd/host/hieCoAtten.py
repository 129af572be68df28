"""HieCoAtten on the HIP path (reference interface: hieCoAtten.py:5-55).

Same constructor keywords, `forward(img_features, que_features) -> (x, av, aq)` and
state_dict keys.  Reference behaviour kept on purpose:
  * functional dropout (p = 0.5) is ALWAYS on, also under .eval() (hieCoAtten.py:26,28,33,39,46);
    `drop_p` is the rate (set it to 0.0 for deterministic runs);
  * fc_Wbv is applied to the question too, fc_Wbq is never used (:31);
  * x = cat((v, q), 0).view(N, -1) pairs rows across samples (:52-53).
"""
import torch
import torch.nn as nn

from .functions import LinearFn, DropoutFn, TanhDropFn, BmmFn, AttPoolFn, JoinRowsFn, HieCoreFn, embed, _plain_embedding
from .mfb import _DropSeeds


class HieCoAtten(nn.Module):
    def __init__(self, block_num=196, word_num=22, img_size=1024, vocab_size=15881, embed_size=512,
                 att_num=6, output_size=3000):
        super(HieCoAtten, self).__init__()
        self.img_emb = nn.Linear(img_size, embed_size, bias=True)
        self.que_emb = nn.Embedding(vocab_size, embed_size)
        self.fc_Wbv = nn.Linear(embed_size, embed_size)
        self.fc_Wbq = nn.Linear(embed_size, embed_size)
        self.fc_Wv = nn.Linear(embed_size, embed_size)
        self.fc_Wq = nn.Linear(embed_size, embed_size)
        self.fc_Whv = nn.Linear(embed_size, 1)
        self.fc_Whq = nn.Linear(embed_size, 1)
        self.fc = nn.Linear(2 * embed_size, output_size)
        self.drop_p = 0.5
        # True (default): the whole ladder :25-53 is ONE autograd node with concatenated fc_Wbv / fc_Wv (and fc_Wbv / fc_Wq)
        # products and a hand-ordered backward (functions.HieCoreFn); False: one node per stage, as rounds 1-3 ran it
        self.fused = True
        self._seeds = _DropSeeds()

    def set_keep_masks(self, **masks):
        """Test hook: uint8 keep-masks 'img' (N*L,E), 'que' (N*T,E), 'C' (N*T,L), 'Hv' (N*L,E), 'Hq' (N*T,E)."""
        self._seeds.keep = masks

    def _drop_args(self, tag):
        k = self._seeds.keep.get(tag)
        seed, p = self._seeds.next(True, self.drop_p)          # always on, like F.dropout(x)
        return k, seed, (self.drop_p if k is not None else p)

    def forward(self, img_features, que_features):
        N, L, D = img_features.shape
        T = que_features.shape[1]
        lin = lambda x, m, relu=False: LinearFn.apply(x, m.weight, m.bias, relu)
        E = self.img_emb.out_features
        if (self.fused and img_features.is_cuda and img_features.dtype == torch.float32 and _plain_embedding(self.que_emb, que_features)
                and E % 4 == 0 and L % 4 == 0 and not img_features.requires_grad):
            drops = {tag: self._drop_args(tag) for tag in ("img", "que", "C", "Hv", "Hq")}      # the draw order of the staged form
            x, av, aq = HieCoreFn.apply(img_features, que_features, self.img_emb.weight, self.img_emb.bias, self.que_emb.weight,
                                        self.fc_Wbv.weight, self.fc_Wbv.bias, self.fc_Wv.weight, self.fc_Wv.bias,
                                        self.fc_Wq.weight, self.fc_Wq.bias, self.fc_Whv.weight, self.fc_Whv.bias,
                                        self.fc_Whq.weight, self.fc_Whq.bias, drops)
            x = lin(x, self.fc)                                                     # :54
            return x, torch.squeeze(av.view(N, L, 1)), torch.squeeze(aq.view(N, T, 1))   # :43,50,55
        img = lin(img_features.reshape(N * L, D), self.img_emb, True)          # :25-26 (relu fused)
        img = DropoutFn.apply(img, *self._drop_args('img'))
        E = img.shape[1]
        que = embed(self.que_emb, que_features).reshape(N * T, E)                      # :27
        que = DropoutFn.apply(que, *self._drop_args('que'))                     # :28

        Cv = lin(img, self.fc_Wbv)                                              # :30
        Cq = lin(que, self.fc_Wbv)                                              # :31 (Wbv, as the reference)
        aff = BmmFn.apply(Cq.view(N, T, E), Cv.view(N, L, E), False, False)     # (N,T,L)  :32
        C = TanhDropFn.apply(aff.view(N * T, L), None, *self._drop_args('C')).view(N, T, L)   # :32-33

        img_ = lin(img, self.fc_Wv)                                             # :35
        que_ = lin(que, self.fc_Wq)                                             # :36
        tq = BmmFn.apply(C, que_.view(N, T, E), True, True)                     # (N,L,E) = C^T que_   :38
        Hv = TanhDropFn.apply(img_, tq.view(N * L, E), *self._drop_args('Hv'))  # :38-39
        xcat = torch.empty((2 * N, E), dtype=torch.float32, device=img.device)      # v and q are written into its two row blocks
        v, av = AttPoolFn.apply(Hv, img.view(N, L, E), self.fc_Whv.weight, self.fc_Whv.bias, xcat[:N])   # :40-42

        ti = BmmFn.apply(C, img_.view(N, L, E), False, True)                    # (N,T,E) = C img_     :45
        Hq = TanhDropFn.apply(que_, ti.view(N * T, E), *self._drop_args('Hq'))  # :45-46
        q, aq = AttPoolFn.apply(Hq, que.view(N, T, E), self.fc_Whq.weight, self.fc_Whq.bias, xcat[N:])   # :47-49

        x = JoinRowsFn.apply(v, q, xcat).view(N, -1)                            # :52-53 cat((v, q), 0).view(N, -1): row pairing, no copy
        x = lin(x, self.fc)                                                     # :54
        return x, torch.squeeze(av.view(N, L, 1)), torch.squeeze(aq.view(N, T, 1))   # :43,50,55
