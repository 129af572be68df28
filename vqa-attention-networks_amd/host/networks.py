"""networks.py of the reference on the HIP path: iBOWIMG, AttentionNet (networks.py:7-69).

Projections, attention pooling and dropout run in libvqa_fusion.so; embedding lookup,
BatchNorm1d, ReLU/add glue and the cat/view reshapes stay torch ops (they are not part of the
fusion arithmetic).  Functional dropout is always on in the reference (networks.py:22,24,55,57);
`drop_p` is its rate here.
"""
import torch
import torch.nn as nn

from .functions import LinearFn, DropoutFn
from .modules import Attention_layer
from .mfb import _DropSeeds


class _AlwaysDropout:
    def _drop(self, x, tag):
        k = self._seeds.keep.get(tag)
        if k is None and self.drop_p <= 0.0:
            return x
        seed, p = self._seeds.next(True, self.drop_p)
        return DropoutFn.apply(x, k, seed, self.drop_p if k is not None else p)

    def set_keep_masks(self, **masks):
        self._seeds.keep = masks


class iBOWIMG(nn.Module, _AlwaysDropout):
    def __init__(self, img_size, vocab_size, embed_size, output_size):
        super(iBOWIMG, self).__init__()
        self.img_emb = nn.Linear(img_size, embed_size, bias=True)
        self.img_bn = nn.BatchNorm1d(embed_size)
        self.que_emb = nn.Embedding(vocab_size, embed_size)
        self.fc = nn.Linear(2 * embed_size, output_size)
        self.drop_p = 0.5
        self._seeds = _DropSeeds()

    def forward(self, img_features, que_features):
        img = self.img_bn(LinearFn.apply(img_features, self.img_emb.weight, self.img_emb.bias))
        img = self._drop(torch.relu(img), 'img')
        que = self.que_emb(que_features)
        N, T, E = que.shape
        que = self._drop(que.reshape(N * T, E), 'que').view(N, T, E)
        que = torch.sum(que, 1)
        x = torch.cat((img, que), 1)
        return LinearFn.apply(x, self.fc.weight, self.fc.bias)


class AttentionNet(nn.Module, _AlwaysDropout):
    def __init__(self, block_num=196, word_num=22, img_size=1024, vocab_size=15881, embed_size=512,
                 att_num=6, output_size=3000):
        super(AttentionNet, self).__init__()
        self.img_emb = nn.Linear(img_size, embed_size, bias=True)
        self.que_emb = nn.Embedding(vocab_size, embed_size)
        for i in range(att_num):
            self.add_module("att{}".format(i), Attention_layer(embed_size, 1))   # both branches use type 1 (:37-41)
        self.fc = nn.Linear(2 * block_num * word_num, output_size)
        self.batchnorm = nn.BatchNorm1d(output_size)
        self.att_num = att_num
        self.drop_p = 0.5
        self._seeds = _DropSeeds()

    def forward(self, img_features, que_features):
        N, L, D = img_features.shape
        img = LinearFn.apply(img_features.reshape(N * L, D), self.img_emb.weight, self.img_emb.bias, True)
        E = img.shape[1]
        img = self._drop(img, 'img').view(N, L, E)                                # :54-55
        que = self.que_emb(que_features)                                          # :56
        T = que.shape[1]
        que = self._drop(que.reshape(N * T, E), 'que').view(N, T, E)              # :57
        que_att = img_att = None
        for i in range(self.att_num):                                             # :58-62
            if i % 2 == 0:
                img, que, que_att = self._modules['att{}'.format(i)](img, que)
            else:
                que, img, img_att = self._modules['att{}'.format(i)](que, img)
        x = torch.cat((que_att, img_att.transpose(1, 2)), 0)                      # :64
        x = x.reshape(N, -1)                                                      # :65
        x = LinearFn.apply(x, self.fc.weight, self.fc.bias)                       # :66
        x = self.batchnorm(x)                                                     # :68
        return x, que_att, img_att
