"""networks.py of the reference on the HIP path: iBOWIMG, AttentionNet (networks.py:7-69).

Projections, attention pooling and dropout run in libvqa_fusion.so; embedding lookup,
BatchNorm1d, ReLU/add glue and the cat/view reshapes stay torch ops (they are not part of the
fusion arithmetic).  Functional dropout is always on in the reference (networks.py:22,24,55,57);
`drop_p` is its rate here.

Parameter names (and therefore state_dict keys) are the reference's:
  iBOWIMG       img_emb.*, img_bn.*, que_emb.weight, fc.*
  AttentionNet  img_emb.*, que_emb.weight, att{0..att_num-1}.att_layer.fc.*, fc.*, batchnorm.*
"""
import torch
import torch.nn as nn

from .functions import LinearFn, DropoutFn, embed
from .modules import Attention_layer
from .mfb import _DropSeeds


def _linear(layer, x, relu=False):
    """nn.Linear parameters, HIP GEMM (+ fused ReLU)."""
    return LinearFn.apply(x, layer.weight, layer.bias, relu)


class _AlwaysDropout:
    """F.dropout(x) with training left at its default True: active in eval() too (reference behaviour).
    Masks come from the in-kernel Philox stream; tests may inject explicit keep-masks by tag."""

    drop_p = 0.5

    def _init_dropout(self):
        self._seeds = _DropSeeds()

    def _drop(self, x, tag):
        explicit = self._seeds.keep.get(tag)
        if explicit is None and self.drop_p <= 0.0:
            return x
        seed, rate = self._seeds.next(True, self.drop_p)
        return DropoutFn.apply(x, explicit, seed, self.drop_p if explicit is not None else rate)

    def _drop_tokens(self, emb, tag):
        """dropout over a (N, T, E) embedding, applied on its flat (N*T, E) view."""
        n, t, e = emb.shape
        return self._drop(emb.reshape(n * t, e), tag).view(n, t, e)

    def set_keep_masks(self, **masks):
        self._seeds.keep = masks


class iBOWIMG(nn.Module, _AlwaysDropout):
    """Bag-of-words baseline (networks.py:7-28): BN(ReLU-less Linear(img)) -> ReLU -> dropout, summed word
    embeddings -> dropout, concat, Linear."""

    def __init__(self, img_size, vocab_size, embed_size, output_size):
        nn.Module.__init__(self)
        self._init_dropout()
        self.img_emb = nn.Linear(img_size, embed_size, bias=True)        # networks.py:10
        self.img_bn = nn.BatchNorm1d(embed_size)                         # :11
        self.que_emb = nn.Embedding(vocab_size, embed_size)              # :12
        self.fc = nn.Linear(embed_size * 2, output_size)                 # :13

    def forward(self, img_features, que_features):
        image = torch.relu(self.img_bn(_linear(self.img_emb, img_features)))      # :17-21
        image = self._drop(image, 'img')                                          # :22
        words = self._drop_tokens(embed(self.que_emb, que_features), 'que')       # :23-24
        bag = words.sum(dim=1)                                                    # :25
        return _linear(self.fc, torch.cat((image, bag), dim=1))                   # :26-28


class AttentionNet(nn.Module, _AlwaysDropout):
    """att_num alternating Attention_layer(type 1) blocks over (image regions, question tokens), then the
    reference's cat(.., 0).view(N, -1) of the last two attention maps (it pairs rows ACROSS samples,
    networks.py:64-65 -- reproduced), Linear and BatchNorm1d."""

    def __init__(self, block_num=196, word_num=22, img_size=1024, vocab_size=15881, embed_size=512,
                 att_num=6, output_size=3000):
        nn.Module.__init__(self)
        self._init_dropout()
        self.att_num = att_num
        self.img_emb = nn.Linear(img_size, embed_size, bias=True)        # networks.py:34
        self.que_emb = nn.Embedding(vocab_size, embed_size)              # :35
        for idx in range(att_num):                                       # :36-41: both branches build type 1
            self.add_module("att%d" % idx, Attention_layer(embed_size, 1))
        self.fc = nn.Linear(block_num * word_num * 2, output_size)       # :42
        self.batchnorm = nn.BatchNorm1d(output_size)                     # :43

    def _block(self, idx):
        return self._modules["att%d" % idx]

    def forward(self, img_features, que_features):
        n, regions, channels = img_features.shape
        image = _linear(self.img_emb, img_features.reshape(n * regions, channels), relu=True)   # :51-54
        image = self._drop(image, 'img').view(n, regions, -1)                                   # :55
        words = self._drop_tokens(embed(self.que_emb, que_features), 'que')                     # :56-57
        que_att = img_att = None
        for idx in range(self.att_num):                                                         # :58-62
            if idx % 2 == 0:
                image, words, que_att = self._block(idx)(image, words)
            else:
                words, image, img_att = self._block(idx)(words, image)
        mixed = torch.cat((que_att, img_att.transpose(1, 2)), 0).reshape(n, -1)                 # :64-65
        out = self.batchnorm(_linear(self.fc, mixed))                                           # :66-68
        return out, que_att, img_att
