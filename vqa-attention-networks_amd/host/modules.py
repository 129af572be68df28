"""modules.py of the reference on the HIP path: Attention_layer, Attention_1, Attention_2,
Nonlinear_layer (modules.py:8-109).  Same constructors, forward signatures, state_dict keys.

Attention_1's additive score softmax_l(fc(f1[l] + f2[t])) does not depend on t (softmax is
shift-invariant), so the (N,T,L,D) broadcast of modules.py:57 is never materialised:
one single-glimpse attention pooling over f1 is computed and broadcast over T.
"""
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

from .functions import LinearFn, BmmFn, AttPoolFn, SoftmaxRowsFn, GateFn


class Attention_1(nn.Module):
    def __init__(self, feature_size):
        super(Attention_1, self).__init__()
        self.fc = nn.Linear(feature_size, 1)
        self.tanh = nn.Tanh()

    def forward(self, feature_1, feature_2):
        N, L, D = feature_1.shape
        T, V = feature_2.shape[1], feature_2.shape[2]
        assert (D == V), "dimension of feature_1 and feature_2 not match"
        f1 = feature_1.contiguous()
        pooled, wts = AttPoolFn.apply(f1.view(N * L, D), f1, self.fc.weight, self.fc.bias)
        f_hat = pooled.view(N, 1, D).expand(N, T, D)
        att = wts.view(N, 1, L).expand(N, T, L)
        return f_hat, att


class Attention_2(nn.Module):
    def __init__(self, feature_size):
        super(Attention_2, self).__init__()
        self.fc1 = nn.Linear(feature_size, feature_size, bias=False)
        self.fc2 = nn.Linear(feature_size, 1)

    def forward(self, feature_1, feature_2):
        N, L, D = feature_1.shape
        T, V = feature_2.shape[1], feature_2.shape[2]
        assert (D == V), "dimension of img_feature and q_feature not match"
        f1 = feature_1.contiguous()
        g = LinearFn.apply(f1.view(N * L, D), self.fc1.weight, None).view(N, L, D)       # :90
        s = BmmFn.apply(feature_2.contiguous(), g, False, False)                          # (N,T,L) :91
        att = SoftmaxRowsFn.apply(s.view(N * T, L)).view(N, T, L)                         # :92
        f_hat = BmmFn.apply(att, f1, False, True)                                         # (N,T,D) :94
        return f_hat, att


class Attention_layer(nn.Module):
    def __init__(self, feature_size, att_type=1):
        super(Attention_layer, self).__init__()
        self.nonlinear_1 = nn.ReLU()
        self.nonlinear_2 = nn.ReLU()
        if att_type == 1:
            self.att_layer = Attention_1(feature_size)
        elif att_type == 2:
            self.att_layer = Attention_2(feature_size)
        else:
            sys.exit(0)                                                                   # modules.py:20
        self.nonlinear_3 = nn.ReLU()

    def forward(self, feature_1, feature_2):
        # modules.py:26-33: ReLU both inputs, attend over the first, ReLU-residual into the second
        src = self.nonlinear_1(feature_1)
        dst = self.nonlinear_2(feature_2)
        attended, att = self.att_layer(src, dst)
        return (src, self.nonlinear_3(dst + attended), att)


class Nonlinear_layer(nn.Module):
    def __init__(self, f_size):
        super(Nonlinear_layer, self).__init__()
        self.fc1 = nn.Linear(f_size, f_size)
        self.fc2 = nn.Linear(f_size, f_size)

    def forward(self, inputs):
        shp = inputs.shape
        x = inputs.reshape(-1, shp[-1])
        o_1 = LinearFn.apply(x, self.fc1.weight, self.fc1.bias)
        o_2 = LinearFn.apply(x, self.fc2.weight, self.fc2.bias)
        if o_1.is_cuda and o_1.dtype == torch.float32 and o_1.numel() % 4 == 0:
            return GateFn.apply(o_1, o_2).view(shp)                                      # :108: one launch (csrc/elementwise.hip)
        return (torch.tanh(o_1) * torch.sigmoid(o_2)).view(shp)
