"""Drop-in for the reference's `modules` module (networks.py:4)."""
from _pkg import pkg as _p

Attention_layer = _p.Attention_layer
Attention_1 = _p.Attention_1
Attention_2 = _p.Attention_2
Nonlinear_layer = _p.Nonlinear_layer
