"""Drop-in for the reference's `networks` module."""
from _pkg import pkg as _p

iBOWIMG = _p.iBOWIMG
AttentionNet = _p.AttentionNet
