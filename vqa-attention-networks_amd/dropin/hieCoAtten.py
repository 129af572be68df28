"""Drop-in for the reference's `hieCoAtten` module (train_models.py:8)."""
from _pkg import pkg as _p

HieCoAtten = _p.HieCoAtten
