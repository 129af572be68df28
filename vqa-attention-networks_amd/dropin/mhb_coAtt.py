"""Drop-in for the reference's `mhb_coAtt` module (train_models.py:9)."""
from _pkg import pkg as _p

MHBCoAtt = _p.MHBCoAtt
MHB = _p.MHB
