"""Drop-in for the reference's `mfb` module: `mfb.MFB(cfg)` on the MI355X HIP path."""
from _pkg import pkg as _p

MFB = _p.MFB
