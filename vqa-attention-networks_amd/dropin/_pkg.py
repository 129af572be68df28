"""Locate the package (hyphenated directory name) for the drop-in shims."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)
pkg = importlib.import_module("vqa-attention-networks_amd")
