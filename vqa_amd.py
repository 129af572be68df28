"""Alias: `import vqa_amd` == the package in ./vqa-attention-networks_amd (hyphenated name)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("vqa-attention-networks_amd")
sys.modules[__name__] = _pkg
