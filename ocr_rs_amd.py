"""Import alias: ``import ocr_rs_amd`` loads the package directory ``ocr-rs_amd/``
(a hyphen is not a valid identifier, importlib accepts it)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("ocr-rs_amd")
sys.modules[__name__] = _pkg
