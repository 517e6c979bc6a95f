"""Import alias: ``few-shot-vid2vid_amd/`` is not a valid identifier, so the package is loaded by path."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("few-shot-vid2vid_amd")
sys.modules[__name__] = _pkg
sys.modules.setdefault("fsv2v_amd", _pkg)
