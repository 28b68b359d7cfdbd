"""Importable alias of the package directory `point-of-interest-recommendation_amd/` (a hyphen is not
a valid identifier character, so `import point-of-interest-recommendation_amd` cannot be written)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("point-of-interest-recommendation_amd")
sys.modules[__name__] = _pkg
