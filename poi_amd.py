"""Importable alias of the package directory `point-of-interest-recommendation_amd/` (a hyphen is not
a valid identifier character, so `import point-of-interest-recommendation_amd` cannot be written).
`import poi_amd`, `from poi_amd import data`, `from poi_amd.models import OboSpatialGru` all resolve to
the ONE real module object of the package (no duplicate class identities)."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL = "point-of-interest-recommendation_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real

    def create_module(self, spec):
        return importlib.import_module(self.real)

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith("poi_amd."):
            real = _REAL + fullname[len("poi_amd"):]
            return importlib.util.spec_from_loader(fullname, _AliasLoader(real))
        return None


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
