"""MI355X-native next-POI training / evaluation hot path.

The directory name carries a hyphen, so import it through the `poi_amd` shim at the repo root
(`import poi_amd`) or `importlib.import_module("point-of-interest-recommendation_amd")`.
Nothing here imports `oracle/` - that is test infrastructure.
"""
from . import _lib, build, data            # noqa: F401
from ._lib import PoiError                  # noqa: F401

__all__ = ["_lib", "build", "data", "models", "PoiError", "OboSpatialGru", "OboGru", "OboBpr", "OboCARNN", "Gru"]


def __getattr__(name):
    # models needs torch; keep `import poi_amd` light for the build step
    if name in ("models", "OboSpatialGru", "OboGru", "OboBpr", "OboCARNN", "Gru", "GruBasic", "MfBasic", "evaluate", "harness", "dist"):
        import importlib
        if name in ("models", "evaluate", "harness", "dist"):
            return importlib.import_module("." + name, __name__)
        return getattr(importlib.import_module(".models", __name__), name)
    raise AttributeError(name)
