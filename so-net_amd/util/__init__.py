"""Overlay package: the hot-path module ``util.som`` lives here; every other ``util.*`` module
(visualizer, potential_field, ...) is resolved from a reference checkout found later on sys.path."""
import os as _os
import sys as _sys

for _p in list(_sys.path):
    _d = _os.path.join(_p or ".", __name__)
    if _os.path.isdir(_d) and _os.path.abspath(_d) not in [_os.path.abspath(q) for q in __path__]:
        __path__.append(_d)
