"""Overlay package: ``models.layers``, ``models.operations`` and ``models.networks`` (the hot path)
live here together with the loss classes of the heads (``models.losses``: ChamferLoss without faiss, the rest
delegated to the reference file); ``models.classifier`` / ``segmenter`` / ``autoencoder`` are resolved from a
reference checkout found later on sys.path and run unchanged on top."""
import os as _os
import sys as _sys

for _p in list(_sys.path):
    _d = _os.path.join(_p or ".", __name__)
    if _os.path.isdir(_d) and _os.path.abspath(_d) not in [_os.path.abspath(q) for q in __path__]:
        __path__.append(_d)
