"""Import shim for the *reference* hot-path module (container-only tooling).

Used ONLY by ``make_golden.py`` (fixture generation) and by the optional
``tests/test_oracle_vs_reference.py`` cross-check, both of which are skipped
when ``/root/reference`` is absent (i.e. always on the GPU box).  Nothing from
the reference is copied: the module is imported from where it lies.

Recipe: ``anndata``/``scanpy`` are not installed and are only
needed by the reference for a type annotation and ``logging.warning``; register
stub modules, register an empty ``infercnvpy`` package whose ``__path__`` points
at the reference tree (so its ``__init__`` is never executed) and load
``_util.py`` and ``tl/_infercnv.py`` by file location.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("INFERCNVPY_REFERENCE", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src", "infercnvpy")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_SRC, "tl", "_infercnv.py"))


def _stub(name: str, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    """Return (ref_infercnv_module, ref_scores_module)."""
    if not reference_available():
        raise RuntimeError("reference tree not present")

    class _AnnData:  # annotation-only stand-in
        pass

    _stub("anndata", AnnData=_AnnData)
    log = types.SimpleNamespace(warning=lambda *a, **k: None, info=lambda *a, **k: None)
    _stub("scanpy", logging=log)
    if "infercnvpy" not in sys.modules or not hasattr(sys.modules["infercnvpy"], "__path__"):
        pkg = types.ModuleType("infercnvpy")
        pkg.__path__ = [REF_SRC]
        sys.modules["infercnvpy"] = pkg
        tl = types.ModuleType("infercnvpy.tl")
        tl.__path__ = [os.path.join(REF_SRC, "tl")]
        sys.modules["infercnvpy.tl"] = tl

    def _load(modname, relpath):
        if modname in sys.modules and getattr(sys.modules[modname], "__file__", None):
            return sys.modules[modname]
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_SRC, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    _load("infercnvpy._util", "_util.py")
    ref = _load("infercnvpy.tl._infercnv", os.path.join("tl", "_infercnv.py"))
    scores = _load("infercnvpy.tl._scores", os.path.join("tl", "_scores.py"))
    return ref, scores
