"""Overlay plumbing: a module of this repo that shadows a same-named file of the reference checkout serves the
names it does not replace (host-side metrics, plotting, the single-cloud SOM builder: all off the hot path) from
the reference's OWN file, loaded lazily under a private name -- nothing of it is restated or copied here."""
import importlib.util
import os
import sys
import types

_loaded = {}


def reference_module(package, filename, shadow_file, optional_imports=()):
    """Load ``<next directory on package.__path__>/<filename>`` as ``<package>._reference_<stem>``.

    ``optional_imports``: third-party modules that file imports at top level only for the classes this repo
    replaces (faiss, torchvision); when one is not installed an empty stand-in is registered for the duration of
    the load.  Raises ImportError when no reference checkout sits behind the overlay on sys.path."""
    key = (package, filename)
    if key in _loaded:
        return _loaded[key]
    here = os.path.dirname(os.path.abspath(shadow_file))
    for d in list(sys.modules[package].__path__):
        path = os.path.join(d, filename)
        if os.path.abspath(d) == here or not os.path.isfile(path):
            continue
        name = "%s._reference_%s" % (package, os.path.splitext(filename)[0])
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        stubs = []
        for imp in optional_imports:
            if imp not in sys.modules and importlib.util.find_spec(imp) is None:
                sys.modules[imp] = types.ModuleType(imp)
                stubs.append(imp)
        sys.modules[name] = mod                      # so that the file's own relative imports resolve
        try:
            spec.loader.exec_module(mod)
        except BaseException:
            del sys.modules[name]
            raise
        finally:
            for imp in stubs:
                del sys.modules[imp]
        _loaded[key] = mod
        return mod
    raise ImportError("no reference checkout (%s/%s) behind the overlay on sys.path" % (package, filename))


def delegate(package, filename, shadow_file, optional_imports=()):
    """Module-level ``__getattr__`` (PEP 562) for a shadowing module."""
    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        try:
            return getattr(reference_module(package, filename, shadow_file, optional_imports), name)
        except ImportError as e:
            raise AttributeError("%s.%s is not part of the MI355X hot path and there is %s"
                                 % (package, name, e)) from None
    return __getattr__
