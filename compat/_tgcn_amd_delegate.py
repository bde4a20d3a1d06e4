"""Hand names this repo does not implement over to the reference's own module further down `sys.path`.

`compat/` sits in front of the reference checkout on PYTHONPATH, so `import gcn.graph`
finds compat/gcn/graph.py and the reference's gcn/graph.py of the same name is shadowed.  The scripts the
north star names call `graph.grid / distance_sklearn_metrics / adjacency / laplacian / rescale_L / fourier`
next to `graph.chebyshev` (examples/tgcn_mnist.py:52-54,142,173-175,194; examples/pytorch_based/
pytorch_hcp_tgcn.py:13,52-57), and those helpers are out of scope here: they are
served by the reference's file, loaded under a private module name -- nothing of it is copied into this tree.
"""
import importlib.util
import os
import sys

_COMPAT_ROOT = os.path.dirname(os.path.abspath(__file__))


class ReferenceNotFound(ImportError):
    """The attribute lives in the reference checkout and no such checkout is on sys.path."""


def reference_dirs(package):
    """Directories `<entry>/<package>` of sys.path entries other than compat/, in sys.path order."""
    out = []
    for entry in sys.path:
        base = os.path.abspath(entry or os.getcwd())
        if base == _COMPAT_ROOT:
            continue
        cand = os.path.join(base, *package.split("."))
        if os.path.isfile(os.path.join(cand, "__init__.py")) and cand not in out:
            out.append(cand)
    return out


def find_reference_file(package, name):
    for d in reference_dirs(package):
        path = os.path.join(d, name + ".py")
        if os.path.isfile(path):
            return path
    return None


def load_reference_module(package, name):
    """The reference's `<package>/<name>.py`, executed once under `<package>._reference_<name>`."""
    private = "%s._reference_%s" % (package, name)
    mod = sys.modules.get(private)
    if mod is not None:
        return mod
    path = find_reference_file(package, name)
    if path is None:
        raise ReferenceNotFound(
            "%s.%s: this name is not part of the MI355X drop-in; it is served by the reference's own %s/%s.py, and no "
            "cassianobecker/tgcn checkout was found on sys.path behind %s (put the checkout on PYTHONPATH after compat/, "
            "see INTEGRATION.md section 1)" % (package, name, package.replace(".", "/"), name, _COMPAT_ROOT))
    spec = importlib.util.spec_from_file_location(private, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[private] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules[private]
        raise
    return mod


def install(module_globals, package, name, native):
    """Give the shim module `package.name` a module-level __getattr__ / __dir__ that fall through to the reference.

    `native`: the names this repo provides (they stay what the shim bound them to)."""
    qual = "%s.%s" % (package, name)

    def __getattr__(attr):
        if attr.startswith("__") and attr.endswith("__"):
            raise AttributeError("module %r has no attribute %r" % (qual, attr))
        try:
            ref = load_reference_module(package, name)
        except ReferenceNotFound as e:
            # a MISS for attribute probes -- hasattr(), getattr(mod, name, None), inspect / pytest / pickle.whichmodule scanning sys.modules -- with
            # the cause in the message; `from gcn.graph import grid` turns it into the ImportError a missing name is
            raise AttributeError(str(e)) from e
        try:
            return getattr(ref, attr)
        except AttributeError:
            raise AttributeError("module %r has no attribute %r (neither the MI355X drop-in nor the reference's %s)"
                                 % (qual, attr, ref.__file__)) from None

    def __dir__():
        names = set(module_globals) | set(native)
        path = find_reference_file(package, name)
        if path is not None:
            try:
                names |= set(dir(load_reference_module(package, name)))
            except Exception:
                pass
        return sorted(names)

    module_globals["__getattr__"] = __getattr__
    module_globals["__dir__"] = __dir__
    module_globals["NATIVE_NAMES"] = tuple(native)


def extend_package_path(package_path, package):
    """Let `import <package>.<other>` (gcn.models, gcn.utils ...) find the reference's files of a shadowed package."""
    for d in reference_dirs(package):
        if d not in package_path:
            package_path.append(d)
