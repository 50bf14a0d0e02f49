"""Import helper: the package directory is `seq-align-gpu_amd/` (hyphenated, as the
build contract names it), which Python cannot import by name.  `load()` imports
it as module `seq_align_gpu_amd`; `oracle()` imports oracle/oracle.py (test
infrastructure only -- the product never touches it)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def _load(name, path, pkg_dir=None):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(
        name, path, submodule_search_locations=[pkg_dir] if pkg_dir else None)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(name, None)
        raise
    return mod


def load():
    d = os.path.join(ROOT, "seq-align-gpu_amd")
    return _load("seq_align_gpu_amd", os.path.join(d, "__init__.py"), d)


def build_module():
    d = os.path.join(ROOT, "seq-align-gpu_amd")
    return _load("seq_align_gpu_amd_build", os.path.join(d, "build.py"))


def oracle():
    return _load("swg_oracle", os.path.join(ROOT, "oracle", "oracle.py"))
