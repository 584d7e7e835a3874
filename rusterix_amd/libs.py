"""Locates and loads the in-tree native libraries (built by __graft_entry__.build())."""
import ctypes
import os

from .binding import make_api

_HERE = os.path.dirname(os.path.abspath(__file__))
_cached = None


def lib_paths():
    return dict(rxr=os.path.join(_HERE, "csrc", "librxr_hip.so"), host=os.path.join(_HERE, "csrc", "librusterix_host.so"))


def load_rxr():
    """The C-ABI device library (include/rxr.h)."""
    p = lib_paths()["rxr"]
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first "
                           "(there is no CPU fallback)")
    return ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)


def load():
    global _cached
    if _cached is None:
        load_rxr()
        p = lib_paths()["host"]
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: run __graft_entry__.build() first (there is no CPU fallback)")
        lib = ctypes.CDLL(p)
        _cached = make_api(lib, "rxh_", "product")
    return _cached
