"""Locates and loads the in-tree native libraries (built by __graft_entry__.build())."""
import ctypes
import os

from .binding import make_api

_HERE = os.path.dirname(os.path.abspath(__file__))
_cached = None


def lib_paths():
    # RXR_HOST_SO / RXR_DEVICE_SO: other builds of the two libraries (tools/sanitize_cpu.sh: host code under AddressSanitizer)
    return dict(rxr=os.environ.get("RXR_DEVICE_SO") or os.path.join(_HERE, "csrc", "librxr_hip.so"),
                host=os.environ.get("RXR_HOST_SO") or os.path.join(_HERE, "csrc", "librusterix_host.so"))


def load_rxr():
    """The C-ABI device library (include/rxr.h)."""
    p = lib_paths()["rxr"]
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first "
                           "(there is no CPU fallback)")
    return ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)


class RxrStats(ctypes.Structure):
    """rxr_stats (include/rxr.h)."""

    _fields_ = [("setup_us", ctypes.c_float), ("raster_us", ctypes.c_float), ("total_us", ctypes.c_float), ("n_triangles3d", ctypes.c_uint32),
                ("n_triangles2d", ctypes.c_uint32), ("n_bin_entries", ctypes.c_uint32), ("tiles_x", ctypes.c_uint32), ("tiles_y", ctypes.c_uint32)]


_rxr_typed = None


def rxr_abi():
    """The C-ABI library with the argument types of the split-phase entry points declared (include/rxr.h), for callers that
    drive a context themselves after `rxh_rasterizer_upload` (bench.py, tools/, tests/)."""
    global _rxr_typed
    if _rxr_typed is not None:
        return _rxr_typed
    C = ctypes
    L = load_rxr()
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    sig = {
        "rxr_create": (i32, [C.POINTER(vp), i32]),
        "rxr_create_multi": (i32, [C.POINTER(vp), C.POINTER(i32), i32]),
        "rxr_destroy": (None, [vp]),
        "rxr_member_count": (i32, [vp]),
        "rxr_member": (vp, [vp, i32]),
        "rxr_last_error": (C.c_char_p, [vp]),
        "rxr_device_count": (i32, []),
        "rxr_pin_host_buffer": (i32, [vp, vp, C.c_size_t]),
        "rxr_unpin_host_buffer": (i32, [vp, vp]),
        "rxr_render_rows": (i32, [vp, u32, u32]),
        "rxr_render_rows_to": (i32, [vp, u32, u32, vp, vp]),
        "rxr_render_stripes_to": (i32, [vp, u32, u32, vp, vp]),
        "rxr_render_stripes_batch": (i32, [vp, u32, u32, u32, vp, C.c_size_t, vp]),
        "rxr_render_gather": (i32, [vp, i32, vp, vp]),
        "rxr_stream_begin": (i32, [vp, u32, C.POINTER(u32), C.POINTER(u32)]),
        "rxr_stream_batch3d": (i32, [vp, u32, vp]),
        "rxr_debug_stream_info": (i32, [vp]),
        "rxr_debug_rerenders": (u32, [vp]),
        "rxr_render_download": (i32, [vp, vp]),
        "rxr_download_rows": (i32, [vp, vp, u32, u32]),
        "rxr_synchronize": (i32, [vp]),
        "rxr_get_stats": (i32, [vp, C.POINTER(RxrStats)]),
        "rxr_device_framebuffer": (vp, [vp]),
        "rxr_set_light_math": (i32, [vp, i32]),
        "rxr_profile_begin": (i32, [vp, u32]),
        "rxr_profile_stride": (i32, [vp, u32]),
        "rxr_profile_read": (i32, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), u32, C.POINTER(u32)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _rxr_typed = L
    return L


def load():
    global _cached
    if _cached is None:
        load_rxr()
        p = lib_paths()["host"]
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: run __graft_entry__.build() first (there is no CPU fallback)")
        lib = ctypes.CDLL(p)
        lib.rxh_context.restype = ctypes.c_void_p
        lib.rxh_last_error.restype = ctypes.c_char_p
        lib.rxh_set_device.argtypes = [ctypes.c_int]
        lib.rxh_set_devices.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        lib.rxh_set_device_projection.argtypes = [ctypes.c_int]
        lib.rxh_set_device_edges.argtypes = [ctypes.c_int]
        lib.rxh_set_light_math_exact.argtypes = [ctypes.c_int]
        lib.rxh_rasterizer_upload.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        _cached = make_api(lib, "rxh_", "product")
    return _cached
