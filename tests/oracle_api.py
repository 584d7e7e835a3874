"""Loads the CPU oracle (oracle/librusterix_oracle.so) -- test infrastructure only."""
import ctypes
import os
import subprocess

from rusterix_amd.binding import make_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("RXR_ORACLE_SO") or os.path.join(ORACLE_DIR, "librusterix_oracle.so")  # (tools/sanitize_cpu.sh points this at the ASan build)

_cached = None


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def load_oracle():
    global _cached
    if _cached is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("rusterix_oracle.cpp", "oracle_capi.cpp", "rusterix_oracle.hpp", "rusteria_vm.hpp")]
        if not os.environ.get("RXR_ORACLE_SO") and (not os.path.exists(ORACLE_SO) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs)):
            build_oracle()
        lib = ctypes.CDLL(ORACLE_SO)
        api = make_api(lib, "orc_", "oracle")
        _bind_leaf_functions(api, lib)
        _cached = api
    return _cached


def _bind_leaf_functions(api, lib):
    import ctypes as C

    import numpy as np

    from rusterix_amd.binding import RxrLight

    pf, pb = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    lib.orc_hash_u32.restype = C.c_uint32
    lib.orc_hash_u32.argtypes = [C.c_uint32]
    lib.orc_pixel_to_vec4.argtypes = [pb, pf]
    lib.orc_vec4_to_pixel.argtypes = [pf, pb]
    lib.orc_srgb_to_linear_fast.restype = C.c_float
    lib.orc_srgb_to_linear_fast.argtypes = [C.c_float]
    lib.orc_linear_to_srgb_fast.restype = C.c_float
    lib.orc_linear_to_srgb_fast.argtypes = [C.c_float]
    lib.orc_texture_sample.argtypes = [pb, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int, pb]
    lib.orc_light_color_at.restype = C.c_int
    lib.orc_light_color_at.argtypes = [C.POINTER(RxrLight), pf, C.c_uint32, C.c_int, pf]
    lib.orc_light_radiance_at.restype = C.c_int
    lib.orc_light_radiance_at.argtypes = [C.POINTER(RxrLight), pf, pf, C.c_uint32, pf]
    lib.orc_edges_new.argtypes = [pf, pf, pf]
    lib.orc_edges_evaluate.restype = C.c_int
    lib.orc_edges_evaluate.argtypes = [pf, C.c_float, C.c_float]
    lib.orc_mat4_inverted.argtypes = [pf, pf]
    lib.orc_mat4_mul_vec4.argtypes = [pf, pf, pf]
    lib.orc_vm_shade.restype = C.c_int
    lib.orc_vm_shade.argtypes = [C.c_void_p, C.c_void_p, C.c_int, pf]

    FIELDS = ["uv", "color", "roughness", "metallic", "emissive", "opacity", "bump", "normal", "hitpoint", "time"]

    def vm_shade(scene, assets, program_index, **inputs):
        """one Execution::shade on a fresh Execution; returns dict field -> (x, y, z), or None on a fault"""
        f = np.zeros((10, 3), np.float32)
        f[2] = 0.5  # Execution::new: roughness = broadcast(0.5)
        for k, v in inputs.items():
            f[FIELDS.index(k)] = v
        rc = lib.orc_vm_shade(scene._h, assets._h, program_index, f.ctypes.data_as(pf))
        if rc != 0:
            return None
        return {k: tuple(float(x) for x in f[i]) for i, k in enumerate(FIELDS)}

    api.vm_shade = vm_shade
    lib.orc_rasterizer_set_threads.argtypes = [C.c_void_p, C.c_int]
    lib.orc_rasterizer_get_threads.restype = C.c_int
    lib.orc_rasterizer_get_threads.argtypes = [C.c_void_p]

    def fp(a):
        return a.ctypes.data_as(pf)

    def hash_u32(seed):
        return lib.orc_hash_u32(seed)

    def vec4_to_pixel(v):
        v = np.ascontiguousarray(v, np.float32)
        out = np.zeros(4, np.uint8)
        lib.orc_vec4_to_pixel(fp(v), out.ctypes.data_as(pb))
        return out

    def pixel_to_vec4(p):
        p = np.ascontiguousarray(p, np.uint8)
        out = np.zeros(4, np.float32)
        lib.orc_pixel_to_vec4(p.ctypes.data_as(pb), fp(out))
        return out

    def texture_sample(tex, u, v, sample_mode, repeat_mode):
        out = np.zeros(4, np.uint8)
        lib.orc_texture_sample(tex.data.ctypes.data_as(pb), tex.width, tex.height, u, v, sample_mode, repeat_mode,
                               out.ctypes.data_as(pb))
        return out

    def light_color_at(light, point, hash_, d2=False):
        p = np.ascontiguousarray(point, np.float32)
        out = np.zeros(3, np.float32)
        ok = lib.orc_light_color_at(C.byref(light), fp(p), hash_, int(d2), fp(out))
        return out if ok else None

    def light_radiance_at(light, point, normal, hash_):
        p = np.ascontiguousarray(point, np.float32)
        n = np.ascontiguousarray(normal, np.float32) if normal is not None else None
        out = np.zeros(3, np.float32)
        ok = lib.orc_light_radiance_at(C.byref(light), fp(p), fp(n) if n is not None else None, hash_, fp(out))
        return out if ok else None

    def edges_new(v0, v1):
        a = np.ascontiguousarray(v0, np.float32).reshape(6)
        b = np.ascontiguousarray(v1, np.float32).reshape(6)
        out = np.zeros(9, np.float32)
        lib.orc_edges_new(fp(a), fp(b), fp(out))
        return out

    def edges_evaluate(abc, px, py):
        abc = np.ascontiguousarray(abc, np.float32)
        return bool(lib.orc_edges_evaluate(fp(abc), px, py))

    def mat4_inverted(m):
        m = np.ascontiguousarray(m, np.float32)
        out = np.zeros(16, np.float32)
        lib.orc_mat4_inverted(fp(m), fp(out))
        return out

    def mat4_mul_vec4(m, v):
        m = np.ascontiguousarray(m, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        out = np.zeros(4, np.float32)
        lib.orc_mat4_mul_vec4(fp(m), fp(v), fp(out))
        return out

    def set_threads(rasterizer, n):
        lib.orc_rasterizer_set_threads(rasterizer._h, n)
        return rasterizer

    api.hash_u32 = hash_u32
    api.vec4_to_pixel = vec4_to_pixel
    api.pixel_to_vec4 = pixel_to_vec4
    api.srgb_to_linear_fast = lambda x: lib.orc_srgb_to_linear_fast(x)
    api.linear_to_srgb_fast = lambda x: lib.orc_linear_to_srgb_fast(x)
    api.texture_sample = texture_sample
    api.light_color_at = light_color_at
    api.light_radiance_at = light_radiance_at
    api.edges_new = edges_new
    api.edges_evaluate = edges_evaluate
    api.mat4_inverted = mat4_inverted
    api.mat4_mul_vec4 = mat4_mul_vec4
    api.set_threads = set_threads
    api.get_threads = lambda r: lib.orc_rasterizer_get_threads(r._h)
