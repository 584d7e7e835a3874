"""Cost of Rusteria programs on the device: the reduced box grid (1920x1080, 110 592 triangles) without a program, with an
empty one and with programs of growing length -- kernel-level overhead of k_raster_vm and cost per VM instruction."""
import sys, time, ctypes as C
sys.path.insert(0,'.')
import numpy as np
import rusterix_amd
from rusterix_amd import scenes, binding as B
prod = rusterix_amd.load()
host = prod.lib
rxr = C.CDLL(rusterix_amd.lib_paths()["rxr"])
host.rxh_context.restype = C.c_void_p
host.rxh_rasterizer_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
rxr.rxr_render_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
rxr.rxr_synchronize.argtypes = [C.c_void_p]
rxr.rxr_profile_begin.argtypes = [C.c_void_p, C.c_uint32]
rxr.rxr_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]
def run(name, prog):
    orig = scenes.box_grid_shader
    scenes.box_grid_shader = lambda: prog
    cfg = scenes.box_grid_scene(prod, n=96, width=1920, height=1080, shader=prog is not None)
    scenes.box_grid_shader = orig
    r = cfg.setup()
    assert host.rxh_rasterizer_upload(r._h, cfg.scene._h, cfg.width, cfg.height, cfg.tile_size, cfg.assets._h) == 0
    ctx = host.rxh_context()
    for _ in range(3): rxr.rxr_render_rows(ctx, 0, cfg.height)
    rxr.rxr_synchronize(ctx)
    n=20
    rxr.rxr_profile_begin(ctx, n)
    for _ in range(n): rxr.rxr_render_rows(ctx, 0, cfg.height)
    rxr.rxr_synchronize(ctx)
    su=(C.c_float*n)(); ru=(C.c_float*n)(); k=C.c_uint32()
    rxr.rxr_profile_read(ctx, su, ru, n, C.byref(k))
    print(f"{name:28s} raster {np.median(ru[:k.value]):8.1f} us")
P=B.Program
run("no program", None)
import os

os.environ.setdefault("RXR_SHADER_JIT", "0")  # (measurements name their mode: interpreted unless asked otherwise)
if os.environ.get("ONLY_NOPROG"): sys.exit(0)
run("empty shade", P([[]]))
run("1 op (Color SetColor)", P([["Color","SetColor"]]))
run("10 adds", P([["Color"]+[("Push",0.01),"Add"]*10+["SetColor"]]))
run("40 adds", P([["Color"]+[("Push",0.01),"Add"]*40+["SetColor"]]))
run("C5 shader", scenes.box_grid_shader())
