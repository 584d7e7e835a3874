import os, sys
sys.path.insert(0, os.getcwd())
os.environ["RXR_SHADER_JIT"] = "1"; os.environ["RXR_JIT_CACHE"] = "0"
import ctypes as C
import rusterix_amd
from rusterix_amd import scenes
prod = rusterix_amd.load()
cfg = scenes.box_grid_scene(prod, n=96, width=1920, height=1080, shader=True)
scenes.render(cfg)
rxr = C.CDLL(rusterix_amd.lib_paths()["rxr"])
rxr.rxr_debug_jit_info.restype = C.c_char_p; rxr.rxr_debug_jit_info.argtypes = [C.c_void_p]
prod.lib.rxh_context.restype = C.c_void_p
print(rxr.rxr_debug_jit_info(prod.lib.rxh_context()).decode())
