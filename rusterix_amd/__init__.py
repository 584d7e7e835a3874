"""rusterix_amd -- MI355X (gfx950) back end for Rusterix' tile rasterizer hot path.

`load()` returns the reference-shaped classes (Scene, Batch3D, Batch2D, Assets, Rasterizer, cameras)
bound to the product host library, whose `Rasterizer::rasterize` hands the projected frame across the
C ABI of include/rxr.h to the HIP kernels.  There is no CPU fallback: if the libraries are not built
or no GPU is visible, calls fail loudly.
"""
from .libs import load, load_rxr, lib_paths, rxr_abi, RxrStats  # noqa: F401
