"""ctypes binder for the handle-based builder C API (`<prefix>scene_new`, `<prefix>batch3d_from_box`, ...).

The product host library (rusterix_amd/csrc/host, prefix ``rxh_``) exports this API over its C++
mirror of the reference's Scene / Batch2D / Batch3D / Assets / Rasterizer types; the classes made by
:func:`make_api` give Python the same names and builder methods as the reference
(reference src/scene.rs, src/batch/batch3d.rs, src/batch/batch2d.rs, src/rasterizer.rs:92-193), so the
parity tests read like the reference's examples (examples/cube.rs:30-94).

tests/ binds the same classes to the CPU oracle (prefix ``orc_``); the product never does.
"""
from __future__ import annotations

import ctypes as C
import types

import numpy as np

# ---- enums (include/rxr.h) ---------------------------------------------------------------------
SAMPLE_NEAREST, SAMPLE_LINEAR = 0, 1
REPEAT_CLAMP_XY, REPEAT_REPEAT_XY, REPEAT_REPEAT_X, REPEAT_REPEAT_Y = 0, 1, 2, 3
MODE_TRIANGLES, MODE_LINES, MODE_LINE_STRIP, MODE_LINE_LOOP = 0, 1, 2, 3
CULL_OFF, CULL_FRONT, CULL_BACK = 0, 1, 2
LIGHT_POINT, LIGHT_AMBIENT, LIGHT_AMBIENT_DAYLIGHT, LIGHT_SPOT, LIGHT_AREA, LIGHT_DAYLIGHT = range(6)
SOURCE_OTHER, SOURCE_STATIC_TILE, SOURCE_DYNAMIC_TILE, SOURCE_PIXEL, SOURCE_TERRAIN, SOURCE_MISSING = range(6)
HOST_SOURCE_ENTITY_TILE, HOST_SOURCE_ITEM_TILE = 64, 65  # never cross the ABI: resolved by the host (include/rxr.h)
LIST_CHUNK_OPACITY, LIST_CHUNK, LIST_CHUNK_TERRAIN, LIST_STATIC, LIST_DYNAMIC, LIST_OVERLAY = range(6)
BG_NONE, BG_VGRADIENT, BG_HOST_PIXELS, BG_GRID = 0, 1, 2, 3

RXR_OK, RXR_ERR_INVALID, RXR_ERR_NO_DEVICE, RXR_ERR_HIP, RXR_ERR_UNSUPPORTED, RXR_ERR_OOM = 0, -1, -2, -3, -4, -5


class RxrLight(C.Structure):
    """rxr_light == CompiledLight (reference src/map/light.rs:456-477)."""

    _fields_ = [
        ("light_type", C.c_uint32),
        ("position", C.c_float * 3),
        ("color", C.c_float * 3),
        ("intensity", C.c_float),
        ("emitting", C.c_uint32),
        ("start_distance", C.c_float),
        ("end_distance", C.c_float),
        ("flicker", C.c_float),
        ("direction", C.c_float * 3),
        ("cone_angle", C.c_float),
        ("normal", C.c_float * 3),
        ("width", C.c_float),
        ("height", C.c_float),
        ("from_linedef", C.c_uint32),
    ]


class RasterizeError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"rasterize failed with status {code}: {msg}")
        self.code = code


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _u32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _bp(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


# ---- small vek-like helpers for callers (column-major float32[16], m[c*4+r]) ---------------------
class Mat4:
    @staticmethod
    def identity():
        return np.eye(4, dtype=np.float32).T.reshape(16).copy()

    @staticmethod
    def scaling_3d(v):
        m = np.eye(4, dtype=np.float32)
        m[0, 0], m[1, 1], m[2, 2] = v
        return np.ascontiguousarray(m.T).reshape(16)

    @staticmethod
    def translation_3d(v):
        m = np.eye(4, dtype=np.float32)
        m[0, 3], m[1, 3], m[2, 3] = v
        return np.ascontiguousarray(m.T).reshape(16)

    @staticmethod
    def from_rows(rows):
        return np.ascontiguousarray(np.asarray(rows, dtype=np.float32).reshape(4, 4).T).reshape(16)


class Mat3:
    @staticmethod
    def from_rows(rows):
        return np.ascontiguousarray(np.asarray(rows, dtype=np.float32).reshape(3, 3).T).reshape(9)


class PixelSource:
    """reference src/map/pixelsource.rs:23-37 (variants the raster loops distinguish)."""

    def __init__(self, kind, index=0, pixel=(0, 0, 0, 0)):
        self.kind, self.index, self.pixel = kind, index, tuple(pixel)

    Off = None  # filled below

    @staticmethod
    def StaticTileIndex(i):
        return PixelSource(SOURCE_STATIC_TILE, i)

    @staticmethod
    def DynamicTileIndex(i):
        return PixelSource(SOURCE_DYNAMIC_TILE, i)

    @staticmethod
    def Pixel(rgba):
        return PixelSource(SOURCE_PIXEL, 0, rgba)

    @staticmethod
    def Terrain():
        return PixelSource(SOURCE_TERRAIN)

    @staticmethod
    def EntityTile(entity_id, index):
        """PixelSource::EntityTile(id, index): assets.entity_tiles[id].get_index(index) (reference src/rasterizer.rs:1140-1163)"""
        s = PixelSource(HOST_SOURCE_ENTITY_TILE, entity_id)
        s.seq = index
        return s

    @staticmethod
    def ItemTile(item_id, index):
        s = PixelSource(HOST_SOURCE_ITEM_TILE, item_id)
        s.seq = index
        return s

    @staticmethod
    def Missing():
        return PixelSource(SOURCE_MISSING)


PixelSource.Off = PixelSource(SOURCE_OTHER)


class RenderMode:
    """reference src/rendermode.rs."""

    def __init__(self, d2=True, d3=True, ignore_bg=False):
        self.d2_active, self.d3_active, self.ignore_background_shader_flag = d2, d3, ignore_bg

    @staticmethod
    def render_all():
        return RenderMode(True, True)

    @staticmethod
    def render_2d():
        return RenderMode(True, False)

    @staticmethod
    def render_3d():
        return RenderMode(False, True)

    def ignore_background_shader(self, value):
        self.ignore_background_shader_flag = bool(value)
        return self


class Texture:
    """reference src/texture.rs:46-54: RGBA8 row-major."""

    def __init__(self, data, width, height):
        self.data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        assert self.data.size == width * height * 4, "Invalid texture data size."
        self.width, self.height = int(width), int(height)


class Tile:
    """reference src/map/tile.rs (`textures: Vec<Texture>`)."""

    def __init__(self, textures):
        self.textures = list(textures)

    @staticmethod
    def from_texture(tex):
        return Tile([tex])


class Light:
    """reference src/map/light.rs:30-245: property bag + compile() defaults."""

    def __init__(self, light_type=LIGHT_POINT):
        self.light_type = light_type
        self.position = (0.0, 0.0, 0.0)
        self.color = (1.0, 1.0, 1.0)
        self.intensity = 1.0
        self.start_distance = 1.0
        self.end_distance = 2.0
        self.flicker = 0.0
        self.direction = (0.0, 0.0, -1.0)
        self.cone_angle = float(np.float32(np.pi / 4))
        self.normal = (0.0, 1.0, 0.0)
        self.width = 1.0
        self.height = 1.0
        self.emitting = True
        self.from_linedef = False

    def with_position(self, p):
        self.position = tuple(p)
        return self

    def with_color(self, c):
        self.color = tuple(c)
        return self

    def with_intensity(self, i):
        self.intensity = i
        return self

    def with_start_distance(self, s):
        self.start_distance = s
        return self

    def with_end_distance(self, e):
        self.end_distance = e
        return self

    def with_flicker(self, f):
        self.flicker = f
        return self

    def compile(self):
        def norm3(v):
            v = np.asarray(v, dtype=np.float32)
            m = np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1]) + np.float32(v[2] * v[2]), dtype=np.float32)
            return tuple(np.float32(v / m))

        l = RxrLight()
        l.light_type = self.light_type
        l.position[:] = self.position
        l.color[:] = self.color
        l.intensity = self.intensity
        l.emitting = 1 if self.emitting else 0
        l.start_distance = self.start_distance
        l.end_distance = self.end_distance
        l.flicker = self.flicker
        l.direction[:] = norm3(self.direction)
        l.cone_angle = self.cone_angle
        l.normal[:] = norm3(self.normal)
        l.width = self.width
        l.height = self.height
        l.from_linedef = 1 if self.from_linedef else 0
        return l


class VGrayGradientShader:
    """reference src/shader/vgradient.rs"""

    kind = BG_VGRADIENT


class GridShader:
    """reference src/shader/grid.rs: `Shader::new()` defaults (:12-16) and the two parameter setters (:19-34)"""

    kind = BG_GRID

    def __init__(self):
        self.grid_size, self.subdivisions, self.offset = 30.0, 2.0, (0.0, 0.0)

    def set_parameter_f32(self, key, value):
        if key == "grid_size":
            self.grid_size = float(value)
        elif key == "subdivisions":
            self.subdivisions = float(value)
        return self

    def set_parameter_vec2(self, key, value):
        if key == "offset":
            self.offset = (float(value[0]), float(value[1]))
        return self


# ---- Rusteria programs (reference rusteria/src/node/{nodeop,program}.rs) -------------------------------
# NodeOp variants in declaration order = the RXR_NODE_* opcodes of include/rxr.h
NODE_OPS = [
    "LoadGlobal", "StoreGlobal", "LoadLocal", "StoreLocal", "Swap", "GetComponents", "SetComponents", "If", "For", "Push",
    "FunctionCall", "Return", "Dup", "Clear", "Pack2", "Pack3", "Add", "Sub", "Mul", "Div", "Length", "Length2", "Length3",
    "Abs", "Sin", "Sin1", "Sin2", "Cos", "Cos1", "Cos2", "Tan", "Atan", "Atan2", "Rotate2D", "Dot", "Dot2", "Dot3", "Cross",
    "Normalize", "Floor", "Ceil", "Round", "Fract", "Mod", "Degrees", "Radians", "Min", "Max", "Mix", "Smoothstep", "Step",
    "Clamp", "Sqrt", "Pow", "Log", "Print", "Eq", "Ne", "Lt", "Le", "Gt", "Ge", "And", "Or", "Not", "Neg", "UV", "SetUV",
    "Normal", "SetNormal", "Hitpoint", "Time", "Sample", "SampleNormal", "Color", "SetColor", "Roughness", "SetRoughness",
    "Metallic", "SetMetallic", "Emissive", "SetEmissive", "Opacity", "SetOpacity", "Bump", "SetBump", "Alloc", "Iterate",
    "Save", "PaletteIndex",
]
NODE_OPCODE = {n: i for i, n in enumerate(NODE_OPS)}


def assemble(ops):
    """NodeOp tree -> the word serialisation of include/rxr.h.  An op is a name ("Add") or a tuple:
    ("Push", x, y, z) | ("Push", x) (splat) | ("LoadLocal", i) | ("GetComponents", [0, 1]) |
    ("If", then_ops, else_ops_or_None) | ("For", init, cond, incr, body) | ("FunctionCall", arity, total_locals, index)."""
    out = []
    for op in ops:
        if isinstance(op, str):
            op = (op,)
        name, args = op[0], op[1:]
        code = NODE_OPCODE[name]
        out.append(code)
        if name in ("LoadGlobal", "StoreGlobal", "LoadLocal", "StoreLocal"):
            out.append(int(args[0]))
        elif name in ("GetComponents", "SetComponents"):
            out.append(len(args[0]))
            out.extend(int(c) for c in args[0])
        elif name == "If":
            t = assemble(args[0])
            e = assemble(args[1]) if len(args) > 1 and args[1] is not None else []
            out.extend([len(t), 1 if (len(args) > 1 and args[1] is not None) else 0, len(e)])
            out.extend(t)
            out.extend(e)
        elif name == "For":
            blocks = [assemble(b) for b in args]
            assert len(blocks) == 4
            out.extend(len(b) for b in blocks)
            for b in blocks:
                out.extend(b)
        elif name == "Push":
            v = args if len(args) == 3 else (args[0],) * 3
            out.extend(int(x) for x in np.asarray(v, np.float32).view(np.uint32))
        elif name == "FunctionCall":
            out.extend(int(a) for a in args)
        else:
            assert not args, f"{name} takes no payload"
    return out


class Program:
    """reference rusteria/src/node/program.rs:7-29: user functions (NodeOp trees), the index of `shade`, its
    local count and the number of globals.  There is no parser / compiler here (out of scope): programs are
    given as op lists, see `assemble`."""

    def __init__(self, functions, shade_index=0, shade_locals=0, globals=0):
        self.functions = [assemble(f) for f in functions]
        self.shade_index = -1 if shade_index is None else int(shade_index)
        self.shade_locals = int(shade_locals)
        self.globals = int(globals)


def pinned_pixels(rxr, nbytes):
    """a uint8 array of `nbytes` in page-locked, device-readable memory from the library's own allocator (rxr_alloc_pinned: nothing of
    the malloc heap is locked) and the function that frees it; (None, None) when there is no such memory"""
    rxr.rxr_alloc_pinned.restype = C.c_void_p
    rxr.rxr_alloc_pinned.argtypes = [C.c_size_t]
    rxr.rxr_free_pinned.argtypes = [C.c_void_p]
    p = rxr.rxr_alloc_pinned(nbytes)
    if not p:
        return None, None
    arr = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p))
    return arr, (lambda: rxr.rxr_free_pinned(C.c_void_p(p)))


def make_api(lib: C.CDLL, prefix: str, name: str):
    """Build Scene/Batch3D/... classes bound to `lib`'s `<prefix>*` entry points."""

    def fn(sym, restype, *argtypes):
        f = getattr(lib, prefix + sym)
        f.restype = restype
        f.argtypes = list(argtypes)
        return f

    vp, u32, i32, f32, u64 = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_uint64
    pf, pu, pb = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    ppb = C.POINTER(pb)

    L = types.SimpleNamespace(
        scene_new=fn("scene_new", vp),
        scene_free=fn("scene_free", None, vp),
        scene_set_animation_frame=fn("scene_set_animation_frame", None, vp, u64),
        scene_set_background=fn("scene_set_background", None, vp, i32),
        scene_set_background_grid=fn("scene_set_background_grid", None, vp, f32, f32, f32, f32),
        scene_add_light=fn("scene_add_light", None, vp, C.POINTER(RxrLight), i32),
        scene_add_dynamic_tile=fn("scene_add_dynamic_tile", None, vp, ppb, pu, pu, u32),
        scene_add_chunk=fn("scene_add_chunk", i32, vp),
        chunk_add_occluder=fn("chunk_add_occluder", None, vp, i32, f32, f32, f32, f32, f32),
        chunk_add_light=fn("chunk_add_light", None, vp, i32, C.POINTER(RxrLight)),
        chunk_set_terrain=fn("chunk_set_terrain", None, vp, i32, pb, u32, u32, i32, i32, i32),
        chunk_set_terrain_batch2d=fn("chunk_set_terrain_batch2d", None, vp, i32, vp),
        chunk_add_shader_texture=fn("chunk_add_shader_texture", None, vp, i32, pb, u32, u32),
        scene_num_dynamic_lights=fn("scene_num_dynamic_lights", u32, vp),
        scene_add_program=fn("scene_add_program", i32, vp, i32, u32, i32, u32, C.POINTER(pu), pu, u32),
        assets_set_patterns=fn("assets_set_patterns", None, vp, i32, C.POINTER(pf), pu, pu, u32),
        assets_set_palette=fn("assets_set_palette", None, vp, pf, pb, u32),
        batch3d_new=fn("batch3d_new", vp, pf, u32, pu, u32, pf),
        batch3d_from_box=fn("batch3d_from_box", vp, f32, f32, f32, f32, f32, f32),
        batch3d_from_obj=fn("batch3d_from_obj", vp, C.c_char_p),
        batch3d_free=fn("batch3d_free", None, vp),
        batch3d_add=fn("batch3d_add", None, vp, pf, u32, pu, u32, pf),
        batch3d_set_normals=fn("batch3d_set_normals", None, vp, pf, u32),
        batch3d_compute_vertex_normals=fn("batch3d_compute_vertex_normals", None, vp),
        batch3d_set_source=fn("batch3d_set_source", None, vp, u32, u32, pb),
        batch3d_set_source_seq=fn("batch3d_set_source_seq", None, vp, i32, u32, u32),
        batch2d_set_source_seq=fn("batch2d_set_source_seq", None, vp, i32, u32, u32),
        assets_add_sequence_id=fn("assets_add_sequence_id", None, vp, i32, u32),
        assets_add_sequence_tile=fn("assets_add_sequence_tile", None, vp, i32, u32, ppb, pu, pu, u32),
        batch3d_set_repeat_mode=fn("batch3d_set_repeat_mode", None, vp, i32),
        batch3d_set_cull_mode=fn("batch3d_set_cull_mode", None, vp, i32),
        batch3d_set_ambient_color=fn("batch3d_set_ambient_color", None, vp, f32, f32, f32),
        batch3d_set_transform=fn("batch3d_set_transform", None, vp, pf),
        batch3d_set_profile_id=fn("batch3d_set_profile_id", None, vp, i32, u32),
        batch3d_set_shader=fn("batch3d_set_shader", None, vp, i32),
        batch3d_counts=fn("batch3d_counts", None, vp, pu, pu),
        batch3d_num_normals=fn("batch3d_num_normals", u32, vp),
        batch3d_get_geometry=fn("batch3d_get_geometry", None, vp, pf, pu, pf, pf),
        scene_push_batch3d=fn("scene_push_batch3d", i32, vp, vp, i32, i32),
        batch2d_new=fn("batch2d_new", vp, pf, u32, pu, u32, pf),
        batch2d_from_rectangle=fn("batch2d_from_rectangle", vp, f32, f32, f32, f32),
        batch2d_free=fn("batch2d_free", None, vp),
        batch2d_set_mode=fn("batch2d_set_mode", None, vp, i32),
        batch2d_set_repeat_mode=fn("batch2d_set_repeat_mode", None, vp, i32),
        batch2d_set_source=fn("batch2d_set_source", None, vp, u32, u32, pb),
        batch2d_set_receives_light=fn("batch2d_set_receives_light", None, vp, i32),
        batch2d_set_shader=fn("batch2d_set_shader", None, vp, i32),
        scene_push_batch2d=fn("scene_push_batch2d", i32, vp, vp, i32, i32),
        assets_new=fn("assets_new", vp),
        assets_free=fn("assets_free", None, vp),
        assets_add_tile=fn("assets_add_tile", None, vp, ppb, pu, pu, u32),
        rasterizer_setup=fn("rasterizer_setup", vp, pf, pf, pf),
        rasterizer_free=fn("rasterizer_free", None, vp),
        rasterizer_render_mode=fn("rasterizer_render_mode", None, vp, i32, i32, i32),
        rasterizer_sample_mode=fn("rasterizer_sample_mode", None, vp, i32),
        rasterizer_background=fn("rasterizer_background", None, vp, pb),
        rasterizer_brush_preview=fn("rasterizer_brush_preview", None, vp, i32, f32, f32, f32, f32, f32),
        rasterizer_ambient=fn("rasterizer_ambient", None, vp, pf),
        rasterizer_time=fn("rasterizer_time", None, vp, f32),
        rasterizer_preserve_transparency=fn("rasterizer_preserve_transparency", None, vp, i32),
        rasterizer_sun=fn("rasterizer_sun", None, vp, pf, f32),
        rasterizer_mapmini_add_occluder=fn("rasterizer_mapmini_add_occluder", None, vp, f32, f32, f32, f32, f32),
        rasterizer_mapmini_add_linedef=fn("rasterizer_mapmini_add_linedef", None, vp, f32, f32, f32, f32),
        rasterizer_get_derived=fn("rasterizer_get_derived", None, vp, pf, pf, pf),
        rasterizer_rasterize=fn("rasterizer_rasterize", i32, vp, vp, pb, u32, u32, u32, vp),
        scene_project=fn("scene_project", i32, vp, vp, u32, u32),
        scene_batch3d_counts=fn("scene_batch3d_counts", i32, vp, i32, i32, u32, pu, pu, pu),
        scene_batch3d_copy=fn("scene_batch3d_copy", i32, vp, i32, i32, u32, pf, pf, pf, pu, pf, pf),
        camera_orbit=fn("camera_orbit", None, pf, f32, f32, f32, f32, f32, f32, f32, f32, pf, pf),
        camera_firstp=fn("camera_firstp", None, pf, pf, f32, f32, f32, f32, f32, pf, pf),
    )

    def tile_args(tile: Tile):
        n = len(tile.textures)
        frames = (pb * n)(*[_bp(t.data) for t in tile.textures])
        ws = (C.c_uint32 * n)(*[t.width for t in tile.textures])
        hs = (C.c_uint32 * n)(*[t.height for t in tile.textures])
        return frames, ws, hs, n

    class Batch3D:
        """reference src/batch/batch3d.rs:15-480 (builder surface)."""

        def __init__(self, handle):
            self._h = handle

        def __del__(self):
            if getattr(self, "_h", None):
                L.batch3d_free(self._h)
                self._h = None

        @staticmethod
        def new(vertices, indices, uvs):
            v = _f32(vertices, (-1, 4))
            i = _u32(indices, (-1, 3))
            uv = _f32(uvs, (-1, 2))
            assert uv.shape[0] == v.shape[0]
            return Batch3D(L.batch3d_new(_fp(v), v.shape[0], _up(i), i.shape[0], _fp(uv)))

        @staticmethod
        def from_box(x, y, z, w, h, d):
            return Batch3D(L.batch3d_from_box(x, y, z, w, h, d))

        @staticmethod
        def from_obj(text: str):
            return Batch3D(L.batch3d_from_obj(text.encode("utf-8")))

        def add(self, vertices, indices, uvs):
            v = _f32(vertices, (-1, 4))
            i = _u32(indices, (-1, 3))
            uv = _f32(uvs, (-1, 2))
            L.batch3d_add(self._h, _fp(v), v.shape[0], _up(i), i.shape[0], _fp(uv))
            return self

        def normals(self, n):
            n = _f32(n, (-1, 3))
            L.batch3d_set_normals(self._h, _fp(n), n.shape[0])
            return self

        def with_computed_normals(self):
            L.batch3d_compute_vertex_normals(self._h)
            return self

        compute_vertex_normals = with_computed_normals

        def source(self, src: PixelSource):
            if src.kind in (HOST_SOURCE_ENTITY_TILE, HOST_SOURCE_ITEM_TILE):
                L.batch3d_set_source_seq(self._h, 1 if src.kind == HOST_SOURCE_ITEM_TILE else 0, src.index, src.seq)
                return self
            px = (C.c_uint8 * 4)(*src.pixel)
            L.batch3d_set_source(self._h, src.kind, src.index, px)
            return self

        def repeat_mode(self, m):
            L.batch3d_set_repeat_mode(self._h, m)
            return self

        def cull_mode(self, m):
            L.batch3d_set_cull_mode(self._h, m)
            return self

        def ambient_color(self, c):
            L.batch3d_set_ambient_color(self._h, c[0], c[1], c[2])
            return self

        def transform(self, m16):
            m = _f32(m16, (16,))
            L.batch3d_set_transform(self._h, _fp(m))
            return self

        def profile_id(self, pid):
            L.batch3d_set_profile_id(self._h, 1, pid)
            return self

        def shader(self, idx):
            L.batch3d_set_shader(self._h, idx)
            return self

        def counts(self):
            nv, nt = C.c_uint32(), C.c_uint32()
            L.batch3d_counts(self._h, C.byref(nv), C.byref(nt))
            return nv.value, nt.value

        def geometry(self):
            nv, nt = self.counts()
            v = np.zeros((nv, 4), np.float32)
            i = np.zeros((nt, 3), np.uint32)
            uv = np.zeros((nv, 2), np.float32)
            nn = L.batch3d_num_normals(self._h)
            n = np.zeros((nn, 3), np.float32)
            L.batch3d_get_geometry(self._h, _fp(v), _up(i), _fp(uv), _fp(n) if nn else None)
            return v, i, uv, n

    class Batch2D:
        """reference src/batch/batch2d.rs:10-371 (builder surface)."""

        def __init__(self, handle):
            self._h = handle

        def __del__(self):
            if getattr(self, "_h", None):
                L.batch2d_free(self._h)
                self._h = None

        @staticmethod
        def new(vertices, indices, uvs):
            v = _f32(vertices, (-1, 2))
            i = _u32(indices, (-1, 3))
            uv = _f32(uvs, (-1, 2))
            return Batch2D(L.batch2d_new(_fp(v), v.shape[0], _up(i), i.shape[0], _fp(uv)))

        @staticmethod
        def from_rectangle(x, y, w, h):
            return Batch2D(L.batch2d_from_rectangle(x, y, w, h))

        def mode(self, m):
            L.batch2d_set_mode(self._h, m)
            return self

        def repeat_mode(self, m):
            L.batch2d_set_repeat_mode(self._h, m)
            return self

        def source(self, src: PixelSource):
            if src.kind in (HOST_SOURCE_ENTITY_TILE, HOST_SOURCE_ITEM_TILE):
                L.batch2d_set_source_seq(self._h, 1 if src.kind == HOST_SOURCE_ITEM_TILE else 0, src.index, src.seq)
                return self
            px = (C.c_uint8 * 4)(*src.pixel)
            L.batch2d_set_source(self._h, src.kind, src.index, px)
            return self

        def receives_light(self, v):
            L.batch2d_set_receives_light(self._h, 1 if v else 0)
            return self

        def shader(self, idx):
            L.batch2d_set_shader(self._h, idx)
            return self

    class Chunk:
        """reference src/chunk.rs (fields the raster loops read)."""

        def __init__(self, scene, index):
            self._scene, self.index = scene, index

        def add_batch3d(self, b):
            assert L.scene_push_batch3d(self._scene._h, b._h, LIST_CHUNK, self.index) == 0
            return self

        def add_batch3d_opacity(self, b):
            assert L.scene_push_batch3d(self._scene._h, b._h, LIST_CHUNK_OPACITY, self.index) == 0
            return self

        def add_batch2d(self, b):
            assert L.scene_push_batch2d(self._scene._h, b._h, 0, self.index) == 0
            return self

        def terrain(self, texture, origin=(0, 0), size=1):
            """chunk.terrain_texture (a Texture or None), chunk.origin, chunk.size (reference src/chunk.rs:25-36)"""
            if texture is None:
                L.chunk_set_terrain(self._scene._h, self.index, None, 0, 0, origin[0], origin[1], size)
            else:
                L.chunk_set_terrain(self._scene._h, self.index, _bp(texture.data), texture.width, texture.height, origin[0], origin[1], size)
            return self

        def terrain_batch3d(self, b):
            assert L.scene_push_batch3d(self._scene._h, b._h, LIST_CHUNK_TERRAIN, self.index) == 0
            return self

        def terrain_batch2d(self, b):
            L.chunk_set_terrain_batch2d(self._scene._h, self.index, b._h)
            return self

        def add_shader(self, program: Program, baked_texture=None):
            """chunk.add_shader (reference src/chunk.rs:84-131) without the compiler and the 64x64 bake: the program and
            the texture the reference would have baked from it (or None) are given directly; returns the shader index"""
            idx = self._scene.add_program(program, chunk=self.index)
            if baked_texture is None:
                L.chunk_add_shader_texture(self._scene._h, self.index, None, 0, 0)
            else:
                L.chunk_add_shader_texture(self._scene._h, self.index, _bp(baked_texture.data), baked_texture.width, baked_texture.height)
            return idx

        def add_occluder(self, mn, mx, occlusion):
            L.chunk_add_occluder(self._scene._h, self.index, mn[0], mn[1], mx[0], mx[1], occlusion)
            return self

        def add_light(self, light: RxrLight):
            L.chunk_add_light(self._scene._h, self.index, C.byref(light))
            return self

    class Scene:
        """reference src/scene.rs:8-150."""

        def __init__(self):
            self._h = L.scene_new()
            self._keep = []

        def __del__(self):
            if getattr(self, "_h", None):
                L.scene_free(self._h)
                self._h = None

        @staticmethod
        def empty():
            return Scene()

        @staticmethod
        def from_static(d2, d3):
            s = Scene()
            for b in d2:
                s.add_d2_static(b)
            for b in d3:
                s.add_d3_static(b)
            return s

        def background(self, shader):
            L.scene_set_background(self._h, shader.kind if shader is not None else BG_NONE)
            if shader is not None and shader.kind == BG_GRID:
                L.scene_set_background_grid(self._h, shader.grid_size, shader.subdivisions, shader.offset[0], shader.offset[1])
            return self

        def lights(self, lights):
            for l in lights:
                L.scene_add_light(self._h, C.byref(l), 0)
            return self

        def add_dynamic_light(self, l):
            L.scene_add_light(self._h, C.byref(l), 1)
            return self

        def set_animation_frame(self, f):
            L.scene_set_animation_frame(self._h, f)
            return self

        def add_d3_static(self, b):
            assert L.scene_push_batch3d(self._h, b._h, LIST_STATIC, -1) == 0
            return self

        def add_d3_dynamic(self, b):
            assert L.scene_push_batch3d(self._h, b._h, LIST_DYNAMIC, -1) == 0
            return self

        def add_d3_overlay(self, b):
            assert L.scene_push_batch3d(self._h, b._h, LIST_OVERLAY, -1) == 0
            return self

        def add_d2_static(self, b):
            assert L.scene_push_batch2d(self._h, b._h, 0, -1) == 0
            return self

        def add_d2_dynamic(self, b):
            assert L.scene_push_batch2d(self._h, b._h, 1, -1) == 0
            return self

        def add_dynamic_texture(self, tile: Tile):
            frames, ws, hs, n = tile_args(tile)
            L.scene_add_dynamic_tile(self._h, frames, ws, hs, n)
            return self

        def add_chunk(self):
            return Chunk(self, L.scene_add_chunk(self._h))

        def add_program(self, program: Program, chunk=-1):
            """scene.add_shader (reference src/scene.rs:104-134) without the compiler; returns the shader index"""
            n = len(program.functions)
            arrs = [np.asarray(f, np.uint32) if len(f) else np.zeros(1, np.uint32) for f in program.functions]
            ptrs = (pu * max(n, 1))(*[_up(a) for a in arrs])
            lens = (C.c_uint32 * max(n, 1))(*[len(f) for f in program.functions])
            idx = L.scene_add_program(self._h, chunk, program.globals, program.shade_index, program.shade_locals, ptrs, lens, n)
            if idx < 0:
                raise RasterizeError(f"add_program failed ({idx})")
            return idx

        def num_dynamic_lights(self):
            return L.scene_num_dynamic_lights(self._h)

        def projected_batch3d(self, list_kind, index, chunk=-1):
            """Outputs of clip_and_project for one batch (after rasterize): dict of numpy arrays."""
            nv, nt, hn = C.c_uint32(), C.c_uint32(), C.c_uint32()
            rc = L.scene_batch3d_counts(self._h, list_kind, chunk, index, C.byref(nv), C.byref(nt), C.byref(hn))
            if rc != 0:
                raise IndexError("no such batch")
            pv = np.zeros((nv.value, 4), np.float32)
            uv = np.zeros((nv.value, 2), np.float32)
            nr = np.zeros((nv.value, 3), np.float32)
            idx = np.zeros((nt.value, 3), np.uint32)
            ed = np.zeros((nt.value, 10), np.float32)
            bb = np.zeros(5, np.float32)
            L.scene_batch3d_copy(self._h, list_kind, chunk, index, _fp(pv), _fp(uv), _fp(nr), _up(idx), _fp(ed), _fp(bb))
            return dict(projected_vertices=pv, clipped_uvs=uv, clipped_normals=nr, clipped_indices=idx, edges=ed,
                        bounding_box=bb, has_normals=bool(hn.value))

    class Assets:
        """reference src/server/assets.rs (`tile_list`, `.textures(..)` builder)."""

        def __init__(self):
            self._h = L.assets_new()

        def __del__(self):
            if getattr(self, "_h", None):
                L.assets_free(self._h)
                self._h = None

        @staticmethod
        def default():
            return Assets()

        def textures(self, tiles):
            for t in tiles:
                frames, ws, hs, n = tile_args(t)
                L.assets_add_tile(self._h, frames, ws, hs, n)
            return self

        def entity_tiles(self, tiles_by_id, item=False):
            """assets.entity_tiles (or, with item=True, assets.item_tiles): {id: [Tile, ...]} -- the sequences of an id in
            IndexMap insertion order (reference src/server/assets.rs:28, :34); an id with an empty list is known but has no sequence"""
            for ident, tiles in tiles_by_id.items():
                L.assets_add_sequence_id(self._h, 1 if item else 0, int(ident))
                for t in tiles:
                    frames, ws, hs, n = tile_args(t)
                    L.assets_add_sequence_tile(self._h, 1 if item else 0, int(ident), frames, ws, hs, n)
            return self

        def item_tiles(self, tiles_by_id):
            return self.entity_tiles(tiles_by_id, item=True)

        def patterns(self, textures, normal=False):
            """rusteria's global pattern bank (rusteria/src/textures/patterns.rs) as data: a list of float32
            arrays of shape (h, w, 3); `normal=True` sets the bank NodeOp::SampleNormal reads"""
            arrs = [_f32(t, (-1,)) for t in textures]
            shapes = [np.asarray(t).shape for t in textures]
            n = len(arrs)
            ptrs = (pf * max(n, 1))(*[_fp(a) for a in arrs])
            ws = (C.c_uint32 * max(n, 1))(*[sh[1] for sh in shapes])
            hs = (C.c_uint32 * max(n, 1))(*[sh[0] for sh in shapes])
            L.assets_set_patterns(self._h, 1 if normal else 0, ptrs, ws, hs, n)
            return self

        def palette(self, colors):
            """assets.palette.colors: a list of (r, g, b) floats or None (empty slot)"""
            rgb = np.zeros((max(len(colors), 1), 3), np.float32)
            present = np.zeros(max(len(colors), 1), np.uint8)
            for i, c in enumerate(colors):
                if c is not None:
                    rgb[i] = c
                    present[i] = 1
            L.assets_set_palette(self._h, _fp(rgb), _bp(present), len(colors))
            return self

    class Rasterizer:
        """reference src/rasterizer.rs:35-193."""

        def __init__(self, handle):
            self._h = handle

        def __del__(self):
            if getattr(self, "_h", None):
                L.rasterizer_free(self._h)
                self._h = None

        @staticmethod
        def setup(projection_matrix_2d, view_matrix, projection_matrix):
            m2d = _f32(projection_matrix_2d, (9,)) if projection_matrix_2d is not None else None
            v = _f32(view_matrix, (16,))
            p = _f32(projection_matrix, (16,))
            return Rasterizer(L.rasterizer_setup(_fp(m2d) if m2d is not None else None, _fp(v), _fp(p)))

        def render_mode(self, rm: RenderMode):
            L.rasterizer_render_mode(self._h, int(rm.d2_active), int(rm.d3_active), int(rm.ignore_background_shader_flag))
            return self

        def sample_mode(self, m):
            L.rasterizer_sample_mode(self._h, m)
            return self

        def background(self, pixel):
            px = (C.c_uint8 * 4)(*pixel)
            L.rasterizer_background(self._h, px)
            return self

        def brush_preview(self, position, radius, falloff):
            """`rasterizer.brush_preview = Some(BrushPreview { position, radius, falloff })` (a public field in the reference,
            src/rasterizer.rs:13-17, :65); position None = `None`"""
            if position is None:
                L.rasterizer_brush_preview(self._h, 0, 0.0, 0.0, 0.0, 0.0, 0.0)
            else:
                L.rasterizer_brush_preview(self._h, 1, float(position[0]), float(position[1]), float(position[2]), float(radius), float(falloff))
            return self

        def ambient(self, v4):
            a = _f32(v4, (4,))
            L.rasterizer_ambient(self._h, _fp(a))
            return self

        def time(self, t):
            L.rasterizer_time(self._h, t)
            return self

        def preserve_transparency(self, v):
            L.rasterizer_preserve_transparency(self._h, 1 if v else 0)
            return self

        def sun(self, direction, day_factor):
            d = _f32(direction, (3,))
            L.rasterizer_sun(self._h, _fp(d), day_factor)
            return self

        def mapmini_add_occluder(self, mn, mx, occlusion):
            L.rasterizer_mapmini_add_occluder(self._h, mn[0], mn[1], mx[0], mx[1], occlusion)
            return self

        def mapmini_add_linedef(self, start, end):
            L.rasterizer_mapmini_add_linedef(self._h, start[0], start[1], end[0], end[1])
            return self

        def derived(self):
            iv, ip, cp = np.zeros(16, np.float32), np.zeros(16, np.float32), np.zeros(3, np.float32)
            L.rasterizer_get_derived(self._h, _fp(iv), _fp(ip), _fp(cp))
            return iv, ip, cp

        def project(self, scene, width, height):
            """Host-side `scene.project(..)` only (reference src/scene.rs:154-200)."""
            rc = L.scene_project(self._h, scene._h, width, height)
            if rc != 0:
                raise RasterizeError(rc, "projection failed (batch without normals panics in the reference)")
            return self

        def rasterize(self, scene, pixels, width, height, tile_size, assets):
            assert pixels.dtype == np.uint8 and pixels.size == width * height * 4 and pixels.flags["C_CONTIGUOUS"]
            rc = L.rasterizer_rasterize(self._h, scene._h, _bp(pixels), width, height, tile_size, assets._h)
            if rc != 0:
                msg = ""
                if hasattr(lib, prefix + "last_error"):
                    f = getattr(lib, prefix + "last_error")
                    f.restype = C.c_char_p
                    msg = (f() or b"").decode()
                raise RasterizeError(rc, msg)
            return self

    class D3OrbitCamera:
        """reference src/camera/d3orbit.rs:23-56,186-195."""

        def __init__(self):
            self.center = (0.0, 0.0, 0.0)
            self.distance = 20.0
            self.azimuth = float(np.float32(np.pi) / np.float32(2.0))
            self.elevation = 0.698
            self.fov, self.near, self.far = 75.0, 0.01, 100.0

        @staticmethod
        def new():
            return D3OrbitCamera()

        def set_parameter_f32(self, key, value):
            if key == "distance":
                self.distance = value

        def matrices(self, width, height):
            v, p = np.zeros(16, np.float32), np.zeros(16, np.float32)
            c = _f32(self.center, (3,))
            L.camera_orbit(_fp(c), self.distance, self.azimuth, self.elevation, self.fov, self.near, self.far, width,
                           height, _fp(v), _fp(p))
            return v, p

        def view_matrix(self):
            return self.matrices(1.0, 1.0)[0]

        def projection_matrix(self, width, height):
            return self.matrices(width, height)[1]

    class D3FirstPCamera:
        """reference src/camera/d3firstp.rs:17-42."""

        def __init__(self):
            self.position = (0.0, 0.0, 0.0)
            self.center = (0.0, 0.0, 0.0)
            self.fov, self.near, self.far = 75.0, 0.01, 100.0

        @staticmethod
        def new():
            return D3FirstPCamera()

        def matrices(self, width, height):
            v, p = np.zeros(16, np.float32), np.zeros(16, np.float32)
            pos = _f32(self.position, (3,))
            c = _f32(self.center, (3,))
            L.camera_firstp(_fp(pos), _fp(c), self.fov, self.near, self.far, width, height, _fp(v), _fp(p))
            return v, p

        def view_matrix(self):
            return self.matrices(1.0, 1.0)[0]

        def projection_matrix(self, width, height):
            return self.matrices(width, height)[1]

    return types.SimpleNamespace(
        name=name, lib=lib, prefix=prefix, raw=L,
        Scene=Scene, Batch3D=Batch3D, Batch2D=Batch2D, Chunk=Chunk, Assets=Assets, Rasterizer=Rasterizer,
        D3OrbitCamera=D3OrbitCamera, D3FirstPCamera=D3FirstPCamera,
        # shared value types
        Texture=Texture, Tile=Tile, Light=Light, PixelSource=PixelSource, RenderMode=RenderMode,
        VGrayGradientShader=VGrayGradientShader, GridShader=GridShader, Mat4=Mat4, Mat3=Mat3, Program=Program,
    )
