"""Rusteria programs: the device VM (flattened jump code, rusterix_amd/csrc/rxr_vm.h)
against the oracle's tree interpreter (oracle/rusteria_vm.hpp) on the same scenes, through the C ABI.

Programs that only use exact arithmetic must match bit for bit; sin / cos / tan / atan / atan2 / pow / ln come
from different math libraries (OCML vs glibc) and are compared at +-1 per 8-bit channel (TOLERANCE).  All
programs here are "pure" in the sense of include/rxr.h, so the reference's per-tile Execution gives the same
result as a per-fragment one."""
import numpy as np
import pytest

from rusterix_amd import binding as B
from rusterix_amd import scenes
from rusterix_amd.binding import Program

pytestmark = pytest.mark.gpu

TOLERANCE = 1
W, H = 192, 128


def patterns():
    rng = np.random.default_rng([0x52585231, 77])
    return [rng.random((16, 32, 3), dtype=np.float32), rng.random((8, 8, 3), dtype=np.float32)]


def rect_scene(api, program, textured=True, time=0.0, lights=False):
    """a 2D rectangle over the whole frame whose batch runs `program` (src/rasterizer.rs:760-797)"""
    scene = api.Scene.empty()
    idx = scene.add_program(program)
    rect = api.Batch2D.from_rectangle(0.0, 0.0, float(W), float(H))
    rect.source(B.PixelSource.StaticTileIndex(0) if textured else B.PixelSource.Pixel((200, 100, 50, 255)))
    rect.shader(idx)
    scene.add_d2_static(rect)
    if lights:
        scene.lights([B.Light(B.LIGHT_POINT).with_position((60.0, 0.0, 40.0)).with_color((1.0, 0.9, 0.8)).with_intensity(1.5)
                      .with_start_distance(10.0).with_end_distance(120.0).compile()])
    assets = api.Assets.default().textures([B.Tile.from_texture(scenes.noise_texture(5, 32, 32))])
    assets.patterns(patterns()).patterns(patterns()[::-1], normal=True).palette([(0.9, 0.1, 0.2), None, (0.2, 0.3, 0.9)])

    def setup():
        return api.Rasterizer.setup(None, B.Mat4.identity(), B.Mat4.identity()).time(time)

    return scenes._result(api, scene, assets, setup, W, H, 40, "shader-rect")


def cube_scene(api, program, opacity_list=False, behind=True):
    """textured cube with lights whose batch runs `program` (src/rasterizer.rs:1283-1304 / :1642-1667)"""
    scene = api.Scene.empty()
    idx = scene.add_program(program)
    box = api.Batch3D.from_box(-0.5, -0.5, -0.5, 1.0, 1.0, 1.0).cull_mode(B.CULL_OFF).with_computed_normals()
    box.source(B.PixelSource.StaticTileIndex(0)).repeat_mode(B.REPEAT_REPEAT_XY).shader(idx)
    if behind:
        back = api.Batch3D.from_box(-1.2, -1.2, -2.0, 2.4, 2.4, 0.2).with_computed_normals().source(B.PixelSource.Pixel((30, 160, 60, 255)))
        scene.add_d3_static(back)
    if opacity_list:
        chunk = scene.add_chunk()
        chunk.add_batch3d_opacity(box)
    else:
        scene.add_d3_static(box)
    scene.lights([B.Light(B.LIGHT_POINT).with_position((1.5, 1.0, 2.0)).with_color((1.0, 0.95, 0.8)).with_intensity(2.0)
                  .with_start_distance(1.0).with_end_distance(8.0).compile()])
    assets = api.Assets.default().textures([B.Tile.from_texture(scenes.noise_texture(6, 32, 32))])
    assets.patterns(patterns())
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 2.5)
    cam.azimuth = 1.0
    cam.elevation = 0.5

    def setup():
        v, p = cam.matrices(float(W), float(H))
        return api.Rasterizer.setup(None, v, p).ambient((0.3, 0.3, 0.35, 1.0)).time(1.25)

    return scenes._result(api, scene, assets, setup, W, H, 40, "shader-cube")


def compare(oracle, product, build, tol=0):
    got = scenes.render(build(product))
    ref = scenes.render(build(oracle))
    diff = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    assert int(diff.max()) <= tol, f"{(diff > tol).sum()} pixels differ by more than {tol} (max {diff.max()}); first at {np.argwhere(diff > tol)[:3].tolist()}"
    return got


# per-pixel varying operands: A = (u*4 - 2, v*4 - 2, u*v) and B = (v + 0.25, u - 0.5, 1.5 - u)
A = ["UV", ("GetComponents", [0]), ("Push", 16.0), "Mul", ("Push", 2.0), "Sub",
     "UV", ("GetComponents", [1]), ("Push", 16.0), "Mul", ("Push", 2.0), "Sub",
     "UV", ("GetComponents", [0]), "UV", ("GetComponents", [1]), "Mul", ("Push", 16.0), "Mul", "Pack3"]
Bv = ["UV", ("GetComponents", [1]), ("Push", 4.0), "Mul", ("Push", 0.25), "Add",
      "UV", ("GetComponents", [0]), ("Push", 4.0), "Mul", ("Push", 0.5), "Sub",
      ("Push", 1.5), "UV", ("GetComponents", [0]), ("Push", 4.0), "Mul", "Sub", "Pack3"]
TO_COLOR = [("Push", 0.25), "Mul", ("Push", 0.5), "Add", "SetColor"]

EXACT_UNARY = ["Abs", "Neg", "Floor", "Ceil", "Round", "Fract", "Length", "Length2", "Length3", "Normalize", "Sqrt", "Radians", "Degrees", "Not"]
EXACT_BINARY = ["Add", "Sub", "Mul", "Div", "Min", "Max", "Mod", "Step", "Dot", "Dot2", "Dot3", "Cross", "Eq", "Ne", "Lt", "Le", "Gt", "Ge", "And", "Or"]
LIBM_UNARY = ["Sin", "Sin1", "Sin2", "Cos", "Cos1", "Cos2", "Tan", "Atan", "Log"]
LIBM_BINARY = ["Atan2", "Pow", "Rotate2D"]


@pytest.mark.parametrize("op", EXACT_UNARY)
def test_exact_unary_ops(oracle, product, op):
    compare(oracle, product, lambda api: rect_scene(api, Program([A + [op] + TO_COLOR])))


@pytest.mark.parametrize("op", EXACT_BINARY)
def test_exact_binary_ops(oracle, product, op):
    compare(oracle, product, lambda api: rect_scene(api, Program([A + Bv + [op] + TO_COLOR])))


@pytest.mark.parametrize("op", LIBM_UNARY)
def test_libm_unary_ops(oracle, product, op):
    compare(oracle, product, lambda api: rect_scene(api, Program([A + [op] + TO_COLOR])), tol=TOLERANCE)


@pytest.mark.parametrize("op", LIBM_BINARY)
def test_libm_binary_ops(oracle, product, op):
    args = A + [("Push", 37.0)] if op == "Rotate2D" else A + Bv
    compare(oracle, product, lambda api: rect_scene(api, Program([args + [op] + TO_COLOR])), tol=TOLERANCE)


def test_ternary_ops_swizzles_and_fields(oracle, product):
    prog = Program([A + Bv + [("Push", 0.3, 0.6, 0.9), "Mix",                      # mix(A, B, t)
                              ("Push", -1.0), ("Push", 1.0), "Clamp",
                              ("Push", 0.0), ("Push", 1.0), "UV", ("GetComponents", [0]), ("Push", 4.0), "Mul", "Smoothstep", "Mul",
                              "Color", "Add",                                      # + the texel (pixel_to_vec4, not linearised in 2D)
                              "Hitpoint", ("Push", 0.001), "Mul", "Add",
                              "Time", ("Push", 0.1), "Mul", "Add",
                              "Dup", ("GetComponents", [2, 0]), ("SetComponents", [1, 2]),
                              ("Push", 0.5), "Mul", "SetColor"]])
    compare(oracle, product, lambda api: rect_scene(api, prog, time=0.75, lights=True))


def test_patterns_and_palette(oracle, product):
    prog = Program([["UV", ("Push", 8.0), "Mul", ("Push", 0.0), "Sample",
                     "UV", ("Push", 5.0), "Mul", ("Push", 1.0), "SampleNormal", ("Push", 0.25), "Mul", "Add",
                     # slot 1 is empty: PaletteIndex then pushes nothing and the Add below consumes the sample instead
                     ("Push", 0.1, 0.1, 0.1), "UV", ("GetComponents", [0]), ("Push", 12.0), "Mul", "PaletteIndex", "Add",
                     "UV", ("Push", 3.0), "Mul", ("Push", 9.0), "Sample", "Add",                                          # no such pattern
                     "SetColor"]])
    got = compare(oracle, product, lambda api: rect_scene(api, prog))
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 50


def test_control_flow(oracle, product):
    # per-pixel trip count, nested If, a helper function with a Return inside an If, recursion
    helper = [("LoadLocal", 0), ("Push", 0.5), "Gt", ("If", [("LoadLocal", 0), ("Push", 0.5), "Sub", "Return"], None), ("LoadLocal", 0), "Return"]
    fact = [("LoadLocal", 0), ("Push", 1.0), "Le", ("If", [("Push", 1.0), "Return"], None),
            ("LoadLocal", 0), ("LoadLocal", 0), ("Push", 1.0), "Sub", ("FunctionCall", 1, 1, 2), "Mul", "Return"]
    shade = [("Push", 0.0), ("StoreLocal", 0),
             "UV", ("GetComponents", [0]), ("Push", 40.0), "Mul", "Floor", ("StoreLocal", 2),          # 0..9 iterations
             ("For", [("Push", 0.0), ("StoreLocal", 1)], [("LoadLocal", 1), ("LoadLocal", 2), "Lt"],
              [("LoadLocal", 1), ("Push", 1.0), "Add", ("StoreLocal", 1)],
              [("LoadLocal", 1), ("Push", 2.0), "Mod", ("Push", 0.0), "Eq",
               ("If", [("LoadLocal", 0), ("Push", 0.07), "Add", ("StoreLocal", 0)], [("LoadLocal", 0), ("Push", 0.02), "Add", ("StoreLocal", 0)]),
               ("Push", 5.0)]),                                                                          # a temporary the loop truncates
             ("LoadLocal", 0), "UV", ("GetComponents", [1]), ("Push", 4.0), "Mul", ("FunctionCall", 1, 1, 1),
             "UV", ("GetComponents", [0]), ("Push", 16.0), "Mul", "Floor", ("FunctionCall", 1, 1, 2), ("Push", 0.04), "Mul", "Pack3", "SetColor"]
    got = compare(oracle, product, lambda api: rect_scene(api, Program([shade, helper, fact], shade_locals=3), textured=False))
    assert len(np.unique(got[..., 0])) > 5 and len(np.unique(got[..., 2])) >= 3


def test_globals_written_before_read(oracle, product):
    prog = Program([["UV", ("StoreGlobal", 1), ("LoadGlobal", 1), ("Push", 4.0), "Mul", ("FunctionCall", 0, 0, 1), "Add", "SetColor"],
                    [("LoadGlobal", 1), ("Push", 1.0), "Mul"]], globals=2)
    compare(oracle, product, lambda api: rect_scene(api, prog))


def test_cube_material_program(oracle, product):
    """the 3D opaque pass: colour, roughness, metallic and normal come back from the program and feed the lighting"""
    prog = Program([["Color", "UV", ("Push", 6.0), "Mul", ("Push", 0.0), "Sample", "Mul", ("Push", 1.5), "Mul", "SetColor",
                     "UV", ("GetComponents", [0]), ("Push", 4.0), "Mul", "Fract", "SetRoughness",
                     "UV", ("GetComponents", [1]), ("Push", 4.0), "Mul", "Fract", "SetMetallic",
                     "Normal", "Hitpoint", ("Push", 0.15), "Mul", "Add", "SetNormal"]])
    got = compare(oracle, product, lambda api: cube_scene(api, prog), tol=TOLERANCE)
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 200


def test_cube_program_without_shade_function_is_ignored(oracle, product):
    compare(oracle, product, lambda api: cube_scene(api, Program([[("Push", 1.0), "SetColor"]], shade_index=None)), tol=TOLERANCE)


def test_program_opacity_decides_visibility(oracle, product):
    """a program that writes opacity: fragments whose encoded alpha is not 255 are not written (src/rasterizer.rs:1403-1412),
    so the wall behind shows through the holes and the depth of the holes is the wall's"""
    prog = Program([["UV", ("Push", 12.0), "Mul", "Fract", ("GetComponents", [0]), ("Push", 0.5), "Gt",
                     ("If", [("Push", 1.0), "SetOpacity"], [("Push", 0.4), "SetOpacity"]),
                     "Color", ("Push", 1.2), "Mul", "SetColor"]])
    got = compare(oracle, product, lambda api: cube_scene(api, prog), tol=TOLERANCE)
    green = (got[..., 1] > got[..., 0] + 40) & (got[..., 1] > got[..., 2] + 40)
    centre = green[H // 2 - 20:H // 2 + 20, W // 2 - 20:W // 2 + 20]
    assert 0.2 < centre.mean() < 0.8, "the cut-outs should expose the wall behind the cube"


def test_opacity_pass_program(oracle, product):
    """chunk.batches3d_opacity with a program (src/rasterizer.rs:1642-1673): colour and opacity come from the program"""
    prog = Program([["Color", "Hitpoint", ("Push", 0.5), "Mul", "Abs", "Add", "SetColor",
                     "UV", ("GetComponents", [1]), ("Push", 4.0), "Mul", "Fract", "SetOpacity", "Normal", "Length", ("Push", 0.0), "Eq",
                     ("If", [], [("Push", 0.0), "SetColor"])]])    # the opacity pass hands the program a zero normal
    got = compare(oracle, product, lambda api: cube_scene(api, prog, opacity_list=True), tol=TOLERANCE)
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 100


def test_tile_size_does_not_matter_for_pure_programs(oracle):
    prog = Program([["Color", "UV", "Add", "SetColor", "UV", ("GetComponents", [0]), ("Push", 4.0), "Mul", "SetRoughness"]])
    cfg = cube_scene(oracle, prog)
    a = scenes.render(cfg).copy()
    cfg.tile_size = 16
    assert np.array_equal(a, scenes.render(cfg))


# ---- what the device refuses or reports --------------------------------------------------------------------
@pytest.mark.parametrize("ops, why", [
    ([("LoadGlobal", 0), "SetColor"], "global read before written"),
    ([("LoadLocal", 0), "SetColor"], "stale local of the previous fragment"),
    (["UV", "SetColor", ("Push", 1.0, 2.0, 3.0), "SetUV"], "uv.z of the previous fragment"),
    (["Roughness", "SetColor", ("Push", 0.1, 0.2, 0.3), "SetRoughness"], "roughness.yz of the previous fragment"),
    ([("Push", 1.0), ("If", [("Push", 2.0), ("StoreLocal", 0)], None), ("LoadLocal", 0), "SetColor"], "local written on one path only"),
    ([("For", [], [("Push", 0.0)], [], ["Return"])], "Return inside For"),
    ([("Push", 4.0), ("Push", 4.0), "Alloc"], "texture baking"),
])
def test_unsupported_programs_are_refused(product, ops, why):
    cfg = rect_scene(product, Program([ops], globals=1, shade_locals=1))
    with pytest.raises(B.RasterizeError) as e:
        scenes.render(cfg)
    assert e.value.code == B.RXR_ERR_UNSUPPORTED, why


# ---- emissive (rasterizer.rs:1323, :1394): accepted where no fragment can see another fragment's value -----------------------
EMISSIVE = ["UV", ("GetComponents", [0]), ("Push", 4.0), "Mul", ("Push", 0.3), "Mul", ("Push", 0.05), ("Push", 0.0), "Pack3", "SetEmissive"]


def emissive_scene(api, programs, lights=True, opacity_pane=None):
    """three boxes next to each other, box i running programs[i] (None: no program), over nothing: every 3D batch on screen
    decides the frame's fate.  `opacity_pane`: a program for an opacity-pass pane in front of them."""
    scene = api.Scene.empty()
    for i, prog in enumerate(programs):
        box = api.Batch3D.from_box(-1.7 + 1.15 * i, -0.5, -0.5, 1.0, 1.0, 1.0).cull_mode(B.CULL_OFF).with_computed_normals()
        box.source(B.PixelSource.StaticTileIndex(0)).repeat_mode(B.REPEAT_REPEAT_XY)
        if prog is not None:
            box.shader(scene.add_program(prog))
        scene.add_d3_static(box)
    if opacity_pane is not None:
        chunk = scene.add_chunk()
        pane = api.Batch3D.from_box(-1.0, -0.2, 0.9, 2.0, 0.4, 0.02).with_computed_normals().source(B.PixelSource.Pixel((90, 120, 250, 120)))
        pane.shader(scene.add_program(opacity_pane, chunk=chunk.index))
        chunk.add_batch3d_opacity(pane)
    if lights:
        scene.lights([B.Light(B.LIGHT_POINT).with_position((1.5, 1.0, 2.0)).with_color((1.0, 0.95, 0.8)).with_intensity(2.0)
                      .with_start_distance(1.0).with_end_distance(8.0).compile()])
    assets = api.Assets.default().textures([B.Tile.from_texture(scenes.noise_texture(6, 32, 32))])
    cam = api.D3OrbitCamera.new()
    cam.set_parameter_f32("distance", 3.2)
    cam.azimuth = float(np.float32(np.pi / 2))
    cam.elevation = 0.35

    def setup():
        v, p = cam.matrices(float(W), float(H))
        return api.Rasterizer.setup(None, v, p).ambient((0.3, 0.3, 0.35, 1.0))

    return scenes._result(api, scene, assets, setup, W, H, 40, "emissive-boxes")


GLOW = Program([EMISSIVE])
GLOW_BOTH_BRANCHES = Program([["UV", ("GetComponents", [1]), ("Push", 0.1), "Lt",
                               ("If", [("Push", 0.4, 0.1, 0.0), "SetEmissive"], [("Push", 0.0, 0.1, 0.5), "SetEmissive"]), "Color", ("Push", 0.5), "Mul", "SetColor"]])
NO_GLOW_BUT_ASSIGNS = Program([[("Push", 0.0), "SetEmissive", "Color", "SetColor"]])


@pytest.mark.parametrize("lights", [False, True])
def test_emissive_when_every_opaque_batch_assigns_it(oracle, product, lights):
    got = compare(oracle, product, lambda api: emissive_scene(api, [GLOW, GLOW_BOTH_BRANCHES, NO_GLOW_BUT_ASSIGNS], lights), tol=TOLERANCE if lights else 0)
    plain = scenes.render(emissive_scene(product, [NO_GLOW_BUT_ASSIGNS] * 3, lights))
    assert (got != plain).any(), "the emissive term left no trace"


def test_emissive_frame_does_not_depend_on_the_tile_size(oracle):
    """what makes the accepted case safe: with every opaque fragment assigning emissive itself, the reference's per-tile
    Execution (rasterizer.rs:310) has nothing to leak -- unlike tests/test_oracle_vm.py's leaking scene"""
    cfg = emissive_scene(oracle, [GLOW, GLOW_BOTH_BRANCHES, NO_GLOW_BUT_ASSIGNS])
    a = scenes.render(cfg).copy()
    for ts in (16, 200):
        cfg.tile_size = ts
        assert np.array_equal(a, scenes.render(cfg))


def test_emissive_written_in_the_opacity_pass_is_harmless_when_the_opaque_batches_assign_theirs(oracle, product):
    compare(oracle, product, lambda api: emissive_scene(api, [GLOW, NO_GLOW_BUT_ASSIGNS, GLOW], opacity_pane=GLOW), tol=TOLERANCE)


@pytest.mark.parametrize("programs, pane, why", [
    ([GLOW, None, GLOW], None, "a batch without a program would add its neighbour's emissive"),
    ([GLOW, Program([["Color", "SetColor"]]), GLOW], None, "a program that never assigns emissive"),
    ([GLOW, Program([["UV", ("GetComponents", [0]), ("Push", 0.1), "Lt", ("If", [("Push", 0.5), "SetEmissive"], None)]]), GLOW], None, "assigned on one path only"),
    ([GLOW, Program([["UV", ("GetComponents", [0]), ("Push", 0.1), "Lt", ("If", ["Return"], None), ("Push", 0.5), "SetEmissive"]]), GLOW], None,
     "a Return leaves before the assignment"),
    ([None, None, None], GLOW, "an opacity-pass program leaks into the opaque batches behind it"),
])
def test_emissive_leaks_are_refused(product, programs, pane, why):
    with pytest.raises(B.RasterizeError) as e:
        scenes.render(emissive_scene(product, programs, opacity_pane=pane))
    assert e.value.code == B.RXR_ERR_UNSUPPORTED and "emissive" in str(e.value), why


def test_emissive_in_a_2d_program_is_accepted(oracle, product):
    """the 2D pass runs after every 3D pass of the tile (rasterizer.rs:501) and nothing there reads emissive"""
    compare(oracle, product, lambda api: rect_scene(api, Program([[("Push", 0.5), "SetEmissive", "Color", ("Push", 0.5), "Mul", "SetColor"]])))


@pytest.mark.parametrize("ops, locals_", [
    (["Add"], 0),                                            # pop().unwrap() on an empty stack
    ([("Push", 1.0), ("StoreLocal", 5)], 1),                 # locals[5]
    ([("FunctionCall", 0, 0, 9)], 0),                        # user_functions[9]
    ([("Push", 1.0), ("Push", 2.0), ("Push", 1.0), "Clamp"], 0),
    ([("For", [], [("Push", 1.0)], [], [])], 0),             # endless loop: the reference panics after 10 M iterations
])
def test_program_faults_are_reported(oracle, product, ops, locals_):
    """where the reference panics both sides return an error instead of a frame"""
    for api in (product,):
        with pytest.raises(B.RasterizeError) as e:
            scenes.render(rect_scene(api, Program([ops], shade_locals=locals_), textured=False))
        assert e.value.code == B.RXR_ERR_INVALID
    # and the context is usable afterwards
    compare(oracle, product, lambda api: rect_scene(api, Program([["UV", "SetColor"]])))


# ---- random programs: the tree interpreter (oracle) against the flattened jump code (device) ---------------------------
UNARY = ["Abs", "Neg", "Floor", "Ceil", "Round", "Fract", "Length", "Length2", "Length3", "Normalize", "Sqrt", "Not", "Radians", "Degrees"]
BINARY = ["Add", "Sub", "Mul", "Div", "Min", "Max", "Mod", "Step", "Dot", "Dot2", "Dot3", "Cross", "Eq", "Ne", "Lt", "Le", "Gt", "Ge", "And", "Or"]
SOURCES = ["UV", "UV", "UV", "Hitpoint", "Hitpoint", "Color", "Color", "Time", "Roughness", "Metallic", "Opacity", "Bump"]


class ProgramGen:
    """stack-safe random NodeOp trees: every block leaves the stack exactly `+delta` deeper than it found it"""

    def __init__(self, rng, n_locals, n_functions, setters=None):
        self.rng, self.n_locals, self.n_functions = rng, n_locals, n_functions
        self.first_callable = 0   # functions may only call later ones: no recursion (the reference would overflow its stack)
        self.loadable = n_locals  # locals that are certainly written by now (a stale read would make the program impure)
        # fields this program writes are never read by it (lanes the raster loops do not reset would leak, see rxr_set_shaders)
        writable = ["SetRoughness", "SetMetallic", "SetBump", "SetUV"]
        # (all programs of one scene must agree on it: pass `setters`)
        self.setters = list(setters) if setters is not None else ["SetColor"] + [w for w in writable if rng.random() < 0.5]
        self.sources = [x for x in SOURCES if ("Set" + ("UV" if x == "UV" else x)) not in self.setters]

    def value(self, depth):
        """ops that push exactly one value"""
        r = self.rng
        k = r.integers(0, 12) if depth < 4 else r.integers(1, 4)
        if k == 2 and not self.loadable:
            k = 1
        if k == 11 and not (self.first_callable < self.n_functions and depth < 3):
            k = 3
        if k == 0:
            return [("Push", *[float(x) for x in np.round(r.uniform(-3, 3, 3), 2)])]
        if k == 1:
            return [str(r.choice(self.sources))]
        if k == 2:
            return [("LoadLocal", int(r.integers(0, self.loadable)))]
        if k == 3:
            return [str(self.sources[0]), ("Push", float(r.integers(1, 9))), "Mul"]
        if k in (4, 5):
            return self.value(depth + 1) + [str(r.choice(UNARY))]
        if k in (6, 7, 8):
            return self.value(depth + 1) + self.value(depth + 1) + [str(r.choice(BINARY))]
        if k == 9:
            t = int(r.integers(0, 3))
            if t == 0:
                return self.value(depth + 1) + self.value(depth + 1) + self.value(depth + 1) + ["Mix"]
            if t == 1:
                lo = float(np.round(r.uniform(-2, 0), 2))
                return self.value(depth + 1) + [("Push", lo), ("Push", lo + float(np.round(r.uniform(0, 3), 2))), "Clamp"]
            return self.value(depth + 1) + self.value(depth + 1) + self.value(depth + 1) + ["Smoothstep"]
        if k == 10:
            n = int(r.integers(1, 5))
            sw = [int(x) for x in r.integers(0, 4, n)]   # 3 = "not a component"
            if r.random() < 0.5:
                return self.value(depth + 1) + [("GetComponents", sw)]
            return self.value(depth + 1) + self.value(depth + 1) + [("SetComponents", sw)]
        if k == 11:
            f = int(r.integers(self.first_callable, self.n_functions))
            return self.value(depth + 1) + self.value(depth + 1) + [("FunctionCall", 2, 3 + 4, 1 + f)]
        return self.value(depth + 1) + self.value(depth + 1) + self.value(depth + 1) + ["Pack3"]

    def statement(self, depth):
        """ops that leave the stack as they found it"""
        r = self.rng
        k = r.integers(0, 8) if depth < 3 else 0
        if k <= 2:
            return self.value(depth + 1) + [("StoreLocal", int(r.integers(0, self.n_locals)))]
        if k == 3:
            return self.value(depth + 1) + [str(r.choice(self.setters))]
        if k in (4, 5):
            els = self.block(depth + 1) if r.random() < 0.6 else None
            return self.value(depth + 1) + [("If", self.block(depth + 1), els)]
        if k == 6:
            i = self.n_locals + depth        # a counter local no random store touches (blocks nest at most 3 deep)
            trips = float(r.integers(0, 5))
            return [("For", [("Push", 0.0), ("StoreLocal", i)], [("LoadLocal", i), ("Push", trips), "Lt"],
                     [("LoadLocal", i), ("Push", 1.0), "Add", ("StoreLocal", i)], self.block(depth + 1) + self.value(depth + 1))]  # + a temporary
        return self.value(depth + 1) + ["Clear"]

    def block(self, depth):
        out = []
        for _ in range(int(self.rng.integers(1, 4))):
            out += self.statement(depth)
        return out

    def function(self):
        body = self.block(2)
        tail = self.value(2) + (["Return"] if self.rng.random() < 0.7 else [])
        early = self.value(2) + [("If", self.value(2) + ["Return"], None)] if self.rng.random() < 0.5 else []
        return early + body + tail

    def program(self):
        shade = []
        for i in range(self.n_locals):
            self.loadable = i
            shade += self.value(1) + [("StoreLocal", i)]
        self.loadable = self.n_locals
        # the result: a random value plus a per-pixel term, wrapped into [0, 1) so that the frame shows structure
        shade += self.block(0) + self.block(0) + self.value(0) + self.value(0) + ["Add", str(self.sources[0]), ("Push", 3.7), "Mul", "Add", "Fract", "SetColor"]
        functions = []
        shade_locals = self.n_locals
        self.n_locals = self.loadable = 3       # helper functions: two arguments + one scratch local (a fresh zeroed frame)
        for k in range(self.n_functions):
            self.first_callable = k + 1
            functions.append(self.function())
        self.raw = [shade] + functions   # kept for debugging a failing seed
        return Program([shade] + functions, shade_locals=shade_locals + 4)


@pytest.mark.parametrize("seed", range(40))
def test_random_programs(oracle, product, seed):
    rng = np.random.default_rng([0x52585231, 4242, seed])
    # helper functions use locals 0..2 (two arguments + one scratch), `shade` has n_locals >= 3
    gen = ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=int(rng.integers(0, 3)))
    prog = gen.program()
    compare(oracle, product, lambda api: rect_scene(api, prog, time=0.5))


@pytest.mark.parametrize("seed", range(12))
def test_random_programs_as_cube_materials(oracle, product, seed):
    """the same generator on the 3D opaque pass: colour / roughness / metallic feed the lighting (log2 / exp2: +-1)"""
    rng = np.random.default_rng([0x52585231, 777, seed])
    gen = ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=int(rng.integers(0, 3)))
    prog = gen.program()
    got = scenes.render(cube_scene(product, prog))
    ref = scenes.render(cube_scene(oracle, prog))
    diff = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    # a colour that lands within an ulp of a quantisation or Fract boundary may flip on the +-1-ulp pow: allow a handful
    assert (diff > TOLERANCE).sum() <= 5, f"{(diff > TOLERANCE).sum()} pixels differ by more than {TOLERANCE} (max {diff.max()})"


@pytest.mark.parametrize("seed", [0, 3, 7, 11, 19, 23])
def test_static_and_dynamic_stack_pointer_agree(oracle, product, seed, monkeypatch):
    """Programs whose stack depth is a function of the program counter run with a wave-uniform stack pointer read from the
    code stream (k_raster_vm_s, rxr_api.hip tag_static_depths); RXR_VM_NO_STATIC forces the per-lane bookkeeping.  Both must
    give the oracle's frame -- for the grid's colour program (an If with two pushes) and for random programs."""
    if seed == 0:
        build = lambda api: scenes.box_grid_scene(api, n=24, width=320, height=200, shader=True)   # noqa: E731
    else:
        rng = np.random.default_rng([0x52585231, 4242, seed])
        prog = ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=0).program()
        build = lambda api: rect_scene(api, prog, time=0.5)   # noqa: E731
    ref = scenes.render(build(oracle)).copy()
    monkeypatch.delenv("RXR_VM_NO_STATIC", raising=False)
    static = scenes.render(build(product)).copy()
    monkeypatch.setenv("RXR_VM_NO_STATIC", "1")
    dynamic = scenes.render(build(product)).copy()
    assert np.array_equal(static, ref) and np.array_equal(dynamic, ref)


@pytest.mark.parametrize("scene", ["grid", "cube", "cube-calls", "rows-cutout"])
def test_visibility_with_and_without_interpreter_calls_agree(oracle, product, scene, monkeypatch):
    """Frames whose opaque-pass programs never write `opacity` run k_raster_vm_sv (kernel level 4: the visibility loop of the
    chunk kernel, no call of the interpreter in it; level 5, k_raster_vm_v, when a program has calls and the stack pointer
    is per lane); RXR_VM_VIS_CALLS forces levels 3 / 2.  Both must give the oracle's frame."""
    colour = Program([["Color", "UV", ("Push", 3.0), "Mul", "Fract", "Mul", ("Push", 1.3), "Mul", "SetColor"]])
    if scene == "cube-calls":
        colour = Program([["Color", "UV", ("FunctionCall", 1, 1, 1), "Mul", "SetColor"], [("LoadLocal", 0), ("Push", 3.0), "Mul", "Fract", ("Push", 1.3), "Mul"]])
    if scene == "grid":
        build = lambda api: scenes.box_grid_scene(api, n=24, width=320, height=200, shader=True)   # noqa: E731
        tol = 0
    elif scene in ("cube", "cube-calls"):
        build = lambda api: cube_scene(api, colour)   # noqa: E731
        tol = TOLERANCE
    else:
        from tests import test_gpu_rows as R
        from tests.test_gpu_fuzz import random_texture

        def build(api):   # small-triangle meshes with cut-out textures (the alpha test of the visibility loop), every batch with the program
            rng = np.random.default_rng(99)
            textures = [B.Tile([random_texture(rng, 16, 16, mode)]) for mode in (0, 2, 1)]
            assets = api.Assets.default().textures(textures)
            scene = api.Scene.empty()
            index = scene.add_program(colour)
            for m in range(3):
                v4, idx, uv = R.small_triangles(rng, 400, 0.09, 1.2)
                b = api.Batch3D.new(v4, idx, uv).with_computed_normals().cull_mode(0)
                b.source(B.PixelSource.StaticTileIndex(m)).repeat_mode(B.REPEAT_REPEAT_XY).shader(index)
                scene.add_d3_static(b)
            cam = api.D3OrbitCamera.new()
            cam.set_parameter_f32("distance", 3.0)

            def setup():
                v, p = cam.matrices(219.0, 140.0)
                return api.Rasterizer.setup(None, v, p)

            return scenes._result(api, scene, assets, setup, 219, 140, 40, "rows-cutout-program")
        tol = 0
    ref = scenes.render(build(oracle)).copy()
    monkeypatch.delenv("RXR_VM_VIS_CALLS", raising=False)
    level4 = scenes.render(build(product)).copy()
    monkeypatch.setenv("RXR_VM_VIS_CALLS", "1")
    level3 = scenes.render(build(product)).copy()
    for name, got in (("level 4", level4), ("level 3", level3)):
        diff = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
        assert (diff > tol).sum() <= (5 if tol else 0), f"{scene} {name}: {(diff > tol).sum()} pixels differ (max {diff.max()})"
