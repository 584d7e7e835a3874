"""Known answers for the oracle's restatement of the Rusteria stack VM (oracle/rusteria_vm.hpp), derived by
hand from rusteria/src/node/execution.rs.  The reference ships no tests for the VM; these pin the
restatement to the cited source text, opcode by opcode, including the reference's quirks."""
import math

import numpy as np
import pytest

from rusterix_amd.binding import Program


def f32(x):
    return float(np.float32(x))


def run(oracle, ops, functions=(), shade_locals=0, globals=0, palette=None, patterns=None, normal_patterns=None, **inputs):
    scene = oracle.Scene.empty()
    assets = oracle.Assets.default()
    if palette is not None:
        assets.palette(palette)
    if patterns is not None:
        assets.patterns(patterns)
    if normal_patterns is not None:
        assets.patterns(normal_patterns, normal=True)
    idx = scene.add_program(Program([ops] + list(functions), shade_index=0, shade_locals=shade_locals, globals=globals))
    return oracle.vm_shade(scene, assets, idx, **inputs)


def color_of(oracle, ops, **kw):
    out = run(oracle, list(ops) + ["SetColor"], **kw)
    assert out is not None, "the program faulted"
    return out["color"]


def test_stack_and_swizzle(oracle):
    assert color_of(oracle, [("Push", 1, 2, 3), ("Push", 4, 5, 6), "Swap", "Clear"]) == (4, 5, 6)
    assert color_of(oracle, [("Push", 1, 2, 3), "Dup", "Add"]) == (2, 4, 6)
    assert color_of(oracle, [("Push", 1, 2, 3), ("GetComponents", [2])]) == (3, 3, 3)          # one component: broadcast
    assert color_of(oracle, [("Push", 1, 2, 3), ("GetComponents", [1, 0])]) == (2, 1, 0)       # two: z = 0
    assert color_of(oracle, [("Push", 1, 2, 3), ("GetComponents", [2, 1, 0])]) == (3, 2, 1)
    assert color_of(oracle, [("Push", 1, 2, 3), ("GetComponents", [0, 1, 2, 0])]) == (0, 0, 0)  # four: broadcast(0)
    assert color_of(oracle, [("Push", 1, 2, 3), ("GetComponents", [7, 1])]) == (2, 2, 2)       # index > 2 is skipped
    assert color_of(oracle, [("Push", 1, 2, 3), ("Push", 9, 8, 7), ("SetComponents", [2, 0])]) == (8, 2, 9)
    assert color_of(oracle, [("Push", 1), ("Push", 2), "Pack2"]) == (1, 2, 0)
    assert color_of(oracle, [("Push", 1), ("Push", 2), ("Push", 3), "Pack3"]) == (1, 2, 3)


def test_arithmetic(oracle):
    a, b = ("Push", 1.5, -2.0, 4.0), ("Push", 0.5, 3.0, -8.0)
    assert color_of(oracle, [a, b, "Sub"]) == (1.0, -5.0, 12.0)
    assert color_of(oracle, [a, b, "Mul"]) == (0.75, -6.0, -32.0)
    assert color_of(oracle, [a, b, "Div"]) == (3.0, f32(np.float32(-2.0) / np.float32(3.0)), -0.5)
    assert color_of(oracle, [("Push", 3, 4, 12), "Length"]) == (13, 13, 13)
    assert color_of(oracle, [("Push", 3, 4, 12), "Length2"]) == (5, 0, 0)
    assert color_of(oracle, [("Push", 3, 4, 12), "Length3"]) == (13, 0, 0)
    assert color_of(oracle, [a, "Abs"]) == (1.5, 2.0, 4.0)
    assert color_of(oracle, [a, "Neg"]) == (-1.5, 2.0, -4.0)
    assert color_of(oracle, [("Push", 1.5, -1.5, 2.5), "Floor"]) == (1, -2, 2)
    assert color_of(oracle, [("Push", 1.5, -1.5, 2.5), "Ceil"]) == (2, -1, 3)
    assert color_of(oracle, [("Push", 1.5, -1.5, 2.5), "Round"]) == (2, -2, 3)            # half away from zero
    assert color_of(oracle, [("Push", 1.25, -1.25, 2.0), "Fract"]) == (0.25, 0.75, 0.0)    # x - floor(x)
    assert color_of(oracle, [("Push", 5.5, -5.5, 7.0), ("Push", 2.0, 2.0, -3.0), "Mod"]) == (1.5, 0.5, -2.0)
    assert color_of(oracle, [a, b, "Min"]) == (0.5, -2.0, -8.0)
    assert color_of(oracle, [a, b, "Max"]) == (1.5, 3.0, 4.0)
    assert color_of(oracle, [("Push", 0, 10, 1), ("Push", 10, 20, 3), ("Push", 0.5, 0.25, 2.0), "Mix"]) == (5, 12.5, 5)
    assert color_of(oracle, [("Push", 0.5, 1.0, 2.0), ("Push", 0.5, 0.5, 3.0), "Step"]) == (1, 0, 1)
    assert color_of(oracle, [("Push", 5, -5, 0.5), ("Push", 0), ("Push", 1), "Clamp"]) == (1, 0, 0.5)
    assert color_of(oracle, [("Push", 4, 9, 2.25), "Sqrt"]) == (2, 3, 1.5)
    assert color_of(oracle, [("Push", 1, 2, 3), ("Push", 4, 5, 6), "Dot"]) == (32, 32, 32)
    assert color_of(oracle, [("Push", 1, 2, 3), ("Push", 4, 5, 6), "Dot2"]) == (14, 0, 0)
    assert color_of(oracle, [("Push", 1, 2, 3), ("Push", 4, 5, 6), "Dot3"]) == (32, 0, 0)
    assert color_of(oracle, [("Push", 1, 0, 0), ("Push", 0, 1, 0), "Cross"]) == (0, 0, 1)
    assert color_of(oracle, [("Push", 3, 0, 4), "Normalize"]) == (f32(np.float32(3) / np.float32(5)), 0.0, f32(np.float32(4) / np.float32(5)))
    assert color_of(oracle, [("Push", 0, 0, 0), "Normalize"]) == (0, 0, 0)                 # zero length: unchanged
    assert color_of(oracle, [("Push", 180.0), "Radians"])[0] == pytest.approx(math.pi, rel=1e-6)
    assert color_of(oracle, [("Push", math.pi), "Degrees"])[0] == pytest.approx(180.0, rel=1e-6)
    # smoothstep uses the x lanes only and guards a zero denominator
    assert color_of(oracle, [("Push", 0), ("Push", 2), ("Push", 1), "Smoothstep"]) == (0.5, 0.5, 0.5)
    assert color_of(oracle, [("Push", 1), ("Push", 1), ("Push", 5), "Smoothstep"]) == (0, 0, 0)


def test_transcendentals_and_quirks(oracle):
    x = 0.7
    s, c = f32(math.sin(np.float32(x))), f32(math.cos(np.float32(x)))
    assert color_of(oracle, [("Push", x), "Sin"])[0] == pytest.approx(s, rel=1e-6)
    assert color_of(oracle, [("Push", x), "Cos"])[0] == pytest.approx(c, rel=1e-6)
    assert color_of(oracle, [("Push", x, x, x), "Sin1"]) == pytest.approx((s, 0, 0), rel=1e-6)
    assert color_of(oracle, [("Push", x, x, x), "Sin2"]) == pytest.approx((s, s, 0), rel=1e-6)
    # execution.rs:337-344: Cos1 and Cos2 compute the SINE
    assert color_of(oracle, [("Push", x, x, x), "Cos1"]) == pytest.approx((s, 0, 0), rel=1e-6)
    assert color_of(oracle, [("Push", x, x, x), "Cos2"]) == pytest.approx((s, s, 0), rel=1e-6)
    assert color_of(oracle, [("Push", x), "Tan"])[0] == pytest.approx(math.tan(x), rel=1e-5)
    assert color_of(oracle, [("Push", x), "Atan"])[0] == pytest.approx(math.atan(x), rel=1e-6)
    assert color_of(oracle, [("Push", 1.0), ("Push", -1.0), "Atan2"])[0] == pytest.approx(math.atan2(1.0, -1.0), rel=1e-6)
    assert color_of(oracle, [("Push", 2.0), ("Push", 10.0), "Pow"])[0] == 1024.0
    assert color_of(oracle, [("Push", math.e), "Log"])[0] == pytest.approx(1.0, rel=1e-6)
    r = color_of(oracle, [("Push", 1.0, 0.0, 5.0), ("Push", 90.0), "Rotate2D"])
    assert r == pytest.approx((0.0, 1.0, 5.0), abs=1e-6)


def test_comparisons_and_logic(oracle):
    one, zero = (1, 1, 1), (0, 0, 0)
    assert color_of(oracle, [("Push", 1, 9, 9), ("Push", 1, 0, 0), "Eq"]) == one   # x lanes only
    assert color_of(oracle, [("Push", 1), ("Push", 2), "Ne"]) == one
    assert color_of(oracle, [("Push", 1), ("Push", 2), "Lt"]) == one
    assert color_of(oracle, [("Push", 2), ("Push", 2), "Le"]) == one
    assert color_of(oracle, [("Push", 1), ("Push", 2), "Gt"]) == zero
    assert color_of(oracle, [("Push", 2), ("Push", 2), "Ge"]) == one
    assert color_of(oracle, [("Push", 2), ("Push", 0), "And"]) == zero
    assert color_of(oracle, [("Push", 2), ("Push", 0), "Or"]) == one
    assert color_of(oracle, [("Push", 0), "Not"]) == one
    nan = float("nan")
    assert color_of(oracle, [("Push", nan), ("Push", nan), "Eq"]) == zero
    assert color_of(oracle, [("Push", nan), ("Push", nan), "Ne"]) == one


def test_control_flow(oracle):
    # If: any non-zero x (NaN included) takes the then-branch
    assert color_of(oracle, [("Push", 2.0), ("If", [("Push", 1, 0, 0)], [("Push", 0, 1, 0)])]) == (1, 0, 0)
    assert color_of(oracle, [("Push", 0.0), ("If", [("Push", 1, 0, 0)], [("Push", 0, 1, 0)])]) == (0, 1, 0)
    assert color_of(oracle, [("Push", float("nan")), ("If", [("Push", 1, 0, 0)], [("Push", 0, 1, 0)])]) == (1, 0, 0)
    assert color_of(oracle, [("Push", 7.0), ("Push", 0.0), ("If", [("Push", 1, 0, 0)], None)]) == (7, 7, 7)
    # For: sum of 0..4, temporaries left by the body are truncated away
    loop = [("Push", 0.0), ("StoreLocal", 0),
            ("For", [("Push", 0.0), ("StoreLocal", 1)], [("LoadLocal", 1), ("Push", 5.0), "Lt"],
             [("LoadLocal", 1), ("Push", 1.0), "Add", ("StoreLocal", 1)],
             [("LoadLocal", 0), ("LoadLocal", 1), "Add", ("StoreLocal", 0), ("Push", 99.0)]),
            ("LoadLocal", 0)]
    assert color_of(oracle, loop, shade_locals=2) == (10, 10, 10)
    # FunctionCall: arguments in call order, fresh zeroed locals, exactly one return value, temporaries dropped
    f = [("LoadLocal", 0), ("LoadLocal", 1), "Sub", ("LoadLocal", 2), "Add", ("Push", 123.0), "Swap", "Return"]
    assert color_of(oracle, [("Push", 5.0), ("Push", 10.0), ("Push", 3.0), ("FunctionCall", 2, 3, 1), "Add"], functions=[f]) == (12, 12, 12)  # 5 + ((10 - 3) + 0)
    # without Return the top of the stack is the result; an empty body returns zero
    assert color_of(oracle, [("Push", 4.0), ("FunctionCall", 1, 1, 1)], functions=[[("LoadLocal", 0), "Dup", "Mul"]]) == (16, 16, 16)
    assert color_of(oracle, [("Push", 4.0), ("FunctionCall", 1, 1, 1)], functions=[[]]) == (0, 0, 0)
    # Return inside If unwinds through the nested block; ops after it do not run
    g = [("LoadLocal", 0), ("Push", 0.0), "Gt", ("If", [("Push", 1.0), "Return"], None), ("Push", 2.0), "Return"]
    assert color_of(oracle, [("Push", 5.0), ("FunctionCall", 1, 1, 1)], functions=[g]) == (1, 1, 1)
    assert color_of(oracle, [("Push", -5.0), ("FunctionCall", 1, 1, 1)], functions=[g]) == (2, 2, 2)
    # recursion: factorial
    fact = [("LoadLocal", 0), ("Push", 1.0), "Le", ("If", [("Push", 1.0), "Return"], None),
            ("LoadLocal", 0), ("LoadLocal", 0), ("Push", 1.0), "Sub", ("FunctionCall", 1, 1, 1), "Mul", "Return"]
    assert color_of(oracle, [("Push", 5.0), ("FunctionCall", 1, 1, 1)], functions=[fact]) == (120, 120, 120)


def test_fields(oracle):
    out = run(oracle, ["UV", "Hitpoint", "Add", "Time", "Add", "SetColor", ("Push", 0.25), "SetRoughness", ("Push", 0.75), "SetMetallic",
                       ("Push", 0.5), "SetOpacity", ("Push", 0, 0, 2), "SetNormal", ("Push", 1, 2, 3), "SetBump", "Color", "SetUV"],
              uv=(1, 2, 3), hitpoint=(10, 20, 30), time=(100, 100, 100))
    assert out["color"] == (111, 122, 133) and out["uv"] == (111, 122, 133)
    assert out["roughness"] == (0.25,) * 3 and out["metallic"] == (0.75,) * 3 and out["opacity"] == (0.5,) * 3
    assert out["normal"] == (0, 0, 1) and out["bump"] == (1, 2, 3)       # SetNormal normalises
    assert run(oracle, ["Roughness", "SetColor"])["color"] == (0.5, 0.5, 0.5)  # Execution::new


def test_palette_and_patterns(oracle):
    pal = [(0.1, 0.2, 0.3), None, (0.4, 0.5, 0.6)]
    assert color_of(oracle, [("Push", 2.9), "PaletteIndex"], palette=pal) == pytest.approx((0.4, 0.5, 0.6))
    # a missing or empty slot pushes NOTHING: the value below it becomes the result
    assert color_of(oracle, [("Push", 7.0), ("Push", 1.0), "PaletteIndex"], palette=pal) == (7, 7, 7)
    assert color_of(oracle, [("Push", 7.0), ("Push", 9.0), "PaletteIndex"], palette=pal) == (7, 7, 7)
    tex = np.arange(2 * 4 * 3, dtype=np.float32).reshape(2, 4, 3)  # h = 2, w = 4
    # uv = (0.6, 0.75): x = floor(0.6 * 4) = 2, y = floor(0.75 * 2) = 1; wraps by x - floor(x)
    assert color_of(oracle, [("Push", 0.6, 0.75, 0), ("Push", 0.0), "Sample"], patterns=[tex]) == tuple(tex[1, 2])
    assert color_of(oracle, [("Push", -0.4, 1.75, 0), ("Push", 0.0), "Sample"], patterns=[tex]) == tuple(tex[1, 2])
    assert color_of(oracle, [("Push", 0.6, 0.75, 0), ("Push", 3.0), "Sample"], patterns=[tex]) == (0, 0, 0)  # no such pattern
    nm = np.full((1, 1, 3), 0.75, np.float32)
    assert color_of(oracle, [("Push", 0.1, 0.1, 0), ("Push", 0.0), "SampleNormal"], normal_patterns=[nm]) == (0.5, 0.5, 0.5)


def test_faults_where_the_reference_panics(oracle):
    assert run(oracle, ["Add"]) is None                                     # pop().unwrap() on an empty stack
    assert run(oracle, [("LoadLocal", 3)], shade_locals=2) is None           # locals[3]
    assert run(oracle, [("LoadGlobal", 0)]) is None                          # globals[0] with zero globals
    assert run(oracle, [("FunctionCall", 0, 0, 7)]) is None                  # user_functions[7]
    assert run(oracle, [("Push", 1), ("Push", 2), ("Push", 1), "Clamp"]) is None  # f32::clamp: min > max
    assert run(oracle, [("Push", 1.0), ("Push", 1.0), "Alloc"]) is None


def test_state_leaks_between_fragments_of_a_tile(oracle):
    """The reference keeps one Execution per tile (src/rasterizer.rs:310): a program's emissive reaches every later
    fragment of that tile, including batches without a program.  The oracle reproduces it (which is why the device
    rejects SetEmissive instead of pretending)."""
    from rusterix_amd import binding as B

    def frame(tile_size):
        scene = oracle.Scene.empty()
        prog = scene.add_program(Program([[("Push", 0.5, 0.0, 0.0), "SetEmissive"]]))
        a = oracle.Batch3D.from_box(-0.9, -0.5, -0.5, 0.8, 1.0, 1.0).with_computed_normals().source(B.PixelSource.Pixel((0, 0, 0, 255)))
        a.shader(prog)
        b = oracle.Batch3D.from_box(0.1, -0.5, -0.5, 0.8, 1.0, 1.0).with_computed_normals().source(B.PixelSource.Pixel((0, 0, 0, 255)))
        scene.add_d3_static(a).add_d3_static(b)
        cam = oracle.D3OrbitCamera.new()
        cam.set_parameter_f32("distance", 3.0)
        v, p = cam.matrices(64.0, 32.0)
        out = np.zeros(64 * 32 * 4, np.uint8)
        oracle.Rasterizer.setup(None, v, p).rasterize(scene, out, 64, 32, tile_size, oracle.Assets.default())
        return out.reshape(32, 64, 4)

    one_tile = frame(64)
    left, right = one_tile[16, 24:31, 0], one_tile[16, 33:40, 0]   # the two boxes, either side of x = 32
    assert (left > 0).all() and (right > 0).all()   # the box WITHOUT a program glows too: emissive leaked within the tile
    split = frame(32)                               # now each box has its own tile, i.e. its own Execution
    glow = [(split[16, 24:31, 0] > 0).all(), (split[16, 33:40, 0] > 0).all()]
    assert sorted(glow) == [False, True]            # only the box that runs the program glows
