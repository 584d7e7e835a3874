"""Program sets compiled at run time (rusterix_amd/csrc/rxr_jit.hip, RXR_SHADER_JIT=1): the same jump code as straight-line device
code instead of the interpreter.  Every frame here is rendered three ways -- CPU oracle, interpreter kernels, compiled kernels --
and the compiled frame must EQUAL the interpreter's (same float operations in the same order) and meet the oracle within the
interpreter's own tolerance (exact for the arithmetic opcodes, +-1 for libm).  `rxr_debug_jit_info` proves which path ran."""
import ctypes as C

import numpy as np
import pytest

import rusterix_amd
from rusterix_amd import binding as B
from rusterix_amd import scenes
from rusterix_amd.binding import Program
from tests import test_gpu_shaders as S

pytestmark = pytest.mark.gpu
W, H = 192, 128


def jit_info(product):
    rxr = C.CDLL(rusterix_amd.lib_paths()["rxr"])
    rxr.rxr_debug_jit_info.restype = C.c_char_p
    rxr.rxr_debug_jit_info.argtypes = [C.c_void_p]
    product.lib.rxh_context.restype = C.c_void_p
    return rxr.rxr_debug_jit_info(product.lib.rxh_context()).decode()


def grid_scene(api, programs, time=0.25, lights=False):
    """one textured 2D rectangle per program, side by side (every program of the set runs in the same frame)"""
    scene = api.Scene.empty()
    n = len(programs)
    cols = int(np.ceil(np.sqrt(n)))
    rows = (n + cols - 1) // cols
    cw, ch = W / cols, H / rows
    for k, prog in enumerate(programs):
        idx = scene.add_program(prog)
        r = api.Batch2D.from_rectangle(float(np.float32((k % cols) * cw)), float(np.float32((k // cols) * ch)), float(np.float32(cw)), float(np.float32(ch)))
        r.source(B.PixelSource.StaticTileIndex(0)).shader(idx)
        scene.add_d2_static(r)
    if lights:
        scene.lights([B.Light(B.LIGHT_POINT).with_position((60.0, 0.0, 40.0)).with_color((1.0, 0.9, 0.8)).with_intensity(1.5)
                      .with_start_distance(10.0).with_end_distance(120.0).compile()])
    assets = api.Assets.default().textures([B.Tile.from_texture(scenes.noise_texture(5, 32, 32))])
    assets.patterns(S.patterns()).patterns(S.patterns()[::-1], normal=True).palette([(0.9, 0.1, 0.2), None, (0.2, 0.3, 0.9)])

    def setup():
        return api.Rasterizer.setup(None, B.Mat4.identity(), B.Mat4.identity()).time(time)

    return scenes._result(api, scene, assets, setup, W, H, 40, "jit-grid")


def three_ways(oracle, product, monkeypatch, build, tol=0, max_off=0):
    monkeypatch.setenv("RXR_SHADER_JIT", "0")
    interp = scenes.render(build(product)).copy()
    assert jit_info(product) == ""
    monkeypatch.setenv("RXR_SHADER_JIT", "1")
    got = scenes.render(build(product)).copy()
    info = jit_info(product)
    assert info.startswith("compiled:"), info
    monkeypatch.setenv("RXR_SHADER_JIT", "0")
    assert np.array_equal(got, interp), f"compiled and interpreted frames differ in {(got != interp).any(axis=2).sum()} pixels; first at {np.argwhere((got != interp).any(axis=2))[:3].tolist()}"
    ref = scenes.render(build(oracle))
    diff = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
    assert (diff > tol).sum() <= max_off, f"{(diff > tol).sum()} pixels differ from the oracle by more than {tol} (max {diff.max()})"
    return got, info


def test_every_exact_opcode(oracle, product, monkeypatch):
    programs = [Program([S.A + [op] + S.TO_COLOR]) for op in S.EXACT_UNARY] + [Program([S.A + S.Bv + [op] + S.TO_COLOR]) for op in S.EXACT_BINARY]
    got, info = three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, programs))
    assert "template level" in info
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 500


def test_libm_opcodes(oracle, product, monkeypatch):
    programs = [Program([S.A + [op] + S.TO_COLOR]) for op in S.LIBM_UNARY]
    programs += [Program([(S.A + [("Push", 37.0)] if op == "Rotate2D" else S.A + S.Bv) + [op] + S.TO_COLOR]) for op in S.LIBM_BINARY]
    three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, programs), tol=S.TOLERANCE)


def test_ternaries_swizzles_fields_patterns_loops(oracle, product, monkeypatch):
    tern = Program([S.A + S.Bv + [("Push", 0.3, 0.6, 0.9), "Mix", ("Push", -1.0), ("Push", 1.0), "Clamp",
                                  ("Push", 0.0), ("Push", 1.0), "UV", ("GetComponents", [0]), ("Push", 4.0), "Mul", "Smoothstep", "Mul",
                                  "Color", "Add", "Hitpoint", ("Push", 0.001), "Mul", "Add", "Time", ("Push", 0.1), "Mul", "Add",
                                  "Dup", ("GetComponents", [2, 0]), ("SetComponents", [1, 2]), ("Push", 0.5), "Mul", "SetColor"]])
    pats = Program([["UV", ("Push", 8.0), "Mul", ("Push", 0.0), "Sample", "UV", ("Push", 5.0), "Mul", ("Push", 1.0), "SampleNormal", ("Push", 0.25), "Mul", "Add",
                     "UV", ("Push", 3.0), "Mul", ("Push", 9.0), "Sample", "Add", "SetColor"]])
    loops = Program([[("Push", 0.0), ("StoreLocal", 0), "UV", ("GetComponents", [0]), ("Push", 40.0), "Mul", "Floor", ("StoreLocal", 2),
                      ("For", [("Push", 0.0), ("StoreLocal", 1)], [("LoadLocal", 1), ("LoadLocal", 2), "Lt"],
                       [("LoadLocal", 1), ("Push", 1.0), "Add", ("StoreLocal", 1)],
                       [("LoadLocal", 1), ("Push", 2.0), "Mod", ("Push", 0.0), "Eq",
                        ("If", [("LoadLocal", 0), ("Push", 0.07), "Add", ("StoreLocal", 0)], [("LoadLocal", 0), ("Push", 0.02), "Add", ("StoreLocal", 0)]),
                        ("Push", 5.0)]),
                      ("LoadLocal", 0), "UV", ("GetComponents", [1]), ("Push", 4.0), "Mul", ("LoadLocal", 2), ("Push", 0.04), "Mul", "Pack3", "SetColor"]], shade_locals=3)
    glob = Program([["UV", ("StoreGlobal", 1), ("LoadGlobal", 1), ("Push", 4.0), "Mul", "UV", "Swap", "Clear", ("Push", 0.5), "Add", "SetColor"]], globals=2)
    early = Program([["UV", ("GetComponents", [0]), ("Push", 0.05), "Lt", ("If", [("Push", 1.0, 0.0, 0.0), "SetColor", "Return"], None), "UV", ("Push", 4.0), "Mul", "SetColor"]])
    got, _ = three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, [tern, pats, loops, glob, early], time=0.75, lights=True))
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 200


def test_random_programs(oracle, product, monkeypatch):
    programs = []
    for seed in range(14):
        rng = np.random.default_rng([0x52585231, 9001, seed])
        programs.append(S.ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=0, setters=["SetColor"]).program())
    three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, programs, time=0.5), tol=S.TOLERANCE, max_off=6)


@pytest.mark.parametrize("seed", [1, 4, 9])
def test_random_cube_materials(oracle, product, monkeypatch, seed):
    """3D opaque pass: colour / roughness / metallic of a compiled program feed the lighting (log2 / exp2 against libm: +-1)"""
    rng = np.random.default_rng([0x52585231, 778, seed])
    prog = S.ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=0).program()
    three_ways(oracle, product, monkeypatch, lambda api: S.cube_scene(api, prog), tol=S.TOLERANCE, max_off=5)


def test_program_that_decides_visibility_and_the_opacity_pass(oracle, product, monkeypatch):
    """SetOpacity in the opaque pass (the visibility loop runs the compiled program: k_raster_jit_v) and a program on an opacity-list batch"""
    cut = Program([["UV", ("Push", 8.0), "Mul", "Fract", ("GetComponents", [0]), ("Push", 0.5), "Lt", ("If", [("Push", 0.0), "SetOpacity"], [("Push", 1.0), "SetOpacity"]),
                    "Color", ("Push", 0.9), "Mul", "SetColor"]])
    three_ways(oracle, product, monkeypatch, lambda api: S.cube_scene(api, cut), tol=S.TOLERANCE, max_off=5)
    tint = Program([["Color", ("Push", 0.5, 1.0, 0.7), "Mul", "SetColor", ("Push", 0.6), "SetOpacity"]])

    def pane_with_a_chunk_program(api):
        """a chunk's opacity-list batch runs the CHUNK's program (chunk.shaders, rasterizer.rs:1645-1648), over an opaque box"""
        cfg = S.cube_scene(api, Program([["Color", "SetColor"]]))
        chunk = cfg.scene.add_chunk()
        chunk.add_shader(tint)
        pane = api.Batch3D.from_box(-0.8, -0.8, 0.7, 1.6, 1.6, 0.02).with_computed_normals().source(B.PixelSource.Pixel((200, 220, 90, 255))).shader(0)
        chunk.add_batch3d_opacity(pane)
        return cfg

    got, _ = three_ways(oracle, product, monkeypatch, pane_with_a_chunk_program, tol=S.TOLERANCE, max_off=5)
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 100


def test_faults_are_reported_by_compiled_programs(product, monkeypatch):
    monkeypatch.setenv("RXR_SHADER_JIT", "1")
    bad = Program([["UV", ("Push", 1.0), ("Push", 0.0), "Clamp", "SetColor"]])  # clamp with min > max: the reference panics
    with pytest.raises(B.RasterizeError) as e:
        scenes.render(grid_scene(product, [bad]))
    assert e.value.code == B.RXR_ERR_INVALID and "clamp" in str(e.value).lower()
    assert jit_info(product).startswith("compiled:")


def test_functions_are_compiled_too(oracle, product, monkeypatch):
    """FunctionCall: arguments become parameters, every callee its own straight-line function with fresh locals; Return from inside
    an If; a callee calling a callee; globals shared between caller and callee"""
    helper = [("LoadLocal", 0), ("Push", 0.5), "Gt", ("If", [("LoadLocal", 0), ("Push", 0.5), "Sub", "Return"], None), ("LoadLocal", 0), "Return"]
    twice = [("LoadLocal", 0), ("FunctionCall", 1, 1, 1), ("LoadLocal", 1), ("FunctionCall", 1, 1, 1), "Add", ("StoreLocal", 2), ("LoadLocal", 2)]   # falls off its end: top of its stack
    shade = ["UV", ("GetComponents", [1]), ("Push", 4.0), "Mul", ("FunctionCall", 1, 1, 1),
             "UV", ("GetComponents", [0]), ("Push", 4.0), "Mul", "UV", ("GetComponents", [1]), ("Push", 2.0), "Mul", ("FunctionCall", 2, 3, 2),
             ("Push", 0.25), "Pack3", "SetColor"]
    with_globals = Program([["UV", ("StoreGlobal", 1), ("LoadGlobal", 1), ("Push", 4.0), "Mul", ("FunctionCall", 0, 0, 1), "Add", "SetColor"],
                            [("LoadGlobal", 1), ("Push", 1.0), "Mul"]], globals=2)
    got, info = three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, [Program([shade, helper, twice]), with_globals]))
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 100


def test_random_programs_with_functions(oracle, product, monkeypatch):
    programs = []
    for seed in range(12):
        rng = np.random.default_rng([0x52585231, 9002, seed])
        programs.append(S.ProgramGen(rng, n_locals=int(rng.integers(3, 6)), n_functions=int(rng.integers(1, 3)), setters=["SetColor"]).program())
    three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, programs, time=0.5), tol=S.TOLERANCE, max_off=6)


FACT = [("LoadLocal", 0), ("Push", 1.0), "Le", ("If", [("Push", 1.0), "Return"], None),
        ("LoadLocal", 0), ("LoadLocal", 0), ("Push", 1.0), "Sub", ("FunctionCall", 1, 1, 1), "Mul", "Return"]


def test_recursive_programs_are_compiled_as_one_copy_per_call_depth(oracle, product, monkeypatch):
    """the interpreter bounds every call chain by its frame stack, so recursion unrolls into an acyclic program (rxr_jit.hip,
    generate_program_levels): factorial (one recursive call site: inlined copies), Fibonacci (two call sites: 2^8 chains, real
    functions), a mutually recursive pair with different locals"""
    fact_prog = Program([["UV", ("GetComponents", [0]), ("Push", 24.0), "Mul", "Floor", ("FunctionCall", 1, 1, 1), ("Push", 0.004), "Mul", "SetColor"], FACT])
    fib = [("LoadLocal", 0), ("Push", 2.0), "Lt", ("If", [("LoadLocal", 0), "Return"], None),
           ("LoadLocal", 0), ("Push", 1.0), "Sub", ("FunctionCall", 1, 1, 1), ("LoadLocal", 0), ("Push", 2.0), "Sub", ("FunctionCall", 1, 1, 1), "Add", "Return"]
    fib_prog = Program([["UV", ("GetComponents", [1]), ("Push", 28.0), "Mul", "Floor", ("FunctionCall", 1, 1, 1), ("Push", 0.07), "Mul", "SetColor"], fib])
    # even(n) = n == 0 ? 1 : odd(n - 1);  odd(n) = n == 0 ? 0 : even(n - 1)   (odd keeps a second local)
    even = [("LoadLocal", 0), ("Push", 0.0), "Eq", ("If", [("Push", 1.0), "Return"], None), ("LoadLocal", 0), ("Push", 1.0), "Sub", ("FunctionCall", 1, 2, 2), "Return"]
    odd = [("LoadLocal", 0), ("StoreLocal", 1), ("LoadLocal", 1), ("Push", 0.0), "Eq", ("If", [("Push", 0.0), "Return"], None),
           ("LoadLocal", 1), ("Push", 1.0), "Sub", ("FunctionCall", 1, 1, 1), "Return"]
    parity_prog = Program([["UV", ("GetComponents", [0]), ("Push", 28.0), "Mul", "Floor", ("FunctionCall", 1, 1, 1), "UV", "Mul", "SetColor"], even, odd])
    got, info = three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, [fact_prog, fib_prog, parity_prog]))
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 12


def test_min_and_max_of_signed_zeros_return_the_first_operand(oracle, product, monkeypatch):
    """f32::min / f32::max of +0.0 and -0.0 ("either may be returned" in Rust's documentation; `self` in what rustc's x86-64 back end
    emits): the sign shows when a program divides by the result.  Found by the fuzz sweep (seed 52871): the device's v_max_f32 orders
    the zeros, glibc's fmax returns the other operand -- oracle and device now both follow the x86 lowering.  Variables and fused
    constants ("Push c; op"), both orders."""
    z = lambda s: ["UV", ("GetComponents", [0]), ("Push", 0.0), "Mul", ("Push", s), "Mul"]   # a run-time zero of either sign (uv.x * 0 * (+-1))
    programs = []
    for op in ("Min", "Max"):
        for sa in (1.0, -1.0):
            for sb in (1.0, -1.0):
                programs.append(Program([[("Push", 1.0)] + z(sa) + z(sb) + [op, "Div", ("Push", 1e-30), "Mul", ("Push", 0.5), "Add", "SetColor"]]))          # 1 / op(a, b): +-inf -> 1 or 0
                programs.append(Program([[("Push", 1.0)] + z(sa) + [("Push", 0.0 * sb), op, "Div", ("Push", 1e-30), "Mul", ("Push", 0.5), "Add", "SetColor"]]))  # the fused-constant form
    got, _ = three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, programs))
    assert len(np.unique(got.reshape(-1, 4), axis=0)) >= 2   # both signs occur


def random_recursive_program(seed):
    """a random program (S.ProgramGen: two helper functions) plus a self-recursive function R(n, x) -- early exit at n <= 0, a random
    block, R(n - 1, <random>), a random combination -- called from `shade` with n = 0 .. 2 by pixel.  R may call the helpers; frames
    stay within the interpreter's 8 and the chain's locals within its 48"""
    rng = np.random.default_rng([0x52585231, 9003, seed])
    gen = S.ProgramGen(rng, n_locals=3, n_functions=2, setters=["SetColor"])
    prog = gen.program()
    shade, helpers = gen.raw[0], gen.raw[1:]
    gen.n_locals = gen.loadable = 3
    gen.first_callable = 0
    # (the counter moves to local 7 first: the random statements store to locals 0 .. 2 and loop on 3 .. 6)
    rec = ([("LoadLocal", 0), ("StoreLocal", 7), ("LoadLocal", 7), ("Push", 0.0), "Le", ("If", gen.value(2) + ["Return"], None)] + gen.block(2) +
           [("LoadLocal", 7), ("Push", 1.0), "Sub"] + gen.value(2) + [("FunctionCall", 2, 8, 3)] +
           gen.value(2) + [str(rng.choice(["Add", "Mul", "Min", "Max", "Sub"])), "Return"])
    assert shade[-1] == "SetColor"
    call = ["UV", ("GetComponents", [0]), ("Push", 11.0), "Mul", "Fract", ("Push", 3.0), "Mul", "Floor", "UV", ("FunctionCall", 2, 8, 3), ("Push", 0.37), "Mul", "Add", "Fract"]
    return Program([shade[:-1] + call + ["SetColor"]] + helpers + [rec], shade_locals=3 + 4)


def test_random_recursive_programs(oracle, product, monkeypatch):
    programs = [random_recursive_program(seed) for seed in range(10)]
    three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, programs, time=0.5), tol=S.TOLERANCE, max_off=6)


def test_recursion_beyond_the_frame_stack_faults_in_both_forms(product, monkeypatch):
    """nine nested calls: the interpreter raises VMF_CALL_DEPTH at the ninth, the compiled form at the call site of its last copy"""
    deep = Program([["UV", ("GetComponents", [0]), ("Push", 64.0), "Mul", "Floor", ("FunctionCall", 1, 1, 1), ("Push", 0.001), "Mul", "SetColor"], FACT])  # (uv / 4: up to 15)
    for mode, compiled in (("0", False), ("1", True)):
        monkeypatch.setenv("RXR_SHADER_JIT", mode)
        with pytest.raises(B.RasterizeError) as e:
            scenes.render(grid_scene(product, [deep]))
        assert e.value.code == B.RXR_ERR_INVALID, str(e.value)
        assert jit_info(product).startswith("compiled:") == compiled, jit_info(product)
    monkeypatch.setenv("RXR_SHADER_JIT", "0")


def test_palette_lookups_with_a_missing_slot_go_back_to_the_interpreter(oracle, product, monkeypatch):
    monkeypatch.setenv("RXR_SHADER_JIT", "1")
    # (a missing palette slot pushes nothing: the Add then consumes the two constants -- a data-dependent stack depth)
    pal = Program([[("Push", 0.1, 0.1, 0.1), ("Push", 0.2, 0.3, 0.4), "UV", ("GetComponents", [0]), ("Push", 12.0), "Mul", "PaletteIndex", "Add", "Clear", "UV", "SetColor"]])
    got = scenes.render(grid_scene(product, [pal]))
    assert jit_info(product).startswith("not compiled:")
    assert np.array_equal(got, scenes.render(grid_scene(oracle, [pal])))


def test_box_grid_with_the_configuration_c5_program(oracle, product, monkeypatch):
    """the per-batch program of BASELINE.json's C5 on the reduced grid (binned path, row mode, Linear sampling)"""
    build = lambda api: scenes.box_grid_scene(api, n=32, width=640, height=360, shader=True)  # noqa: E731
    three_ways(oracle, product, monkeypatch, build, tol=0, max_off=0)


def test_background_compilation_switches_over_without_changing_a_pixel(product, monkeypatch):
    """the default mode (RXR_SHADER_JIT unset or "async"): the interpreter renders until the child process (rxr_jitc) has compiled the kernel, then the compiled
    kernel does; every frame on the way is the same frame"""
    import time

    prog = Program([["UV", ("Push", 3.25), "Mul", "Fract", "Color", ("Push", 0.37, 0.91, 0.53), "Mul", "Add", "SetColor"]])
    monkeypatch.setenv("RXR_SHADER_JIT", "0")
    want = scenes.render(grid_scene(product, [prog])).copy()
    monkeypatch.delenv("RXR_SHADER_JIT", raising=False)   # (the default IS the background mode)
    cfg = grid_scene(product, [prog])
    first = scenes.render(cfg).copy()
    seen = [jit_info(product)]
    assert np.array_equal(first, want)
    assert seen[0].startswith("compiling in the background"), seen
    deadline = time.time() + 120.0
    while not jit_info(product).startswith("compiled:") and time.time() < deadline:
        assert np.array_equal(scenes.render(cfg), want)
        if jit_info(product) != seen[-1]:
            seen.append(jit_info(product))
        time.sleep(0.05)
    assert jit_info(product).startswith("compiled:"), seen
    assert np.array_equal(scenes.render(cfg), want)
    # a set that is replaced while its compilation runs: the child is killed, nothing is left behind, the new set compiles
    other = Program([["UV", ("Push", 5.5), "Mul", "Fract", "SetColor"]])
    scenes.render(grid_scene(product, [other]))
    assert jit_info(product).startswith("compiling in the background")
    scenes.render(grid_scene(product, [prog]))     # (replaces `other` at once)
    monkeypatch.setenv("RXR_SHADER_JIT", "0")
    assert np.array_equal(scenes.render(grid_scene(product, [prog])), want)



def test_palette_index_is_compiled_while_every_index_has_a_colour(oracle, product, monkeypatch):
    """PaletteIndex -- the one opcode whose stack effect depends on data (a missing or empty slot pushes nothing,
    rusteria/src/node/execution.rs:742-749) -- used to leave a whole set with the interpreter.  Compiled as the push: this frame only
    ever asks for slots 0 and 2 of the palette (0.9, 0.1, 0.2), None, (0.2, 0.3, 0.9)."""
    # (the 2D pass hands uv / 4 to the program: uv.x runs over [0, 0.25) across the rectangle)
    prog = Program([["UV", ("GetComponents", [0]), ("Push", 7.96), "Mul", "Floor", ("Push", 2.0), "Mul", "PaletteIndex",   # 0 or 2
                     "UV", ("GetComponents", [1]), "Mul", "Color", ("Push", 0.25), "Mul", "Add", "SetColor"]])
    plain = Program([S.A + ["Abs"] + S.TO_COLOR])
    got, info = three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, [prog, plain]))
    assert "compiled:" in info
    assert len(np.unique(got[:, : W // 2].reshape(-1, 4), axis=0)) > 50


def test_palette_miss_sends_the_set_back_to_the_interpreter(oracle, product, monkeypatch):
    """... and a frame that DOES meet the empty slot (index 1 in the middle third of the rectangle: nothing is pushed, the Add consumes
    the value below instead) cannot be followed by straight-line code: the compiled kernel raises VMF_JIT_PALETTE_MISS, rxr_synchronize
    hands the set to the interpreter and renders the frame again -- the caller sees the interpreter's frame, this time and afterwards."""
    prog = Program([[("Push", 0.5, 0.4, 0.3), ("Push", 0.1, 0.2, 0.3), "UV", ("GetComponents", [0]), ("Push", 11.96), "Mul", "PaletteIndex", "Add", "SetColor"]])
    build = lambda api: grid_scene(api, [prog])  # noqa: E731
    monkeypatch.setenv("RXR_SHADER_JIT", "0")
    interp = scenes.render(build(product)).copy()
    monkeypatch.setenv("RXR_SHADER_JIT", "1")
    got = scenes.render(build(product)).copy()
    info = jit_info(product)
    assert "PaletteIndex" in info and info.startswith("not compiled"), info
    again = scenes.render(build(product)).copy()   # the same set, uploaded again: compiled again, misses again, falls back again
    monkeypatch.setenv("RXR_SHADER_JIT", "0")
    ref = scenes.render(build(oracle))
    assert np.array_equal(got, interp) and np.array_equal(again, interp)
    assert np.array_equal(got, ref)
    # the three thirds differ: colour 0 + (0.1, 0.2, 0.3); nothing pushed, so the Add takes the two constants; colour 2 + (0.1, 0.2, 0.3)
    assert len({tuple(got[H // 2, x]) for x in (W // 6, W // 2, 5 * W // 6)}) == 3


def test_the_reference_crates_own_test_programs(oracle, product, monkeypatch):
    """`addition` and `fib` of rusteria/src/lib.rs:274-296 -- the only assertions the reference holds for code of this path -- as the NodeOp
    lists its compiler emits (tests/test_oracle_reference_tests.py pins the oracle's Execution to their asserted values, 4.0 and
    fib(27) = 196418.0).  Here the same two programs run on the device, interpreted and compiled, against that oracle: fib over
    n = 0 .. 5 by pixel (the device bounds a call chain by eight frames where the reference recurses on the Rust stack)."""
    from tests.test_oracle_reference_tests import FIB
    addition = Program([[("Push", 2.0), ("StoreGlobal", 0), ("LoadGlobal", 0), ("Push", 2.0), "Add", ("Push", 0.125), "Mul", "SetColor"]], globals=1)
    fib = Program([["UV", ("GetComponents", [0]), ("Push", 24.0), "Mul", "Floor", ("FunctionCall", 1, 1, 1), ("Push", 0.125), "Mul", "SetColor"], FIB])
    got, _ = three_ways(oracle, product, monkeypatch, lambda api: grid_scene(api, [addition, fib]))
    left = got[:, : W // 2 - 1, :3].reshape(-1, 3)
    assert (left == left[0]).all() and 127 <= int(left[0, 0]) <= 128            # 4 * 0.125
    assert len(np.unique(got[:, W // 2 + 1:, 0])) >= 4                           # fib(0 .. 5) / 8 = 0, 1/8, 1/8, 2/8, 3/8, 5/8


def test_compiled_programs_on_a_binned_frame_with_cut_out_batches(oracle, product, monkeypatch):
    """a program on every batch of a binned lattice AND every third batch textured with holes: the compiled kernel's cut variant
    (k_raster_jit_cut: rounds in row mode around the cut-out candidates) gives the interpreter's and the oracle's frame, and the frame of
    the plain compiled kernel (RXR_NO_SPLIT_ROUNDS)"""
    build = lambda api: scenes.box_grid_scene(api, n=20, width=640, height=360, shader=True, cutout_every=3)   # noqa: E731
    got, info = three_ways(oracle, product, monkeypatch, build)
    monkeypatch.setenv("RXR_SHADER_JIT", "1")
    monkeypatch.setenv("RXR_NO_SPLIT_ROUNDS", "1")
    plain = scenes.render(build(product)).copy()
    assert jit_info(product).startswith("compiled:")
    assert np.array_equal(plain, got)
    solid = scenes.render(scenes.box_grid_scene(oracle, n=20, width=640, height=360, shader=True))
    assert (got != solid).any(axis=2).mean() > 0.01, "the holes change nothing: the test tests nothing"
