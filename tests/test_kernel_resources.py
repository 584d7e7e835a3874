"""Guards on the code object of the hot kernels (CPU only: hipcc cross-compiles gfx950 without a GPU).

k_raster is fp32-VALU bound and sits at the edge of the SGPR file: merely compiling an editor-only path (the grid
background) into it once added 9 % VALU instructions per frame as SGPR spill traffic, without any test noticing.  Rarer paths
therefore live in the feature levels >= 1 (k_raster_chunk / k_raster_vm); this test keeps it that way by checking the
registers, the scratch and the spill instructions of the common kernels in the ISA the build flags produce."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    import __graft_entry__ as G

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "rxr_kernels.s"
    flags = [f for f in G.HIP_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", "-o", str(out), os.path.join(G.CSRC, "rxr_kernels.hip")],
                   check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernel_body(isa, name):
    start = isa.index(f"\n{name}:")
    return isa[start:isa.index("s_endpgm", start)]


def descriptor(isa, name, key):
    m = re.search(rf"\.amdhsa_kernel {name}\n(.*?)\.end_amdhsa_kernel", isa, flags=re.S)
    return int(re.search(rf"\.amdhsa_{key} (\d+)", m.group(1)).group(1))


@pytest.mark.parametrize("k_raster", ["k_raster", "k_raster_rl"])  # (_rl: the relaxed light loop, RXR_LIGHT_MATH / rxr_set_light_math)
def test_k_raster_keeps_its_registers(isa, k_raster):
    assert descriptor(isa, k_raster, "next_free_vgpr") <= 64          # 8 waves per SIMD
    assert descriptor(isa, k_raster, "private_segment_fixed_size") <= 16
    spills = len(re.findall(r"v_(?:writelane|readlane)_b32", kernel_body(isa, k_raster)))
    # 17 with the parameter block read in place, 369 by value; 85 since the light records live in SGPRs (scalar loads through the
    # constant address space, 22 dwords per light): 55 of them outside every loop, none inside the 3D light loop; 104 since the
    # implicit-list walk classifies covering candidates (a scalar entry word and one more uniform branch per candidate; A-B neutral in
    # time, profiles/r03/bench_kernel_experiments.txt)
    # (k_raster_rl: 115 since the culling step in front of the light loop reads each light's LightFast record in one round of loads --
    # twelve more outside every loop, none inside the light loop; A-B -0.8 % at 16 lights, -2.3 % at 4)
    assert spills <= (120 if k_raster == "k_raster_rl" else 110), f"{spills} SGPR spill / reload instructions in {k_raster}"
    assert "v_pk_fma_f32" not in kernel_body(isa, k_raster), "packed f32 (SLP vectorisation) is slower on gfx950: build with -fno-slp-vectorize"


def test_the_relaxed_kernel_carries_the_fused_point_light_path(isa):
    """k_raster_rl = k_raster + the fused point-light term (two v_rsq_f32 normalisations, the smoothstep clamp as an output modifier, native
    v_log_f32 / v_exp_f32 without their range votes); the exact sequences stay as the fallback for out-of-window magnitudes"""
    exact, relaxed = kernel_body(isa, "k_raster"), kernel_body(isa, "k_raster_rl")
    assert relaxed.count("v_rsq_f32") > exact.count("v_rsq_f32")
    assert len(re.findall(r"v_mul_f32_e64 .* clamp", relaxed)) > len(re.findall(r"v_mul_f32_e64 .* clamp", exact))   # med3(q, 0, 1) as the output modifier
    assert relaxed.count("v_div_fixup_f32") >= 3   # (the compiler's division expansion: the fallback is still there)


@pytest.mark.parametrize("k_raster_rows", ["k_raster_rows", "k_raster_rows_rl"])
def test_k_raster_rows_keeps_its_occupancy(isa, k_raster_rows):
    assert descriptor(isa, k_raster_rows, "next_free_vgpr") <= 64     # 8 waves per SIMD
    assert descriptor(isa, k_raster_rows, "private_segment_fixed_size") <= 64
    assert descriptor(isa, k_raster_rows, "group_segment_fixed_size") <= 20 * 1024   # 8 workgroups per CU in 160 KB


def test_row_mode_uses_the_lds_atomic(isa):
    assert "ds_min_u64" in kernel_body(isa, "k_raster_rows")


def test_k_raster_chunk_has_no_calls_and_little_scratch(isa):
    """k_raster_chunk ran the bench frame 1.6x slower than k_raster at the same instruction count while the opacity staircase
    lived in scratch (a run-time array index) and a real call forced the call ABI on it"""
    body = kernel_body(isa, "k_raster_chunk")
    assert "s_swappc_b64" not in body
    assert descriptor(isa, "k_raster_chunk", "next_free_vgpr") <= 64
    assert descriptor(isa, "k_raster_chunk", "private_segment_fixed_size") <= 64
    assert descriptor(isa, "k_raster_chunk", "group_segment_fixed_size") <= 20 * 1024


def test_interpreter_kernels_keep_six_waves(isa):
    """k_raster_vm*: 6 waves per SIMD need <= 80 VGPRs and 6 workgroups per CU need <= 160 KB / 6 of LDS each"""
    for k in ("k_raster_vm", "k_raster_vm_s", "k_raster_vm_sv", "k_raster_vm_v"):
        assert descriptor(isa, k, "next_free_vgpr") <= 80, k
        assert descriptor(isa, k, "group_segment_fixed_size") <= 160 * 1024 // 6, k


def test_interpreter_dispatch_is_a_tree_of_scalar_branches(isa):
    """the opcode dispatch must not come back as one structurized switch: that form handed the whole interpreter state through
    flow blocks (30-60 v_mov per VM instruction, rxr_vm.h).  The register moves of the whole kernel are the fingerprint."""
    body = kernel_body(isa, "k_raster_vm_sv")
    moves = len(re.findall(r"\bv_mov_b32", body))
    assert moves < 3200, f"{moves} v_mov_b32 in k_raster_vm_sv"
