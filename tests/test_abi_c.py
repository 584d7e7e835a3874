"""The C ABI from C (tests/abi_host.c) and the generated Rust mirror (tools/gen_ffi.py): CPU only.

VERDICT round 1: the Rust shim had drifted from the header (a 2-field rxr_chunk against the header's 9).  The mirror is now
generated from include/rxr.h, and the layout it asserts is confirmed by the C compiler on the header itself."""
import os
import subprocess
import sys

import rusterix_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_files_are_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "run `python tools/gen_ffi.py`: " + r.stdout


def test_header_compiles_as_c11_and_the_layout_asserts_hold(tmp_path):
    libdir = os.path.dirname(rusterix_amd.lib_paths()["rxr"])
    exe = tmp_path / "abi_host"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", os.path.join(ROOT, "tests", "abi_host.c"), "-o", str(exe),
                    "-L" + libdir, "-lrxr_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_host ok" in r.stdout


def test_rust_mirror_covers_the_header():
    """every struct, every function and every field of the header appears in ffi.rs (the file is generated, so this guards
    the generator's parser: a construct it does not understand must not vanish silently)"""
    import re

    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rxr.h")).read(), flags=re.S)
    rs = open(os.path.join(ROOT, "shim", "rusterix-hip-shim", "src", "ffi.rs")).read()
    for name in set(re.findall(r"\b(rxr_\w+)\s*\(", hdr)):
        assert f"pub fn {name}(" in rs, name
    for name, body in re.findall(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\w+\s*;", hdr, flags=re.S):
        assert f"pub struct {name} {{" in rs, name
        n_fields = sum(len(d.split(",")) for d in body.split(";") if d.strip())
        block = rs[rs.index(f"pub struct {name} {{"):]
        block = block[:block.index("}")]
        assert block.count("pub ") - 1 == n_fields, (name, n_fields)
    assert "pub const RXR_ABI_VERSION: u32 = 5;" in rs


def _struct_literals(src, names):
    """(struct name, set of field names) for every `name { field: .., shorthand, .. }` literal in Rust source text"""
    import re

    out = []
    for m in re.finditer(r"\b(" + "|".join(names) + r")\s*\{", src):
        # skip type positions: `-> rxr_source {` (function bodies) and `struct rxr_x {`
        before = src[max(0, m.start() - 12):m.start()]
        if "->" in before or "struct" in before or "impl" in before:
            continue
        depth, i, fields, token_start = 1, m.end(), set(), m.end()
        while depth and i < len(src):
            c = src[i]
            if c in "({[":
                depth += 1
            elif c in ")}]":
                depth -= 1
            if (depth == 1 and c == ",") or depth == 0:
                piece = src[token_start:i].strip()
                if piece:
                    fields.add(re.match(r"\w+", piece).group(0))
                token_start = i + 1
            i += 1
        out.append((m.group(1), fields))
    return out


def test_shim_uses_the_abi_as_declared():
    """the hand-written shim built a 2-field rxr_chunk against the header's 9, never called rxr_set_shaders or
    rxr_set_meshes.  No Rust compiler exists here, so check textually what a compiler would: every ffi identifier the shim uses
    exists, every struct literal names exactly the fields of the generated mirror, and the calls the docs promise are there."""
    import re

    rs = open(os.path.join(ROOT, "shim", "rusterix-hip-shim", "src", "ffi.rs")).read()
    lib = open(os.path.join(ROOT, "shim", "rusterix-hip-shim", "src", "lib.rs")).read()
    code = re.sub(r"//.*", "", lib)
    code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', code)  # string literals (environment variable names) are not identifiers
    structs = {n: set(re.findall(r"pub (\w+):", body)) for n, body in re.findall(r"pub struct (rxr_\w+) \{(.*?)\n\}", rs, flags=re.S)}
    declared = set(re.findall(r"pub const (RXR_\w+)", rs)) | set(re.findall(r"pub fn (rxr_\w+)", rs)) | set(structs) | {"rxr_ctx"}
    used = set(re.findall(r"\b(RXR_[A-Z0-9_]+|rxr_[a-z0-9_]+)\b", code))
    assert used <= declared, f"the shim uses identifiers ffi.rs does not declare: {sorted(used - declared)}"
    lits = _struct_literals(code, [n for n in structs if n != "rxr_ctx"])
    seen = {n for n, _ in lits}
    for need in ("rxr_frame", "rxr_chunk", "rxr_batch3d", "rxr_batch2d", "rxr_mesh3d", "rxr_shader_set", "rxr_program", "rxr_function", "rxr_light",
                 "rxr_texture", "rxr_tile", "rxr_occluder", "rxr_linedef", "rxr_pattern"):  # (no rxr_edges since ABI 5: the batches travel without them)
        assert need in seen, f"the shim never builds a {need}"
    for name, fields in lits:
        assert fields == structs[name], f"{name} literal: missing {sorted(structs[name] - fields)}, unknown {sorted(fields - structs[name])}"
    for call in ("rxr_create", "rxr_create_multi", "rxr_set_textures", "rxr_set_shaders", "rxr_set_meshes", "rxr_rasterize"):
        assert re.search(r"\b" + call + r"\(", code), call
    # every NodeOp variant of the header has its opcode in the serialiser
    for variant in re.findall(r"RXR_NODE_(\w+)", open(os.path.join(ROOT, "include", "rxr.h")).read()):
        if variant != "COUNT":
            assert f"RXR_NODE_{variant}" in code, variant
