#!/usr/bin/env python3
"""Generates, from include/rxr.h (the single source of truth of the C ABI):

  shim/rusterix-hip-shim/src/ffi.rs   the Rust mirror: constants, #[repr(C)] structs, the extern "C" block and
                                      compile-time asserts of every size and field offset (core::mem::offset_of!)
  tests/abi_layout_asserts.h          the same sizes / offsets as C11 _Static_asserts, compiled by tests/abi_host.c
                                      (a CPU test): the C compiler thereby confirms the layout model used for the Rust side

    python tools/gen_ffi.py            rewrite both files
    python tools/gen_ffi.py --check    exit 1 if either file is not what the header implies (tests/test_abi_c.py)

The parser handles exactly the C subset the header uses: `#define NAME <integer expr>`, anonymous and named enums,
`typedef struct NAME {...} NAME;` with scalar / pointer / fixed-array / nested-struct fields, and plain prototypes.
Layout model: LP64 natural alignment (what C on x86-64 / aarch64 Linux and Rust's repr(C) both implement).
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rxr.h")
FFI_RS = os.path.join(ROOT, "shim", "rusterix-hip-shim", "src", "ffi.rs")
ASSERTS_H = os.path.join(ROOT, "tests", "abi_layout_asserts.h")

SCALARS = {  # C type -> (Rust type, size, alignment)
    "uint8_t": ("u8", 1, 1), "uint32_t": ("u32", 4, 4), "int32_t": ("i32", 4, 4), "uint64_t": ("u64", 8, 8), "float": ("f32", 4, 4),
    "int": ("c_int", 4, 4), "size_t": ("usize", 8, 8), "char": ("c_char", 1, 1), "void": ("c_void", 0, 1),
}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def parse(text):
    text = strip_comments(text)
    defines, enums, structs, funcs = [], [], [], []
    for m in re.finditer(r"^#define\s+(RXR_\w+)\s+(.+)$", text, flags=re.M):
        name, val = m.group(1), m.group(2).strip()
        if name == "RXR_H":
            continue
        defines.append((name, val))
    for m in re.finditer(r"(typedef\s+)?enum\s*(\w*)\s*\{(.*?)\}\s*(\w*)\s*;", text, flags=re.S):
        tname = m.group(4) or m.group(2)
        items, nxt = [], 0
        for part in m.group(3).split(","):
            part = part.strip()
            if not part:
                continue
            if "=" in part:
                k, v = [x.strip() for x in part.split("=")]
                nxt = int(v, 0)
            else:
                k = part
            items.append((k, nxt))
            nxt += 1
        enums.append((tname, items))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"^(const\s+)?(\w+)\s*(\*?)\s*(.+)$", decl)
            const, ctype, star, rest = fm.group(1), fm.group(2), fm.group(3), fm.group(4)
            for item in rest.split(","):
                item = item.strip()
                ptr = bool(star)
                if item.startswith("*"):
                    ptr, item = True, item[1:].strip()
                am = re.match(r"^(\w+)\s*\[\s*(\w+)\s*\]$", item)
                fields.append(dict(name=am.group(1) if am else item, ctype=ctype, ptr=ptr, const=bool(const), array=int(am.group(2)) if am else None))
        structs.append((m.group(1), fields))
    body = text[text.index("typedef struct rxr_ctx rxr_ctx;"):]
    for m in re.finditer(r"^([\w\s\*]+?)\b(rxr_\w+)\s*\(([^;{}]*?)\)\s*;", body, flags=re.M | re.S):
        ret = " ".join(m.group(1).split())
        if ret.startswith("typedef"):
            continue
        args = " ".join(m.group(3).split())
        funcs.append((ret, m.group(2), args))
    return defines, enums, structs, funcs


class Layout:
    def __init__(self, structs):
        self.known = {}
        self.structs = structs
        for name, fields in structs:
            self.known[name] = self.compute(fields)

    def field_size_align(self, f):
        if f["ptr"]:
            size, align = 8, 8
        elif f["ctype"] in SCALARS:
            _, size, align = SCALARS[f["ctype"]]
        else:
            st = self.known[f["ctype"]]
            size, align = st["size"], st["align"]
        if f["array"] is not None:
            size *= f["array"]
        return size, align

    def compute(self, fields):
        off, max_align, out = 0, 1, []
        for f in fields:
            size, align = self.field_size_align(f)
            off = (off + align - 1) // align * align
            out.append((f["name"], off))
            off += size
            max_align = max(max_align, align)
        total = (off + max_align - 1) // max_align * max_align
        return dict(size=total, align=max_align, offsets=out)


def rust_type(f):
    base = SCALARS[f["ctype"]][0] if f["ctype"] in SCALARS else f["ctype"]
    if f["ptr"]:
        base = ("*const " if f["const"] else "*mut ") + base
    if f["array"] is not None:
        base = f"[{base}; {f['array']}]"
    return base


def rust_arg(arg):
    """one C parameter -> (name, rust type)"""
    arg = arg.strip()
    m = re.match(r"^(const\s+)?(\w+)\s*(\*{0,2})\s*(\w+)\s*(\[\s*\w*\s*\])?$", arg)
    const, ctype, stars, name, arr = m.group(1), m.group(2), m.group(3), m.group(4), m.group(5)
    base = SCALARS[ctype][0] if ctype in SCALARS else ctype
    if arr:  # `T x[N]` in a parameter list is a pointer
        return name, ("*const " if const else "*mut ") + base
    if stars == "**":
        return name, "*mut *mut " + base
    if stars == "*":
        return name, ("*const " if const else "*mut ") + base
    return name, base


def rust_ret(ret):
    ret = ret.strip()
    if ret == "void":
        return ""
    m = re.match(r"^(const\s+)?(\w+)\s*(\*?)$", ret)
    base = SCALARS[m.group(2)][0] if m.group(2) in SCALARS else m.group(2)
    if m.group(3):
        base = ("*const " if m.group(1) else "*mut ") + base
    return " -> " + base


def define_value(val):
    v = val.strip()
    m = re.match(r"^\(?\s*(\d+)u?\s*<<\s*(\d+)\s*\)?$", v)
    if m:
        return f"{m.group(1)} << {m.group(2)}"
    m = re.match(r"^(\d+)u?$", v)
    return m.group(1) if m else None


def gen_rust(defines, enums, structs, funcs, layout):
    o = []
    o.append("//! Raw bindings of include/rxr.h -- GENERATED by tools/gen_ffi.py from the C header; do not edit.")
    o.append("//! Constants, `#[repr(C)]` mirrors of every struct, the `extern \"C\"` block, and compile-time asserts of every")
    o.append("//! struct size and field offset (the same numbers tests/abi_host.c asserts with the C compiler).")
    o.append("#![allow(non_camel_case_types, dead_code)]")
    o.append("use core::mem::{offset_of, size_of};")
    o.append("use std::os::raw::{c_char, c_int, c_void};")
    o.append("")
    for name, val in defines:
        v = define_value(val)
        if v is not None:
            o.append(f"pub const {name}: u32 = {v};")
    o.append("")
    for tname, items in enums:
        ty = "c_int" if tname == "rxr_status" else "u32"
        if tname:
            o.append(f"// enum {tname}")
        for k, v in items:
            o.append(f"pub const {k}: {ty} = {v};")
        o.append("")
    o.append("#[repr(C)]\npub struct rxr_ctx {\n    _private: [u8; 0],\n}\n")
    for name, fields in structs:
        o.append("#[repr(C)]\n#[derive(Clone, Copy)]")
        o.append(f"pub struct {name} {{")
        for f in fields:
            o.append(f"    pub {f['name']}: {rust_type(f)},")
        o.append("}")
        st = layout.known[name]
        o.append(f"const _: () = assert!(size_of::<{name}>() == {st['size']});")
        for fname, off in st["offsets"]:
            o.append(f"const _: () = assert!(offset_of!({name}, {fname}) == {off});")
        o.append("")
    o.append('extern "C" {')
    for ret, name, args in funcs:
        params = [] if args in ("", "void") else [rust_arg(a) for a in args.split(",")]
        o.append(f"    pub fn {name}({', '.join(f'{n}: {t}' for n, t in params)}){rust_ret(ret)};")
    o.append("}")
    return "\n".join(o) + "\n"


def gen_asserts(structs, layout):
    o = ["/* GENERATED by tools/gen_ffi.py from include/rxr.h; do not edit.  The sizes and offsets the Rust mirror",
         " * (shim/rusterix-hip-shim/src/ffi.rs) asserts, checked here by the C compiler (tests/abi_host.c). */",
         "#include <stddef.h>", ""]
    for name, fields in structs:
        st = layout.known[name]
        o.append(f'_Static_assert(sizeof({name}) == {st["size"]}, "sizeof({name})");')
        o.append(f'_Static_assert(_Alignof({name}) == {st["align"]}, "alignof({name})");')
        for fname, off in st["offsets"]:
            o.append(f'_Static_assert(offsetof({name}, {fname}) == {off}, "offsetof({name}, {fname})");')
        o.append("")
    return "\n".join(o)


def main():
    defines, enums, structs, funcs = parse(open(HEADER).read())
    layout = Layout(structs)
    outputs = {FFI_RS: gen_rust(defines, enums, structs, funcs, layout), ASSERTS_H: gen_asserts(structs, layout)}
    if "--check" in sys.argv:
        stale = [p for p, text in outputs.items() if not os.path.exists(p) or open(p).read() != text]
        for p in stale:
            print("stale:", os.path.relpath(p, ROOT))
        sys.exit(1 if stale else 0)
    for p, text in outputs.items():
        open(p, "w").write(text)
        print("wrote", os.path.relpath(p, ROOT), f"({len(structs)} structs, {len(funcs)} functions)")


if __name__ == "__main__":
    main()
