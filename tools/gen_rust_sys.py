#!/usr/bin/env python
"""include/hqtick.h + include/hqwire.h -> integration/hqtick_sys.rs: the `extern "C"` block a tako maintainer adds
(crates/tako/src/internal/scheduler/hqtick_sys.rs), every struct with its full field list, every constant, every function.

The header is the source of truth; this file is generated from it and tests/test_abi.py regenerates it and diffs (no Rust toolchain in this image: the
structural check is what can be done here — field names, order and types against the C declarations, whose layout tests/test_abi.py checks against a compiled probe).

  python tools/gen_rust_sys.py            # rewrite integration/hqtick_sys.rs
  python tools/gen_rust_sys.py --check    # exit 1 if the committed file differs
"""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "integration", "hqtick_sys.rs")

PRIM = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int16_t": "i16", "int32_t": "i32", "int64_t": "i64",
        "int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "char": "c_char", "float": "f32", "double": "f64", "size_t": "usize", "void": "c_void"}


def camel(name: str) -> str:
    return "".join(p.capitalize() for p in name.split("_"))


def strip_comments(src: str) -> str:
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def rust_type(ctype: str, structs) -> str:
    """`const uint64_t *` -> `*const u64`, `uint32_t **` -> `*mut *mut u32`, `hqtick_ctx *` -> `*mut HqtickCtx`"""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").strip()
    const = False
    words = [w for w in t.split() if w != "struct"]
    if words and words[0] == "const":
        const = True
        words = words[1:]
    words = [w for w in words if w != "const"]
    base = " ".join(words)
    if base in PRIM:
        r = PRIM[base]
    elif base in structs:
        r = camel(base)
    else:
        raise ValueError(f"unknown C type {ctype!r}")
    for i in range(stars):
        innermost = i == 0
        r = ("*const " if (const and innermost) else "*mut ") + r
    return r


def split_decl(decl: str):
    """`const uint64_t *task_id` -> (type, name, array_len)"""
    decl = decl.strip()
    m = re.match(r"^(.*?)(\w+)\s*(\[\s*(\w+)\s*\])?$", decl, flags=re.S)
    if not m:
        raise ValueError(decl)
    return m.group(1).strip(), m.group(2), m.group(4)


def parse(header_src: str, structs, consts):
    src = strip_comments(header_src)
    out_structs, out_fns, out_consts, out_enums = [], [], [], []
    for m in re.finditer(r"#define\s+(HQ\w+)\s+(.+)", src):
        name, val = m.group(1), m.group(2).strip()
        if name.endswith("_H"):
            continue
        out_consts.append((name, val))
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", src, flags=re.S):
        nxt = 0
        for item in [x.strip() for x in m.group(1).split(",") if x.strip()]:
            if "=" in item:
                n, v = [x.strip() for x in item.split("=", 1)]
                nxt = int(v.rstrip("uU"), 0)
            else:
                n = item
            out_enums.append((n, nxt))
            nxt += 1
    for m in re.finditer(r"typedef\s+([\w \t\*]+?)\(\s*\*\s*(\w+)\s*\)\s*\(([^;]*?)\)\s*;", src, flags=re.S):  # function-pointer types (callbacks)
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = [split_decl(a)[:2] for a in args.split(",")] if args and args != "void" else []
        structs.add(name)
        out_structs.append((name, ("fnptr", ret, params)))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", src):  # opaque handles
        structs.add(m.group(2))
        out_structs.append((m.group(2), None))
    for m in re.finditer(r"typedef\s+struct\s*(\w*)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        name = m.group(3)
        structs.add(name)
        fields = []
        for stmt in [x.strip() for x in m.group(2).split(";") if x.strip()]:
            # `uint32_t a, b;` and `const uint32_t *p, *q;` declare several fields
            first, *rest = [x.strip() for x in stmt.split(",")]
            t, n, arr = split_decl(first)
            fields.append((t, n, arr))
            base = t.replace("*", "").strip()
            for r in rest:
                stars = r.count("*")
                rn, ra = re.match(r"^\**\s*(\w+)\s*(?:\[\s*(\w+)\s*\])?$", r).groups()
                fields.append((base + " " + "*" * stars, rn, ra))
        out_structs.append((name, fields))
    for m in re.finditer(r"^([A-Za-z_][\w \t\*]*?)\b(hq(?:tick|wire)_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                t, n, _ = split_decl(a)
                params.append((t, n))
        out_fns.append((ret, name, params))
    return out_structs, out_fns, out_consts, out_enums


def const_rs(name: str, val: str):
    v = val.strip()
    table = {"UINT64_MAX": ("u64", "u64::MAX"), "UINT32_MAX": ("u32", "u32::MAX"), "INT64_MAX": ("i64", "i64::MAX")}
    if v in table:
        return f"pub const {name}: {table[v][0]} = {table[v][1]};"
    m = re.match(r"^(0x[0-9A-Fa-f]+|\d+)([uU]?)$", v)
    if m:
        return f"pub const {name}: u32 = {m.group(1)};"
    return None


def generate() -> str:
    structs, consts = set(), []
    lines = [
        "// GENERATED by tools/gen_rust_sys.py from include/hqtick.h and include/hqwire.h — do not edit; tests/test_abi.py regenerates and diffs it.",
        "//",
        "// The `extern \"C\"` binding of libhqtick.so for tako (crates/tako/src/internal/scheduler/hqtick_sys.rs): every struct of the C ABI with its full field",
        "// list in declaration order (#[repr(C)]), every constant, every exported function.  What each field means is documented in the headers, next to the",
        "// reference line it mirrors.  build.rs: cargo:rustc-link-search=native=$HQTICK_LIB_DIR, cargo:rustc-link-lib=dylib=hqtick.",
        "#![allow(non_camel_case_types, dead_code)]",
        "use std::os::raw::{c_char, c_int, c_uint, c_void};",
        "",
    ]
    for header in ("hqtick.h", "hqwire.h"):
        src = open(os.path.join(ROOT, "include", header)).read()
        st, fns, cs, ens = parse(src, structs, consts)
        lines.append(f"// ---------------------------------------------------------------- include/{header}")
        for n, v in cs:
            r = const_rs(n, v)
            if r:
                lines.append(r)
        for n, v in ens:
            lines.append(f"pub const {n}: i32 = {v};")
        lines.append("")
        for name, fields in st:
            if isinstance(fields, tuple) and fields[0] == "fnptr":
                _, ret, params = fields
                ps = ", ".join(f"{n}: {rust_type(t, structs)}" for t, n in params)
                rr = "" if ret == "void" else f" -> {rust_type(ret, structs)}"
                lines.append(f"pub type {camel(name)} = Option<unsafe extern \"C\" fn({ps}){rr}>;  // {name}: NULL = none")
                lines.append("")
                continue
            if fields is None:
                lines.append(f"#[repr(C)] pub struct {camel(name)} {{ _opaque: [u8; 0] }}  // {name}: opaque handle")
                lines.append("")
                continue
            lines.append(f"#[repr(C)] #[derive(Clone, Copy)]")
            lines.append(f"pub struct {camel(name)} {{  // {name}")
            for t, n, arr in fields:
                rt = rust_type(t, structs)
                if arr:
                    rt = f"[{rt}; {arr}]"
                lines.append(f"    pub {n}: {rt},")
            lines.append("}")
            lines.append("")
        lines.append('#[link(name = "hqtick")]')
        lines.append('extern "C" {')
        for ret, name, params in fns:
            ps = ", ".join(f"{'r#' + n if n in ('type', 'in', 'ref', 'fn') else n}: {rust_type(t, structs)}" for t, n in params)
            rr = "" if ret == "void" else f" -> {rust_type(ret, structs)}"
            lines.append(f"    pub fn {name}({ps}){rr};")
        lines.append("}")
        lines.append("")
    return "\n".join(lines)


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(OUT, len(text.splitlines()), "lines")
