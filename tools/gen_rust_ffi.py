#!/usr/bin/env python3
"""Generates integration/rust/src/ffi.rs — the raw `extern "C"` block a maintainer of the reference crate adds — from
include/nvr.h, declaration by declaration (structs -> #[repr(C)], opaque handles, status / state constants, every NVR_API
function).  The build image has no Rust toolchain, so the output is NOT compiled here; tests/test_host_parity.py checks that
it is up to date with the header and names every symbol libnvr.so exports.

    python tools/gen_rust_ffi.py            # rewrite integration/rust/src/ffi.rs
    python tools/gen_rust_ffi.py --check    # exit 1 if the committed file is stale
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "nvr.h")
OUT = os.path.join(ROOT, "integration", "rust", "src", "ffi.rs")

SCALARS = {"int": "c_int", "unsigned": "u32", "unsigned int": "u32", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64",
           "size_t": "usize", "float": "f32", "double": "f64", "char": "c_char", "void": "c_void", "uint8_t": "u8", "uint16_t": "u16",
           "nvr_half": "u16"}


def strip_comments(s: str) -> str:
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def rust_type(ctype: str, structs, opaques) -> str:
    """C declarator type (without the name) -> Rust type."""
    t = " ".join(ctype.replace("*", " * ").split())
    parts = t.split(" ")
    # walk right to left over '*' and 'const'
    base, ptrs = [], []
    i = 0
    toks = parts
    # base type = leading tokens up to the first '*'
    while i < len(toks) and toks[i] != "*":
        base.append(toks[i]); i += 1
    base_const = "const" in base
    base_name = " ".join(x for x in base if x not in ("const", "struct"))
    rest = toks[i:]
    # each '*' optionally followed by 'const' (pointer itself const: irrelevant to the pointee's mutability)
    levels = []
    j = 0
    while j < len(rest):
        assert rest[j] == "*", ctype
        const_ptr = j + 1 < len(rest) and rest[j + 1] == "const"
        levels.append(const_ptr)
        j += 2 if const_ptr else 1
    if base_name in SCALARS:
        r = SCALARS[base_name]
    elif base_name in structs or base_name in opaques:
        r = base_name
    elif base_name == "nvr_stream_fn":
        r = "nvr_stream_fn"
    else:
        raise SystemExit(f"unknown C type '{base_name}' in '{ctype}'")
    # innermost pointee constness comes from the base; outer levels from the '* const' of the level below
    pointee_const = base_const
    for const_ptr in levels:
        r = ("*const " if pointee_const else "*mut ") + r
        pointee_const = const_ptr
    return r


def split_decl(decl: str):
    """'const int64_t *prompt' -> ('const int64_t *', 'prompt', array_len or None)"""
    decl = decl.strip()
    m = re.match(r"^(.*?)([A-Za-z_]\w*)\s*(\[\s*(\d+)\s*\])?$", decl)
    assert m, decl
    ctype, name, arr = m.group(1).strip(), m.group(2), m.group(4)
    if not ctype:                                   # unnamed parameter: the "name" was the type
        return decl, None, None
    return ctype, name, int(arr) if arr else None


def main() -> None:
    raw = strip_comments(open(HDR).read())
    src = "\n".join(l for l in raw.split("\n") if not l.lstrip().startswith("#"))       # declarations only
    structs, opaques, out_structs, consts, funcs = {}, [], [], [], []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", src):
        opaques.append(m.group(2))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        structs[m.group(3)] = m.group(2)
    for name, body in structs.items():
        fields = []
        for stmt in body.split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            # 'int32_t a, b' style lists share the base type
            first, *more = [x.strip() for x in stmt.split(",")]
            ctype, fname, arr = split_decl(first)
            base = ctype.replace("*", "").strip()
            for decl in [first] + [f"{base} {x}" if "*" not in x else f"{base} {x}" for x in more]:
                ct, fn, ar = split_decl(decl)
                rt = rust_type(ct, structs, opaques)
                fields.append(f"pub {fn}: " + (f"[{rt}; {ar}]" if ar else rt))
        out_structs.append(f"#[repr(C)]\n#[derive(Clone, Copy)]\npub struct {name} {{\n    " + ",\n    ".join(fields) + ",\n}")
    for m in re.finditer(r"enum\s*(\w*)\s*\{(.*?)\}", src, flags=re.S):
        for item in m.group(2).split(","):
            item = item.strip()
            if "=" in item:
                k, v = [x.strip() for x in item.split("=")]
                consts.append(f"pub const {k}: i32 = {v};")
    for m in re.finditer(r"#define\s+(NVR_[A-Z_]+)\s+(\d+)\b", raw):
        consts.append(f"pub const {m.group(1)}: usize = {m.group(2)};")
    for m in re.finditer(r"NVR_API\s+(.*?)\b(nvr_\w+)\s*\((.*?)\)\s*;", src, flags=re.S):
        ret, name, params = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        args = []
        if params and params != "void":
            for k, prm in enumerate(params.split(",")):
                ctype, pname, arr = split_decl(prm)
                if arr is not None:                 # 'uint8_t id[128]' decays to a pointer
                    ctype = ctype + " *"
                rt = rust_type(ctype, structs, opaques)
                pname = pname or f"arg{k}"
                if pname in ("type", "fn", "in", "ref", "move", "self", "use", "mod", "box", "match", "loop"):
                    pname += "_"
                args.append(f"{pname}: {rt}")
        rr = "" if ret == "void" else " -> " + rust_type(ret, structs, opaques)
        funcs.append(f"    pub fn {name}({', '.join(args)}){rr};")
    text = ("// GENERATED by tools/gen_rust_ffi.py from include/nvr.h — do not edit.  Raw declarations of libnvr.so for the reference crate\n"
            "// (src/ffi.rs there).  Not compiled in the build image (no Rust toolchain); kept in sync with the header by\n"
            "// tests/test_host_parity.py::test_rust_ffi_is_in_sync_with_the_header.\n"
            "#![allow(non_camel_case_types, dead_code)]\n"
            "use std::os::raw::{c_char, c_int, c_void};\n\n"
            + "\n".join(f"#[repr(C)]\npub struct {o} {{\n    _private: [u8; 0],\n}}" for o in opaques) + "\n\n"
            + "\n".join(consts) + "\n\n"
            + "/// `typedef int (*nvr_stream_fn)(const nvr_sequence_output *out, void *user);` (non-zero return = receiver dropped)\n"
            "pub type nvr_stream_fn = Option<unsafe extern \"C\" fn(out: *const nvr_sequence_output, user: *mut c_void) -> c_int>;\n\n"
            + "\n\n".join(out_structs) + "\n\n"
            + "#[link(name = \"nvr\")]\nextern \"C\" {\n" + "\n".join(funcs) + "\n}\n")
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("integration/rust/src/ffi.rs is stale: run python tools/gen_rust_ffi.py", file=sys.stderr)
            raise SystemExit(1)
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {len(funcs)} functions, {len(out_structs)} structs, {len(opaques)} opaque handles")


if __name__ == "__main__":
    main()
