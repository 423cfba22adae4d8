#!/usr/bin/env python3
"""Writes rust/vdf-sys/src/lib.rs from include/vdf.h: every entry point, every #[repr(C)] struct, the status codes and the constants.
    python tools/gen_vdf_sys.py            # rewrite the file
    python tools/gen_vdf_sys.py --check    # exit 1 if the committed file differs from what the header yields
The image has no rustc, so the binding cannot be compiled here; what CAN be held is that it says what the header says
(tests/test_capi_symbols.py parses both files on its own and compares names, arity and the width of every scalar).
Which seam of the crate / app each group serves: INTEGRATION.md."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vdf.h")
OUT = os.path.join(ROOT, "rust", "vdf-sys", "src", "lib.rs")

SCALARS = {"int": "c_int", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "int64_t": "i64", "size_t": "usize",
           "double": "f64", "float": "f32", "long long": "c_longlong", "unsigned long long": "c_ulonglong", "char": "c_char", "void": "c_void"}
OPAQUE = ("vdf_ctx", "vdf_hash_queue")

# one line of context per group of entry points (the header has the contracts; this is the map from the crate's side)
GROUPS = [
    ("context", r"vdf_ctx_|vdf_last_error|vdf_version|vdf_live_"),
    ("host helpers: hamming_distance (video_hash.rs:190-192,311-317), the tolerance cast (search_algorithm.rs:64,82), window counts",
     r"vdf_hamming_u1024|vdf_tolerance_int|vdf_count_pairs_|vdf_groups_free|vdf_buffer_free"),
    ("VideoHash::from_frames (video_hash.rs:45-73) and Cropdetect::Letterbox in front of it (video_hash_builder.rs:188-212)", r"vdf_hash_frames_|vdf_cropdetect_"),
    ("search() / search_with_references() (video_dup_finder.rs:7-46 over search_algorithm.rs:63-185)", r"vdf_search_self$|vdf_search_refs$"),
    ("device-resident building blocks: sharding over processes, Search::sort on the device, the host replay", r"vdf_search_(self|refs)_device|vdf_bitmap_or|vdf_sort_|vdf_apply_|vdf_row_tile|vdf_replay_|vdf_groups_finish|vdf_groups_from"),
    ("multi-GPU contexts: shards already resident on the devices", r"_shards$"),
    ("the batching queue behind VideoHashBuilder::hash called from rayon workers (video_hash_filesystem_cache.rs:237-257)", r"vdf_hash_queue_"),
    ("SearchOutput::sort's distance key (search_output.rs:43-60)", r"vdf_groups_max_distance"),
    ("the app's hash cache: wire format (base_fs_cache.rs:106-118,192-204), metadata sidecar (cache_metadata.rs; "
     "video_hash_filesystem_cache.rs:76-139), PathBuf order, cache -> MatchGroups (app_fns.rs:428-482)", r"vdf_cache_|vdf_path_|vdf_search_cache_entries"),
]


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//.*", "", text)


def rust_type(c, is_return=False):
    """C type text (without the parameter name) -> Rust."""
    c = " ".join(c.replace("*", " * ").split())
    if c == "void" and is_return:
        return ""
    toks = c.split(" ")
    stars = toks.count("*")
    # constness of every level: tokens between stars
    levels, cur = [], []
    for t in toks:
        if t == "*":
            levels.append(cur)
            cur = []
        else:
            cur.append(t)
    tail_const = "const" in cur  # `*const` after the last star binds to the pointer itself (e.g. `const T *const *p`)
    base_toks = [t for t in levels[0] if t != "const"] if stars else [t for t in cur if t != "const"]
    base = " ".join(base_toks)
    rs = SCALARS.get(base, base)
    if stars == 0:
        return rs
    # level 0 constness applies to the pointee of the innermost pointer
    consts = ["const" in levels[0]] + ["const" in lv for lv in levels[1:]] + [tail_const]
    out = rs
    for k in range(stars):
        out = ("*const " if consts[k] else "*mut ") + out
    return out


def parse(text):
    t = strip_comments(text)
    structs = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", t, flags=re.S):
        fields = []
        for line in m.group(2).split(";"):
            line = " ".join(line.split())
            if not line:
                continue
            fp = re.match(r"(\w[\w\s\*]*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", line)
            if fp:  # function pointer member
                args = [a.strip() for a in fp.group(3).split(",")]
                tys = [rust_type(re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") else a) for a in args]
                ret = rust_type(fp.group(1).strip(), True)
                fields.append((fp.group(2), "Option<unsafe extern \"C\" fn(" + ", ".join(tys) + ")" + (f" -> {ret}" if ret else "") + ">"))
                continue
            mm = re.match(r"(.*?)(\w+)$", line)
            fields.append((mm.group(2), rust_type(mm.group(1).strip())))
        structs.append((m.group(3), fields))
    enums = re.findall(r"(VDF_\w+)\s*=\s*(-?\d+)", re.search(r"typedef\s+enum\s+vdf_status\s*\{(.*?)\}", t, flags=re.S).group(1))
    defines = re.findall(r"#define\s+(VDF_\w+)\s+([\d\.]+)", t)
    consts = re.findall(r"(VDF_(?:CACHE|CROPDETECT)_\w+)\s*=\s*(-?\d+)", t)
    protos = []
    for ret, name, args in re.findall(r"^([A-Za-z_][\w\s\*]*?)\b(vdf_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", t, flags=re.M | re.S):
        args = " ".join(args.split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                params.append((mm.group(2), rust_type(mm.group(1).strip())))
        protos.append((name, params, rust_type(ret.strip(), True)))
    return structs, enums, defines, consts, protos


def render():
    structs, enums, defines, consts, protos = parse(open(HEADER).read())
    o = []
    o.append("//! Raw bindings of include/vdf.h: EVERY entry point libvdf_hip.so exports, every `#[repr(C)]` struct, the status codes.")
    o.append("//! GENERATED by tools/gen_vdf_sys.py from the header - edit the header, then run the tool; tests/test_capi_symbols.py (CPU suite of")
    o.append("//! the engine) parses this file and the header independently and compares names, arity and scalar widths.  The contracts are in the")
    o.append("//! header's comments; which seam of vid_dup_finder_lib / vid_dup_finder_app each group serves is in INTEGRATION.md.")
    o.append("#![allow(non_camel_case_types)]")
    o.append("use std::os::raw::{c_char, c_int, c_longlong, c_ulonglong, c_void};")
    o.append("")
    for name in OPAQUE:
        o += ["#[repr(C)]", f"pub struct {name} {{", "    _private: [u8; 0],", "}"]
    o.append("")
    for name, fields in structs:
        plain = all("*" not in ty and "Option<" not in ty for _, ty in fields)
        o.append("#[repr(C)]")
        o.append("#[derive(Clone, Copy, Debug" + (", Default, PartialEq" if plain else "") + ")]")
        o.append(f"pub struct {name} {{")
        for f, ty in fields:
            o.append(f"    pub {f}: {ty},")
        o.append("}")
        o.append("")
    for k, v in enums:
        o.append(f"pub const {k}: c_int = {v};")
    o.append("")
    for k, v in defines:
        if k == "VDF_H":
            continue
        o.append(f"pub const {k}: {'f64' if '.' in v else 'usize'} = {v};")
    for k, v in consts:
        o.append(f"pub const {k}: i32 = {v};")
    o.append("")
    o.append('extern "C" {')
    left = list(protos)
    for title, pat in GROUPS:
        grp = [p for p in left if re.search(pat, p[0])]
        if not grp:
            continue
        left = [p for p in left if p not in grp]
        o.append(f"    // ---- {title}")
        for name, params, ret in grp:
            sig = ", ".join(f"{'r#ref' if p == 'ref' else p}: {ty}" for p, ty in params)
            line = f"    pub fn {name}({sig})" + (f" -> {ret}" if ret else "") + ";"
            if len(line) > 150:
                line = f"    pub fn {name}(\n        " + ",\n        ".join(f"{p}: {ty}" for p, ty in params) + ",\n    )" + (f" -> {ret}" if ret else "") + ";"
            o.append(line)
    assert not left, [p[0] for p in left]
    o.append("}")
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {text.count('pub fn ')} functions")
