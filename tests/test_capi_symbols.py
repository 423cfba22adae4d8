"""The C-ABI library loads on a CPU-only box and exports every symbol include/vdf.h declares; the host-only
entry points (no GPU needed) behave like the oracle.  No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import hashgen as hg
from oracle import vdf_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vdf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vdf_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    from vid_dup_finder_lib_amd import _capi

    lib = _capi.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vdf.h but not exported by libvdf_hip.so"
    assert sorted(_capi.SIGNATURES) == names, "ctypes binding and header disagree"


# ---- the Rust side of the boundary: rust/vdf-sys/src/lib.rs must say what include/vdf.h says ------------------------------------------
_C_CLASS = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "size_t": "usize",
            "double": "f64", "float": "f32", "long long": "i64", "unsigned long long": "u64", "char": "i8", "void": "void"}
_RS_CLASS = {"c_int": "i32", "i32": "i32", "u32": "u32", "u64": "u64", "i64": "i64", "u8": "u8", "usize": "usize", "f64": "f64", "f32": "f32",
             "c_longlong": "i64", "c_ulonglong": "u64", "c_char": "i8", "c_void": "void"}


def _c_shape(ctype):
    """('ptr' depth, scalar class or struct name) of a C type without its parameter name."""
    depth = ctype.count("*")
    base = " ".join(t for t in ctype.replace("*", " ").split() if t != "const")
    return depth, _C_CLASS.get(base, base)


def _rs_shape(rtype):
    depth = rtype.count("*const") + rtype.count("*mut")
    base = re.sub(r"\*(const|mut)\s*", "", rtype).strip()
    return depth, _RS_CLASS.get(base, base)


def _header_protos():
    text = re.sub(r"//.*", "", re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "vdf.h")).read(), flags=re.S))
    out = {}
    for ret, name, args in re.findall(r"^([A-Za-z_][\w\s\*]*?)\b(vdf_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.M | re.S):
        args = " ".join(args.split())
        params = [] if args in ("", "void") else [_c_shape(re.match(r"(.*?)(\w+)$", a.strip()).group(1)) for a in args.split(",")]
        out[name] = (params, _c_shape(ret.strip()))
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for line in m.group(2).split(";"):
            line = " ".join(line.split())
            if not line:
                continue
            if "(*" in line:
                fields.append((re.search(r"\(\s*\*\s*(\w+)", line).group(1), "fnptr"))
            else:
                mm = re.match(r"(.*?)(\w+)$", line)
                fields.append((mm.group(2), _c_shape(mm.group(1).strip())))
        structs[m.group(3)] = fields
    return out, structs


def _rust_protos():
    text = re.sub(r"//.*", "", open(os.path.join(ROOT, "rust", "vdf-sys", "src", "lib.rs")).read())
    out = {}
    for name, args, ret in re.findall(r"pub fn (vdf_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", text, flags=re.S):
        args = " ".join(args.split()).rstrip(",")
        params = [] if not args else [_rs_shape(a.split(":", 1)[1].strip()) for a in args.split(", ")]
        out[name] = (params, _rs_shape(ret.strip()) if ret else (0, "void"))
    structs = {}
    for name, body in re.findall(r"#\[repr\(C\)\][^{]*?pub struct (\w+)\s*\{(.*?)\n\}", text, flags=re.S):
        fields = []
        for line in body.split("\n"):
            line = line.strip().rstrip(",")
            if not line.startswith("pub "):
                continue
            f, ty = line[4:].split(":", 1)
            fields.append((f.strip(), "fnptr" if "extern" in ty else _rs_shape(ty.strip())))
        structs[name] = fields
    return out, structs


def test_rust_sys_crate_declares_the_whole_header():
    """rust/vdf-sys/src/lib.rs (what a maintainer's Rust build links against; no rustc in this image) against include/vdf.h: the same
    functions, the same number of parameters, every parameter and return value the same scalar width / pointer depth / struct, every
    #[repr(C)] struct field for field - and the committed file is what tools/gen_vdf_sys.py makes of the header."""
    import subprocess
    import sys

    c_fns, c_structs = _header_protos()
    r_fns, r_structs = _rust_protos()
    assert sorted(c_fns) == sorted(r_fns) == _declared()
    assert len(c_fns) >= 64
    for name in c_fns:
        assert c_fns[name] == r_fns[name], (name, c_fns[name], r_fns[name])
    for name, fields in c_structs.items():
        assert r_structs.get(name) == fields, (name, fields, r_structs.get(name))
    assert {"vdf_cache_soa", "vdf_cache_metadata", "vdf_cache_search_timing", "vdf_search_timing", "vdf_shard_exchange"} <= set(r_structs)
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_vdf_sys.py"), "--check"]).returncode == 0


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    import vid_dup_finder_lib_amd as vdf

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vdf.VdfError) as ei:
        vdf.Engine(0)
    assert ei.value.code == -3  # VDF_E_HIP


def test_multi_context_fails_loudly_without_gpu_and_rejects_bad_lists():
    import ctypes as C

    import torch

    import vid_dup_finder_lib_amd as vdf
    from vid_dup_finder_lib_amd import _capi

    lib = _capi.load()
    ctx = C.c_void_p()
    assert lib.vdf_ctx_create_multi(None, 0, C.byref(ctx)) == _capi.VDF_E_INVAL and not ctx.value
    arr = (C.c_int * 2)(0, 0)
    assert lib.vdf_ctx_create_multi(arr, 0, C.byref(ctx)) == _capi.VDF_E_INVAL
    assert lib.vdf_ctx_device_count(None) == 0 and lib.vdf_ctx_device_at(None, 0) == -1
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vdf.VdfError) as ei:
        vdf.Engine(devices=[0, 0])
    assert ei.value.code == -3 and "device list entry 0" in str(ei.value)  # VDF_E_HIP, with the slot that failed


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vid_dup_finder_lib_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "vdf_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_host_helpers_match_oracle():
    from vid_dup_finder_lib_amd import engine as ve

    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b = hg.random_hashes(rng, 2)
        assert ve.hamming_distance_words(a, b) == orc.hamming(a, b)
    for d in range(0, 1001, 7):
        assert ve.tolerance_int(d / 1000.0) == d
    assert ve.tolerance_int(0.35) == 350 and ve.tolerance_int(float("nan")) == 0 and ve.tolerance_int(1e30) == 2**32 - 1
    dur = np.sort(np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=5000))).astype(np.uint32))
    assert ve.count_pairs_self(dur) == orc.pairs_self(dur)
    assert ve.count_pairs_self(np.zeros(1000, np.uint32)) == 1000 * 999 // 2
    refs = rng.integers(0, 8000, size=300).astype(np.uint32)
    want = 0
    for r in refs:
        lo = np.searchsorted(dur, np.uint32(int(float(r) * 0.95)), side="left")
        hi = np.searchsorted(dur, np.uint32(int(float(r) * 1.05)), side="right")
        want += max(0, hi - lo)
    assert ve.count_pairs_refs(dur, refs) == want


def test_a_c99_caller_compiles_against_the_header_and_runs(tmp_path):
    """The boundary is a C ABI: the header must be C (cgo, bindgen and ctypes read it as C, not C++).  tests/cpp/c_caller.c is built with
    -std=c99 -pedantic -Werror against libvdf_hip.so and exercises the host-only entry points (and vdf_ctx_create's loud refusal
    without a GPU)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "vid_dup_finder_lib_amd")
    exe = str(tmp_path / "c_caller")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(root, "tests", "cpp", "c_caller.c"),
                           "-L" + lib, "-lvdf_hip", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "c caller ok" in out.stdout, out.stderr[-2000:]


def test_informational_getters_answer_without_a_gpu():
    from vid_dup_finder_lib_amd import _capi

    lib = _capi.load()
    ver = lib.vdf_version()
    assert isinstance(ver, bytes) and ver and all(32 <= c < 127 for c in ver)
    assert lib.vdf_row_tile_size() in (256, 512, 1024)  # rows dealt to a shard at a time (vdf.h: vdf_search_self_device)
    assert lib.vdf_live_device_bytes() == 0 and lib.vdf_live_pinned_bytes() == 0  # nothing is held before a context exists
    assert lib.vdf_ctx_device(None) == -1 and lib.vdf_ctx_device_count(None) <= 0 and lib.vdf_ctx_rccl_ranks(None) <= 0


def _adjacency(words, dur, tol, lo_fn):
    """All thresholded pairs inside the duration windows, by numpy (test-side stand-in for the kernel)."""
    n = len(dur)
    bits = np.unpackbits(words.view(np.uint8), axis=1).astype(np.int16)
    hits = []
    for i in range(n):
        lo, hi = lo_fn(i)
        if hi <= lo:
            continue
        d = (bits[lo:hi] != bits[i]).sum(axis=1)
        for j in np.nonzero(d <= tol)[0]:
            hits.append((i, lo + j))
    return np.array(hits, np.uint32).reshape(-1, 2)


@pytest.mark.parametrize("durations", ["zero", "windowed"])
def test_host_replay_reproduces_search_self(durations):
    """The design basis: thresholded adjacency (any order) + sequential host replay == literal search_self."""
    from vid_dup_finder_lib_amd import engine as ve

    rng = np.random.default_rng(5)
    words, dur = hg.planted_set(rng, 600, n_clusters=25, max_copies=8, durations=durations)
    w, d, _ = hg.sort_by_duration(words, dur)
    for tol in (350, 60):
        def win(i):
            thresh = min(int(float(d[i]) * 1.1), 2**32 - 1)
            return i + 1, int(np.searchsorted(d, np.uint32(thresh), side="right"))
        hits = _adjacency(w, d, tol, win)
        got = ve.finish_self(ve.replay_self(len(d), hits))
        assert got == orc.search_self_sorted(w, d, tol)
        # partial replays with carried consumption state (the overflow protocol) give the same groups
        matched = np.zeros(len(d), np.uint8)
        g = None
        for a, b in ((0, 100), (100, 101), (101, 450), (450, 600)):
            g = ve.replay_self(len(d), hits, matched, a, b, g)
        assert ve.finish_self(g) == got


def test_groups_from_ref_hits_and_empty():
    from vid_dup_finder_lib_amd import engine as ve

    hits = np.array([[2, 5], [2, 9], [7, 1]], np.uint32)
    assert ve.groups_from_ref_hits(hits) == [(2, [5, 9]), (7, [1])]
    assert ve.groups_from_ref_hits(np.zeros((0, 2), np.uint32)) == []
    assert ve.finish_self(ve.replay_self(10, np.zeros((0, 2), np.uint32))) == []


def test_host_hit_sort_matches_lexsort():
    """vdf_sort_hits (host radix sort on row || col) on both sides of its small-input cut-over, with duplicates and extreme values."""
    from vid_dup_finder_lib_amd import engine as ve

    rng = np.random.default_rng(0)
    for n in (0, 1, 5, 4095, 4096, 100_000):
        h = np.stack([rng.integers(0, 50_000, size=n), rng.integers(0, 2**32, size=n)], axis=1).astype(np.uint32).reshape(-1, 2)
        if n > 4:
            h[0] = (0xFFFFFFFF, 0xFFFFFFFF)
            h[1] = (0, 0)
            h[2] = h[3]
        got = ve.sort_hits(h.copy())
        want = h[np.lexsort((h[:, 1], h[:, 0]))] if n else h
        assert np.array_equal(got, want), n
