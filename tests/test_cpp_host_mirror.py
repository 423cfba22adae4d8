"""The C++ host mirror (vid_dup_finder_lib_amd/host/vdf.hpp) over the C ABI: compiled with g++ against
libvdf_hip.so.  CPU: its host-only selftest.  GPU: vdf::search / vdf::search_with_references vs the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import hashgen as hg
from oracle import vdf_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "host_mirror")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "host_mirror_main.cpp")
    hdr = os.path.join(ROOT, "vid_dup_finder_lib_amd", "host", "vdf.hpp")
    lib = os.path.join(ROOT, "vid_dup_finder_lib_amd")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", EXE, src, "-L" + lib, "-lvdf_hip",
                               "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    return EXE


def _write(path, words, dur, paths):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(dur)))
        for i in range(len(dur)):
            p = paths[i].encode()
            f.write(words[i].tobytes() + struct.pack("<II", int(dur[i]), len(p)) + p)


def _parse(out):
    return [line.split("\t") for line in out.strip().splitlines() if line]


def test_cpp_selftest_runs_without_gpu():
    exe = _build()
    out = subprocess.run([exe, "selftest"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "selftest ok" in out.stdout, out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0,0"])
def test_cpp_search_matches_oracle(tmp_path, devices, monkeypatch):
    """devices = "0,0": the compiled C++ caller goes through ONE multi-GPU context (two slots on GPU 0), as a Rust caller of
    search() would with VDF_DEVICES set."""
    if devices:
        monkeypatch.setenv("VDF_DEVICES", devices)
    else:
        monkeypatch.delenv("VDF_DEVICES", raising=False)
    exe = _build()
    rng = np.random.default_rng(42)
    words, dur = hg.planted_set(rng, 1500, n_clusters=40, durations="windowed")
    paths = [f"d{int(rng.integers(0, 4))}/v{i}.mkv" if i % 5 else f"d.{i}/v.mkv" for i in range(len(dur))]
    f = tmp_path / "db.bin"
    _write(f, words, dur, paths)
    out = subprocess.run([exe, "search", str(f), "0.35"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = [g[1:] for g in _parse(out.stdout)]
    assert got == orc.search(words, dur, paths, 0.35) and len(got) > 0
    out = subprocess.run([exe, "refs", str(f), "60", "0.2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    want = orc.search_with_references(words[:60], dur[:60], paths[:60], words[60:], dur[60:], paths[60:], 0.2)
    assert [(g[0], g[1:]) for g in _parse(out.stdout)] == [(r, m) for r, m in want]


@pytest.mark.gpu
def test_cpp_search_disk_on_a_cache_file(tmp_path):
    """The app's side from compiled C++ (host/vdf.hpp: Cache, search_cache - what rust/vdf-app/src/search_disk.rs is in Rust): the bytes of a
    cache FILE -> SoA -> filename filters as index lists -> vdf_search_cache_entries -> MatchGroups of paths, with SearchOutput::sort's
    Sorting::Distance key per group from the same call (app_fns.rs:428-482, search_output.rs:43-60).  Find-all and with-refs, against the oracle
    on the selected entries and a brute-force key."""
    from vid_dup_finder_lib_amd import cache as vc

    exe = _build()
    rng = np.random.default_rng(7)
    words, dur = hg.planted_set(rng, 1200, n_clusters=40, durations="windowed")
    paths = [("/lib/new/" if i % 3 else "/lib/ref/") + f"d{int(rng.integers(0, 4))}/v{i}.mkv" for i in range(len(dur))]
    perm = rng.permutation(len(dur))  # a cache file holds its entries in HashMap order
    words, dur, paths = words[perm], dur[perm], [paths[i] for i in perm]
    f = tmp_path / "cache.bin"
    f.write_bytes(vc.encode_cache(words, dur, paths))
    where = {p: i for i, p in enumerate(paths)}

    def key(group_paths):
        idx = [where[p] for p in group_paths]
        return max(orc.hamming(words[a], words[b]) for k, a in enumerate(idx) for b in idx[k + 1:])

    # find-all over everything under /lib
    out = subprocess.run([exe, "cache", str(f), "0.35", "/lib/", "-"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = _parse(out.stdout)
    want = orc.search(words, dur, paths, 0.35)
    assert [r[2:] for r in rows] == want and len(want) > 10 and all(r[1] == "-" for r in rows)
    assert [int(r[0]) for r in rows] == [key(g) for g in want]
    # with-refs: /lib/ref/ entries are the references, /lib/new/ the candidates
    out = subprocess.run([exe, "cache", str(f), "0.3", "/lib/new/", "/lib/ref/"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = _parse(out.stdout)
    ri = [i for i, p in enumerate(paths) if p.startswith("/lib/ref/")]
    ci = [i for i, p in enumerate(paths) if p.startswith("/lib/new/")]
    want = orc.search_with_references(words[ri], dur[ri], [paths[i] for i in ri], words[ci], dur[ci], [paths[i] for i in ci], 0.3)
    assert [(r[1], r[2:]) for r in rows] == [(r, m) for r, m in want] and len(want) > 5
    assert [int(r[0]) for r in rows] == [key(m + [r]) for r, m in want]


@pytest.mark.gpu
def test_cpp_search_sorts_large_sets_through_the_engine():
    """vdf::search from 2048 hashes on takes Search::sort's order from the library (vdf_sort_order_paths) instead of a PathKey per entry:
    the mirror's own `bench` mode checks that both orders are the same one and prints what each costs."""
    exe = _build()
    out = subprocess.run([exe, "bench", "20000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "same order: yes" in out.stdout and " 1 groups" in out.stdout, (out.stdout, out.stderr[-1000:])
