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
