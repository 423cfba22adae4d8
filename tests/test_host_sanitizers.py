"""Host-only library code under AddressSanitizer + UBSan (CPU build; GPU sanitizers are not available on the pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replay_tables_and_sorts_are_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    csrc = os.path.join(ROOT, "vid_dup_finder_lib_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_sanitize_main.cpp"),
                           os.path.join(csrc, "replay.cpp"), os.path.join(csrc, "resize_tables.cpp"),
                           os.path.join(csrc, "host_sort.cpp"), os.path.join(csrc, "cache_format.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "sanitize ok" in out.stdout, out.stderr[-3000:]
