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


def _build_and_run_mt(tmp_path, flags, name):
    exe = str(tmp_path / name)
    csrc = os.path.join(ROOT, "vid_dup_finder_lib_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", *flags, "-pthread", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "host_sanitize_mt_main.cpp"),
                           os.path.join(csrc, "cache_format.cpp"), os.path.join(csrc, "cache_metadata.cpp"),
                           os.path.join(csrc, "path_order.cpp"), os.path.join(csrc, "host_sort.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sanitize mt ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_mt_decoder_sidecar_parser_and_path_ranker_are_clean_under_asan_ubsan(tmp_path):
    """The multi-threaded cache decoder (valid files on 2..16 threads = the one-thread result; damaged files: the same verdict),
    the sidecar parser on hostile text without a terminator, the path ranker on 1..8 threads against vdf_path_compare."""
    _build_and_run_mt(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], "host_sanitize_mt")


def test_mt_decoder_and_path_ranker_are_clean_under_tsan(tmp_path):
    """The same driver under ThreadSanitizer: the decoder's ranges and the ranker's sample sort share arrays between threads."""
    _build_and_run_mt(tmp_path, ["-fsanitize=thread"], "host_sanitize_mt_tsan")


def test_batching_queue_logic_is_clean_under_tsan(tmp_path):
    """csrc/hash_queue.cpp itself (one mutex, a condition variable per kind of wait - round 6) with the GPU behind it replaced by
    stand-ins (tests/cpp/queue_tsan_main.cpp): 48 callers against batches of 4, one to four slots, batches that never fill, a
    single caller.  A lost wake-up is a hang (the timeout), a wrong hand-over a wrong checksum, an unlocked access a TSan report."""
    import pytest

    if not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("HIP headers not installed")
    exe = str(tmp_path / "queue_tsan")
    csrc = os.path.join(ROOT, "vid_dup_finder_lib_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-DVDF_QUEUE_SYSTEM_CLOCK", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-o", exe, os.path.join(ROOT, "tests", "cpp", "queue_tsan_main.cpp"),
                           os.path.join(csrc, "hash_queue.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "queue tsan ok" in out.stdout and "ThreadSanitizer" not in out.stderr, (out.stdout[-1500:], out.stderr[-3000:])
