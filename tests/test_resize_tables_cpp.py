"""Host-only check of the MFMA coefficient-table builders (plain, vertical, whole-line vertical and band layouts) against the
scalar fixed-point table they are derived from: compiled with g++ from csrc/resize_tables.cpp alone, no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mfma_table_layouts_hold_the_scalar_coefficients():
    out_dir = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "resize_tables")
    src = [os.path.join(ROOT, "tests", "cpp", "resize_tables_main.cpp"),
           os.path.join(ROOT, "vid_dup_finder_lib_amd", "csrc", "resize_tables.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe] + src)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "resize tables ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_stream_kernel_geometry_fits_its_buffers_for_every_width():
    """csrc/resize_dispatch.cpp: LDS pitch, blocks per chunk and kernel class for every width 1..4200 (stream, K-split and
    cropped forms) against the kernels' buffer sizes, the alignment / bank-conflict rules and the sizes DESIGN.md names."""
    out_dir = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "resize_dispatch")
    src = [os.path.join(ROOT, "tests", "cpp", "resize_dispatch_main.cpp"),
           os.path.join(ROOT, "vid_dup_finder_lib_amd", "csrc", "resize_dispatch.cpp"),
           os.path.join(ROOT, "vid_dup_finder_lib_amd", "csrc", "resize_tables.cpp")]  # band tables against the per-wave kernels' table arrays
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe] + src)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "resize dispatch ok" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
