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
