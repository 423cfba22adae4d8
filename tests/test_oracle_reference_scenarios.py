"""The reference's own tests for the path, restated against the CPU oracle.  This is what pins the oracle's
search/Hamming behaviour (the reference cannot be built here: Rust, no cargo).

  vid_dup_finder_lib/tests/test_find_all.rs:134-169   one known group of 50
  vid_dup_finder_lib/tests/test_find_all.rs:171-238   discriminate by duration (50 s vs 250 s)
  vid_dup_finder_lib/tests/test_find_all.rs:240-269   discriminate by distance (2 clusters)
  vid_dup_finder_lib/tests/test_find_all.rs:271-315   search_with_references
  vid_dup_finder_lib/src/video_hashing/video_hash.rs:325-371   metric axioms
  vid_dup_finder_lib/src/video_hashing/search_algorithm.rs:203-208   empty search
"""
import numpy as np

import hashgen as hg
from oracle import vdf_oracle as orc

SCALE = 1000.0  # TOLERANCE_SCALING_FACTOR


def _search(hashes, durations, tol):
    paths = [f"p{i}" for i in range(len(hashes))]
    return orc.search(np.stack(hashes) if len(hashes) else np.zeros((0, 16), np.uint64), durations, paths, tol)


def test_find_dups_finds_a_known_group():
    rng = np.random.default_rng(1)
    groups = hg.HashesWithDistanceSet(1, 50, 201, 100, rng)
    members = groups.all_members(rng)
    dups = _search(members, [0] * len(members), 200 / SCALE)
    assert len(dups) == 1
    assert len(dups[0]) == 50


def test_find_dups_discriminates_by_duration():
    rng = np.random.default_rng(2)
    groups = hg.HashesWithDistanceSet(1, 100, 201, 100, rng)
    short = groups.groups[0].members(rng)
    hashes = list(short) + list(short[:50])
    durs = [50] * 100 + [250] * 50
    perm = rng.permutation(150)
    dups = _search([hashes[i] for i in perm], [durs[i] for i in perm], 200 / SCALE)
    dups.sort(key=len)
    assert [len(d) for d in dups] == [50, 100]


def test_find_dups_discriminates_by_distance():
    rng = np.random.default_rng(3)
    groups = hg.HashesWithDistanceSet(2, 100, 150, 50, rng)
    allh = groups.all_members(rng)
    dups = _search(allh, [0] * len(allh), 100 / SCALE)
    dups.sort(key=len)
    assert [len(d) for d in dups] == [100, 110]


def test_find_with_refs():
    rng = np.random.default_rng(4)
    groups = hg.HashesWithDistanceSet(5, 100, 150, 50, rng)
    start3 = groups.groups[3].start_hash
    cands = groups.all_members(rng)
    assert len(cands) == 100 + 110 + 120 + 130 + 140
    paths = [f"c{i}" for i in range(len(cands))]
    zeros = [0] * len(cands)
    dups = orc.search_with_references(np.stack([start3]), [0], ["ref3"], np.stack(cands), zeros, paths, 50 / SCALE)
    assert len(dups) == 1 and len(dups[0][1]) == 130 and dups[0][0] == "ref3"
    starts = np.stack([groups.groups[0].start_hash, groups.groups[4].start_hash])
    dups2 = orc.search_with_references(starts, [0, 0], ["r0", "r4"], np.stack(cands), zeros, paths, 50 / SCALE)
    assert [len(d[1]) for d in dups2] == [100, 140]  # reference input order
    assert [d[0] for d in dups2] == ["r0", "r4"]


def test_triangle_inequality_and_symmetry():
    rng = np.random.default_rng(1)
    for _ in range(300):
        a, b, c = hg.random_hash(rng), hg.random_hash(rng), hg.random_hash(rng)
        assert orc.hamming(a, b) <= orc.hamming(a, c) + orc.hamming(b, c)
        assert orc.hamming(a, b) == orc.hamming(b, a) == hg.hamming(a, b)


def test_distance_between_equal_extremes_is_zero():
    e = np.zeros(16, np.uint64)
    f = np.full(16, np.uint64(0xFFFFFFFFFFFFFFFF))
    assert orc.hamming(e, e) == 0 and orc.hamming(f, f) == 0
    assert orc.hamming(e, f) == 1024  # padding bits count (video_hash.rs:311-317)


def test_searching_nothing_returns_empty():
    assert _search([], [], 1.0) == []


def test_tolerance_int_truncates_like_rust_cast():
    for d in range(0, 1001):
        assert orc.tolerance_int(d / 1000.0) == d
    assert orc.tolerance_int(0.35) == 350 and orc.tolerance_int(0.3) == 300
    assert orc.tolerance_int(-1.0) == 0 and orc.tolerance_int(float("nan")) == 0
    assert orc.tolerance_int(1e12) == 0xFFFFFFFF


def test_rust_path_ordering_is_componentwise():
    k = orc.rust_path_key
    assert k("a/b") < k("a.b")  # bytewise '.' < '/', but component "a" < "a.b"
    assert k("a//b/") == k("a/b") == k("a/./b")
    assert k("/a") < k("a") and k("./a") < k("../a") < k("a")
    order = orc.sort_order([5, 5, 1, 5], ["b", "a/b", "z", "a.b"])
    assert order == [2, 1, 3, 0]
