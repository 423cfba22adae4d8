"""Host-side mirror of the reference's public surface (no GPU): MatchGroup, Error mapping, Path ordering, sort."""
import numpy as np
import pytest

import vid_dup_finder_lib_amd as vdf
from oracle import vdf_oracle as orc


def test_match_group_contract():
    """matches/match_group.rs:21-105"""
    with pytest.raises(vdf.TooFewEntries):
        vdf.MatchGroup.new(["a"])
    with pytest.raises(vdf.TooFewEntries):
        vdf.MatchGroup.new_with_reference("r", [])
    g = vdf.MatchGroup.new(["a", "b", "c"])
    assert g.len() == 3 and g.reference() is None
    assert list(g.contained_paths()) == ["a", "b", "c"]
    assert [list(x.duplicates()) for x in g.dup_combinations()] == [["a", "b"], ["a", "c"], ["b", "c"]]
    r = vdf.MatchGroup.new_with_reference("ref", ["x", "y"])
    assert r.len() == 2 and r.reference() == "ref"
    assert list(r.contained_paths()) == ["x", "y", "ref"]  # duplicates, then the reference
    assert [(c.reference(), list(c.duplicates())) for c in r.dup_combinations()] == [("ref", ["x"]), ("ref", ["y"])]


def test_rust_path_order_matches_oracle_restatement():
    paths = ["a/b", "a.b", "a//b/", "/a", "a", "./a", "../a", "a/./b", "b", "a/b/c", "a/b.c", "", ".", "/", "a b", "A"]
    keys_p = [vdf.rust_path_key(p) for p in paths]
    keys_o = [orc.rust_path_key(p) for p in paths]
    assert keys_p == keys_o
    assert vdf.rust_path_key("a/b") < vdf.rust_path_key("a.b")
    assert vdf.rust_path_key("a//b/") == vdf.rust_path_key("a/b")
    hashes = [vdf.VideoHash(None, p, d) for p, d in [("b", 5), ("a/b", 5), ("z", 1), ("a.b", 5), ("a/b", 5)]]
    assert vdf.sort_order(hashes) == [2, 1, 4, 3, 0]  # stable: equal keys keep input order


def test_video_hash_value_semantics():
    a = vdf.VideoHash.empty_hash("p")
    f = vdf.VideoHash.full_hash("p")
    assert a.hamming_distance(a) == 0 and f.hamming_distance(f) == 0 and a.hamming_distance(f) == 1024
    assert a.normalized_hamming_distance(f) == 1.024
    assert a.with_duration(7).duration() == 7 and a.with_src_path("q").src_path() == "q"
    assert vdf.VideoHash() == vdf.VideoHash.empty_hash("")  # Default, video_hash.rs:34-42
    assert f.hash_bits().sum() == 1000 and len(f.hash_bits()) == 1000
    assert len({a, vdf.VideoHash.empty_hash("p"), f}) == 2


def test_errors_mirror_the_reference_enum():
    assert str(vdf.NotEnoughFrames()) == "Could not extract enough frames"
    assert str(vdf.NotVideo()) == "File is not a video"
    assert str(vdf.VidProc("x")) == "Video processing error: x"
    assert issubclass(vdf.NotEnoughFrames, vdf.Error)
    # from_frames rejects short input before touching the device (video_hash.rs:53,61)
    with pytest.raises(vdf.NotEnoughFrames):
        vdf.VideoHash.from_frames([np.zeros((8, 8), np.uint8)] * 15, "p", 1)
    with pytest.raises(vdf.NotEnoughFrames):
        vdf.VideoHash.from_frames([], "p", 1)
    assert vdf.search([], 0.3) == [] and vdf.search_with_references([], [], 0.3) == []


def test_gen_hashes_option_mapping():
    assert vdf.Cropdetect.LETTERBOX.value == "letterbox"
    with pytest.raises(vdf.VidProc):
        vdf.gen_hashes(np.zeros((1, 16, 8, 8), np.uint8), ["p"], [1], cropdetect=vdf.Cropdetect.MOTION)


def test_test_util_random_constructors():
    """video_hash.rs:272-306 (feature test-util): random_hash has 1000 fair bits and zero padding; hash_with_spatial_distance
    lands at exactly the requested distance and may flip padding bits; the metric axioms of video_hash.rs:325-371 hold."""
    rng = np.random.default_rng(3)
    a = vdf.VideoHash.random_hash(rng)
    assert str(a.src_path()) in ("", ".")
    assert a.duration() == 0
    bits = np.unpackbits(a.hash.view(np.uint8), bitorder="little")
    assert not bits[1000:].any() and 400 < bits[:1000].sum() < 600
    for d in (0, 1, 350, 600, 1024):
        b = a.hash_with_spatial_distance(d, rng)
        assert a.hamming_distance(b) == d == b.hamming_distance(a)
    assert a.hamming_distance(a) == 0
    c = a.hash_with_spatial_distance(100, rng)
    e = c.hash_with_spatial_distance(50, rng)
    assert a.hamming_distance(e) <= a.hamming_distance(c) + c.hamming_distance(e)  # triangle inequality
    with pytest.raises(ValueError):
        a.hash_with_spatial_distance(1025, rng)
