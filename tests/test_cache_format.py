"""the app's bincode-2 hash cache <-> SoA.  No sample cache ships with the reference, so the
decoder is pinned by bytes assembled BY HAND from the bincode 2 `config::standard()` rules (varint: < 251 one byte,
251 + u16, 252 + u32, 253 + u64; little endian) and the serde shapes of the cached types
(base_fs_cache.rs:26,106-118; processing_fs_cache.rs:23-27; generic_cache_if.rs:23; video_hash.rs:26-32;
video_hashing/mod.rs:17-28), independently of the library's encoder, plus a round trip."""
import numpy as np
import pytest

import hashgen as hg
from vid_dup_finder_lib_amd import VdfError
from vid_dup_finder_lib_amd import cache as vc


def _hand_bytes():
    words = [1, 300, 1 << 32, 1 << 63, 65535, 65536, 250, 251] + [0] * 8
    b = bytearray([4])                                        # map length 4
    # entry 0: "a" -> Ok(VideoHash)
    b += bytes([1, 0x61, 5, 7, 0])                            # key "a", mtime 5 s 7 ns, variant 0 = Ok
    b += bytes([1])                                           # 1
    b += bytes([251, 0x2C, 0x01])                             # 300
    b += bytes([253, 0, 0, 0, 0, 1, 0, 0, 0])                 # 2^32
    b += bytes([253, 0, 0, 0, 0, 0, 0, 0, 0x80])              # 2^63
    b += bytes([251, 0xFF, 0xFF])                             # 65535
    b += bytes([252, 0, 0, 1, 0])                             # 65536
    b += bytes([250])                                         # 250
    b += bytes([251, 251, 0])                                 # 251
    b += bytes([0] * 8)
    b += bytes([1, 0x61])                                     # src_path "a"
    b += bytes([251, 251, 0])                                 # duration 251
    # entry 1: "b/c" -> Err(VidProc("x")), mtime 2^40 s 999_999_999 ns
    b += bytes([3]) + b"b/c" + bytes([253, 0, 0, 0, 0, 0, 1, 0, 0]) + bytes([252, 0xFF, 0xC9, 0x9A, 0x3B])
    b += bytes([1, 1, 1, 0x78])                               # Err, VidProc, "x"
    # entry 2: "d" -> Err(NotVideo); entry 3: "e" -> Err(NotEnoughFrames)
    b += bytes([1, 0x64, 0, 0, 1, 0])
    b += bytes([1, 0x65, 0, 0, 1, 2])
    return bytes(b), words


def test_hand_assembled_cache_decodes():
    data, words = _hand_bytes()
    c = vc.decode_cache(data)
    assert (c["n_entries"], c["n_err"], c["n_key_differs"]) == (4, 3, 0)
    assert c["paths"] == ["a"] and c["durations"].tolist() == [251]
    assert [int(x) for x in c["hashes"][0]] == words
    assert c["mtime_secs"].tolist() == [5] and c["mtime_nanos"].tolist() == [7]


def test_encoder_emits_exactly_the_spec_bytes_for_the_ok_entry():
    data, words = _hand_bytes()
    ok_only = bytes([1]) + data[1:1 + 5 + 1 + 3 + 9 + 9 + 3 + 5 + 1 + 3 + 8 + 2 + 3]
    enc = vc.encode_cache(np.array([words], np.uint64), [251], ["a"], [5], [7])
    assert enc == ok_only


def test_round_trip_random():
    rng = np.random.default_rng(0)
    n = 3000
    h = hg.random_hashes(rng, n)
    h[0, :] = 0
    h[1, :] = np.uint64(0xFFFFFFFFFFFFFFFF)  # full_hash: padding bits set
    d = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    paths = [f"/videos/dir{i % 17}/clip é{i}.mkv" for i in range(n)]  # non-ASCII UTF-8
    secs = rng.integers(0, 2**40, size=n, dtype=np.uint64)
    nanos = rng.integers(0, 10**9, size=n, dtype=np.uint64).astype(np.uint32)
    c = vc.decode_cache(vc.encode_cache(h, d, paths, secs, nanos))
    assert c["n_entries"] == n and c["n_err"] == 0 and c["n_key_differs"] == 0
    assert np.array_equal(c["hashes"], h) and np.array_equal(c["durations"], d) and c["paths"] == paths
    assert np.array_equal(c["mtime_secs"], secs) and np.array_equal(c["mtime_nanos"], nanos)
    e = vc.decode_cache(vc.encode_cache(np.zeros((0, 16), np.uint64), [], []))
    assert e["n_entries"] == 0 and e["paths"] == []


def test_malformed_input_is_rejected():
    data, _ = _hand_bytes()
    for cut in (0, 1, 5, 20, len(data) - 1):
        with pytest.raises(VdfError):
            vc.decode_cache(data[:cut])
    with pytest.raises(VdfError):
        vc.decode_cache(data + b"\x00")            # trailing bytes
    bad = bytearray(data)
    bad[5] = 2                                     # Result variant 2 does not exist
    with pytest.raises(VdfError):
        vc.decode_cache(bytes(bad))
    with pytest.raises(VdfError):
        vc.decode_cache(bytes([254]))              # u128 marker: not a valid length


def _check_against_cases(c, cases):
    """decoded cache `c` against the entries of tools/ref_vectors/cache_cases.py (a HashMap's order is arbitrary)."""
    ok = [x for x in cases if x[1] == 0]
    assert c["n_entries"] == len(cases) and c["n_err"] == len(cases) - len(ok) and c["n_key_differs"] == 0
    assert len(c["paths"]) == len(ok) == len(set(c["paths"]))
    at = {p: i for i, p in enumerate(c["paths"])}
    for path, _, words, dur, secs, nanos, _ in ok:
        i = at[path]
        assert np.array_equal(c["hashes"][i], words), path
        assert (int(c["durations"][i]), int(c["mtime_secs"][i]), int(c["mtime_nanos"][i])) == (dur, secs, nanos), path


def _cases():
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ref_vectors"))
    from cache_cases import cache_cases

    return cache_cases()


def test_app_written_cache_decodes():
    """tests/golden/ref_cache.bin = the cache_cases.py entries written by the APP'S OWN cache writer (tools/ref_vectors/
    cache_dump.rs: BaseFsCache::insert + save, base_fs_cache.rs:56-165).  Skips until a maintainer with cargo has generated it."""
    import os

    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_cache.bin")
    if not os.path.exists(p):
        pytest.skip("ref_cache.bin not present: generate it with the real app (tools/ref_vectors/README.md)")
    _check_against_cases(vc.decode_cache(open(p, "rb").read()), _cases())


def test_cache_handoff_comparison_is_sound():
    """The same comparison on a file synthesised here: the Ok entries through the library's encoder, the three Err entries
    appended by hand from the bincode rules (variant index as a varint, VidProc's string length-prefixed)."""
    cases = _cases()
    ok = [x for x in cases if x[1] == 0]
    body = vc.encode_cache(np.stack([x[2] for x in ok]), [x[3] for x in ok], [x[0] for x in ok], [x[4] for x in ok],
                           [x[5] for x in ok])
    assert body[0] == len(ok) < 251  # one-byte map length

    def varint(v):
        if v < 251:
            return bytes([v])
        if v < 1 << 16:
            return bytes([251]) + v.to_bytes(2, "little")
        if v < 1 << 32:
            return bytes([252]) + v.to_bytes(4, "little")
        return bytes([253]) + v.to_bytes(8, "little")

    extra = b""
    for path, kind, _, _, secs, nanos, msg in cases:
        if kind == 0:
            continue
        pb = path.encode()
        extra += varint(len(pb)) + pb + varint(secs) + varint(nanos) + bytes([1, kind - 1])
        if kind == 2:
            extra += varint(len(msg.encode())) + msg.encode()
    data = bytes([len(cases)]) + body[1:] + extra
    _check_against_cases(vc.decode_cache(data), cases)
    with pytest.raises(AssertionError):
        bad = list(cases)
        bad[3] = (bad[3][0], 0, bad[3][2], bad[3][3] + 1, bad[3][4], bad[3][5], "")
        _check_against_cases(vc.decode_cache(data), bad)


@pytest.mark.gpu
def test_cache_to_search_end_to_end(engine):
    """cache bytes -> SoA -> search(): the route a 10 M-hash user cache takes to the GPU."""
    import vid_dup_finder_lib_amd as vdf
    from oracle import vdf_oracle as orc

    rng = np.random.default_rng(5)
    words, dur = hg.planted_set(rng, 1500, n_clusters=40, durations="windowed")
    paths = [f"lib/{i % 7}/v{i}.mp4" for i in range(len(dur))]
    hashes = vc.video_hashes_from_cache(vc.encode_cache(words, dur, paths))
    got = vdf.search(hashes, 0.35, engine=engine)
    assert [list(g.duplicates()) for g in got] == orc.search(words, dur, paths, 0.35)


def test_decoded_arrays_are_views_that_outlive_the_dict_and_paths_are_lazy():
    """decode_cache hands out views of the decoder's buffers (freed when the last view goes) and a lazy path table: a 1 M-entry cache
    decodes in ~0.12 s instead of 1.9 s.  Short varints inside hashes (words < 251, < 2^16, < 2^32) take the decoder's slow path."""
    import gc

    rng = np.random.default_rng(5)
    n = 5000
    h = rng.integers(0, 2**63, size=(n, 16), dtype=np.int64).astype(np.uint64)
    h[::3, 1] = 7
    h[1::3, 15] = 60000
    h[2::3, 0] = 2**31 + 5
    h[7] = 0
    d = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    paths = [f"/v/é{i}/clip {i}.mkv" for i in range(n)]
    c = vc.decode_cache(vc.encode_cache(h, d, paths))
    assert np.array_equal(c["hashes"], h) and np.array_equal(c["durations"], d)
    p = c["paths"]
    assert len(p) == n and p[0] == paths[0] and p[-1] == paths[-1] and p[10:13] == paths[10:13] and p == paths and list(p) == paths
    assert p != paths[:-1] and repr(p) == f"PathTable({n} paths)"
    with pytest.raises(IndexError):
        p[n]
    hv = c["hashes"]
    del c, p
    gc.collect()
    assert np.array_equal(hv, h)  # the view keeps the decoder's buffer alive
