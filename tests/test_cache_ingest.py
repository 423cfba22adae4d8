"""SURVEY.md 8f N1, the rest of the row: the metadata sidecar (cache_metadata.rs, video_hash_filesystem_cache.rs:76-139), Search::sort's
PathBuf order for a whole cache without per-entry objects (search_algorithm.rs:55-61 -> vdf_path_ranks), the multi-threaded decoder
and the one-call route cache bytes -> MatchGroups (vdf_search_cache_entries)."""
import os

import numpy as np
import pytest

import hashgen as hg
from vid_dup_finder_lib_amd import VdfError, rust_path_key
from vid_dup_finder_lib_amd import cache as vc

HOSTILE = [b"a/b", b"a.b", b"a//b", b"a/./b", b"./a/b", b"/a/b", b"//a/b", b"a/b/", b"a/b/.", b"a", b"a/", b"", b".", b"./", b"./.", b"..", b"../a",
           b"a/..", b"a/../b", b"/", b"//", b"/.", b"/..", b"a b", b"a\tb", b"a-b", b"a0", b"a/b c", b"a/b.c", b"a/bc", b"A/b", b"\xff\xfe/x", b"\xc3\xa9/x",
           b"\x80", b"a/\xff", b"a/b\x00c", b".a", b"..a", b"...", b".../x", b"a/.b", b"a/..b", b"./..", b"../..", b"a//", b"a/.//./b", b"ab/c", b"a/bc/d",
           b"a/b/c", b"a/b!c", b"a/b/!", b"a!", b"a!/b"]


def _sign(x):
    return (x > 0) - (x < 0)


def test_path_compare_is_rusts_component_order_on_hostile_paths():
    """vdf_path_compare against the mirror's rust_path_key (api.py: RootDir < CurDir < ParentDir < Normal(bytes), component by component) on
    every pair of a hostile list: '/' against '.', repeated separators, inner and leading '.', '..', non-UTF-8 bytes, empty paths."""
    keys = [rust_path_key(p) for p in HOSTILE]
    for i, a in enumerate(HOSTILE):
        for j, b in enumerate(HOSTILE):
            want = _sign((keys[i] > keys[j]) - (keys[i] < keys[j]))
            assert vc.path_compare(a, b) == want, (a, b)
    assert vc.path_compare(b"a/b", b"a.b") == -1 and vc.path_compare(b"a//b", b"a/./b") == 0  # bytewise order would say the opposite / differ


def _dense_ranks(paths):
    keys = [rust_path_key(p) for p in paths]
    order = {k: r for r, k in enumerate(sorted(set(keys)))}
    return np.array([order[k] for k in keys], np.uint32)


@pytest.mark.parametrize("n_threads", [1, 3, 8])
def test_path_ranks_equal_the_python_sort_key(n_threads):
    rng = np.random.default_rng(11)
    assert np.array_equal(vc.path_ranks(HOSTILE, n_threads), _dense_ranks(HOSTILE))
    # a cache-like set, large enough for the sample sort's buckets (>= 4096 entries), with duplicates, prefixes and the separator traps
    dirs = [b"/lib", b"/lib/a", b"/lib/a.b", b"/lib/a b", b"/lib//a", b"/lib/./a", b"lib", b"./lib", b"../lib", b"/lib/a/b", b"/lib/ab"]
    paths = []
    for i in range(30000):
        d = dirs[int(rng.integers(len(dirs)))]
        stem = bytes(rng.integers(0x20, 0x7F, size=int(rng.integers(0, 9)), dtype=np.uint8)).replace(b"/", b"_")
        paths.append(d + b"/" + stem + (b"" if rng.random() < 0.3 else b".mp4"))
    paths += paths[:500]  # exact duplicates
    got = vc.path_ranks(paths, n_threads)
    assert np.array_equal(got, _dense_ranks(paths))
    assert vc.path_ranks([], n_threads).shape == (0,)


@pytest.mark.parametrize("n_threads", [1, 4])
def test_path_ranks_of_plain_paths_take_the_byte_order_road(n_threads):
    """Paths as a directory walk produces them (no empty, "." or ".." component - csrc/path_order.cpp: is_plain) are ranked without
    component iteration: byte order with '/' below everything, 8-byte words riding along with the indices.  The traps of that road:
    bytes below '/' (' ', '!', '-', '.', 0x01) next to a separator, names that are prefixes of each other (a path that ends against
    one that goes on with '/' or with a byte), dot-names that are NOT special (".a", "..a", "..."), bytes >= 0x80, absolute against
    relative, long shared prefixes, exact duplicates, and sizes on both sides of the sample sort's threshold."""
    rng = np.random.default_rng(23)
    alphabet = np.frombuffer(b"\x01 !-.0Aa~\x80\xff", np.uint8)

    def name(lo, hi):
        while True:
            b = bytes(alphabet[rng.integers(len(alphabet), size=int(rng.integers(lo, hi)))])
            if b not in (b".", b".."):
                return b

    for n in (300, 6000, 40000):
        roots = [b"", b"/", b"/mnt/library/videos/", b"mnt/library/videos/", b"/mnt/library/videos.old/", b"/mnt/library/videos/x/"]
        dirs = [name(1, 4) for _ in range(12)]
        paths = []
        for _ in range(n):
            depth = int(rng.integers(0, 4))
            comps = [dirs[int(rng.integers(len(dirs)))] for _ in range(depth)] + [name(1, 12)]
            paths.append(roots[int(rng.integers(len(roots)))] + b"/".join(comps))
        paths += paths[: n // 10] + [b"", b"/", b"/mnt", b"/mnt/library/videos", b"mnt"]
        assert np.array_equal(vc.path_ranks(paths, n_threads), _dense_ranks(paths)), n
    # one non-plain path among them sends the whole call down the general road: same answer
    paths[17] = b"/mnt//library/./videos/a"
    assert np.array_equal(vc.path_ranks(paths, n_threads), _dense_ranks(paths))


def test_path_ranks_take_the_decoders_blob_without_objects():
    paths = [f"/v/d{i % 13}/c{i * 7919 % 5000}.mkv" for i in range(5000)]
    rng = np.random.default_rng(2)
    c = vc.decode_cache(vc.encode_cache(hg.random_hashes(rng, 5000), np.zeros(5000, np.uint32), paths))
    assert isinstance(c["paths"], vc.PathTable) and not isinstance(c["paths"].blob, bytes)  # a view of the decoder's buffer
    assert np.array_equal(vc.path_ranks(c["paths"]), _dense_ranks([p.encode() for p in paths]))


# ---- metadata sidecar ------------------------------------------------------------------------------------------------------
def test_metadata_disk_format_is_the_apps():
    """to_disk_fmt = "{:?},{:?},{:?},{},{}" (cache_metadata.rs:80-89): Debug names, Rust's f64 Display (shortest round trip, never an
    exponent, integral values without ".0"), the version."""
    m = vc.CacheMetadata.new("letterbox", 15.0)
    assert m.to_disk_fmt() == "Unix,FfmpegBackend,Letterbox,15,1"
    assert vc.CacheMetadata.new("none", 0.0).to_disk_fmt() == "Unix,FfmpegBackend,None,0,1"
    assert vc.CacheMetadata.new("motion", 0.1).to_disk_fmt() == "Unix,FfmpegBackend,Motion,0.1,1"
    for v, text in [(1e21, "1000000000000000000000"), (1e-7, "0.0000001"), (2.5, "2.5"), (1 / 3, "0.3333333333333333"), (-0.0, "-0"),
                    (float("inf"), "inf"), (float("nan"), "NaN"), (123456789.125, "123456789.125")]:
        assert vc.CacheMetadata.new("none", v).to_disk_fmt().split(",")[3] == text
    win = vc.CacheMetadata(0, 1, 2, 3.0, 7)
    assert win.to_disk_fmt() == "Windows,GstreamerBackend,Motion,3,7"
    with pytest.raises(VdfError):
        vc.CacheMetadata(5, 0, 0, 0.0, 1).to_disk_fmt()


def test_metadata_parse_follows_try_parse():
    """cache_metadata.rs:91-125: five fields; operating system / backend trimmed and lower-cased; crop the exact variant name; numbers by
    str::parse (no white space, '+' allowed, exponents, inf / nan; u64 without sign or overflow)."""
    P = vc.CacheMetadata.try_parse
    m = P("Unix,FfmpegBackend,Letterbox,15,1")
    assert m == vc.CacheMetadata.new("letterbox", 15.0) and P(m.to_disk_fmt()) == m
    assert P("  WINDOWS\t, gstreamerBACKEND ,Motion,+1.5e1,+2") == vc.CacheMetadata(0, 1, 2, 15.0, 2)
    # str::trim takes Unicode White_Space, not only ASCII: NBSP, NEL, EN QUAD, IDEOGRAPHIC SPACE, LINE SEPARATOR (ZERO WIDTH SPACE is not one)
    assert P("\u00a0\u0085Unix\u2000\u3000,\u2028FfmpegBackend\u205f,None,0,1") == vc.CacheMetadata.new("none", 0.0)
    with pytest.raises(vc.CacheMetadataError, match="Could not parse operating_system"):
        P("\u200bUnix,FfmpegBackend,None,0,1")
    for f, v in [("1.", 1.0), (".5", 0.5), ("1e3", 1000.0), ("1E-2", 0.01), ("-3", -3.0), ("inf", float("inf")), ("-Infinity", float("-inf")),
                 ("1e999", float("inf")), ("1e-999", 0.0), ("007", 7.0)]:
        assert P(f"unix,ffmpegbackend,None,{f},1").skip_forward_amount == v
    assert np.isnan(P("unix,ffmpegbackend,None,NaN,1").skip_forward_amount)
    for bad, msg in [("Unix,FfmpegBackend,Letterbox,15", "Could not parse cache metadata"), ("Unix,FfmpegBackend,Letterbox,15,1,", "Could not parse cache metadata"),
                     ("", "Could not parse cache metadata"), ("Linux,FfmpegBackend,None,0,1", "Could not parse operating_system. Got Linux"),
                     ("Unix,Ffmpeg,None,0,1", "Could not parse decode_backend"), ("Unix,FfmpegBackend,letterbox,0,1", "Could not parse crop. Got letterbox"),
                     ("Unix,FfmpegBackend, None,0,1", "Could not parse crop"), ("Unix,FfmpegBackend,None, 0,1", "Could not parse skip_forward amount"),
                     ("Unix,FfmpegBackend,None,0x10,1", "skip_forward"), ("Unix,FfmpegBackend,None,.,1", "skip_forward"), ("Unix,FfmpegBackend,None,1e,1", "skip_forward"),
                     ("Unix,FfmpegBackend,None,nan(1),1", "skip_forward"), ("Unix,FfmpegBackend,None,,1", "skip_forward"),
                     ("Unix,FfmpegBackend,None,0,1\n", "Could not parse cache_version"), ("Unix,FfmpegBackend,None,0,-1", "cache_version"),
                     ("Unix,FfmpegBackend,None,0,1.0", "cache_version"), ("Unix,FfmpegBackend,None,0,18446744073709551616", "cache_version"),
                     ("Unix,FfmpegBackend,None,0,", "cache_version")]:
        with pytest.raises(vc.CacheMetadataError, match=msg):
            P(bad)
    assert P("Unix,FfmpegBackend,None,0,18446744073709551615").cache_version == 2**64 - 1


def test_metadata_validate_reports_the_first_mismatch():
    """cache_metadata.rs:127-168."""
    m = vc.CacheMetadata.new("letterbox", 15.0)
    m.validate("letterbox", 15.0)
    with pytest.raises(vc.CacheMetadataError, match="crop mismatch: Act: Letterbox, Exp: None"):
        m.validate("none", 15.0)
    with pytest.raises(vc.CacheMetadataError, match=r"skip_forward_amount mismatch: Act: 15\.0, Exp: 0\.5"):
        m.validate("letterbox", 0.5)
    with pytest.raises(vc.CacheMetadataError, match="operating_system mismatch: Act: Windows, Exp: Unix"):
        vc.CacheMetadata(0, 1, 0, 1.0, 2).validate("letterbox", 15.0)  # the first difference wins
    with pytest.raises(vc.CacheMetadataError, match="decode_backend mismatch: Act: GstreamerBackend, Exp: FfmpegBackend"):
        vc.CacheMetadata(1, 1, 1, 15.0, 1).validate("letterbox", 15.0)
    with pytest.raises(vc.CacheMetadataError, match="cache_version mismatch: Act: 2, Exp: 1"):
        vc.CacheMetadata(1, 0, 1, 15.0, 2).validate("letterbox", 15.0)
    with pytest.raises(vc.CacheMetadataError, match="skip_forward_amount mismatch: Act: NaN, Exp: NaN"):
        vc.CacheMetadata.new("none", float("nan")).validate("none", float("nan"))  # f64 != : NaN never validates, as in the app
    with pytest.raises(vc.CacheMetadataError, match=r"Act: 1e16, Exp: 1\.5e-7"):
        vc.CacheMetadata.new("none", 1e16).validate("none", 1.5e-7)  # Rust's {:?} switches to the exponent form there


def test_metadata_path_is_file_stem_plus_suffix():
    """video_hash_filesystem_cache.rs:93-104: Path::file_stem + with_file_name."""
    for p, want in [("/home/u/.cache/vid_dup_finder/vid_dup_finder_cache.bin", "/home/u/.cache/vid_dup_finder/vid_dup_finder_cache.metadata.txt"),
                    ("cache", "cache.metadata.txt"), ("c.tar.gz", "c.tar.metadata.txt"), (".hidden", ".hidden.metadata.txt"), ("d/.h.bin", "d/.h.metadata.txt"),
                    ("d/c.bin/", "d/c.metadata.txt"), ("d/c.bin/.", "d/c.metadata.txt"), ("/c.", "/c.metadata.txt"), ("a/b.c/d", "a/b.c/d.metadata.txt"),
                    # with_file_name pops to Path::parent(), which normalises the directory's tail: repeated separators and "." pieces go
                    ("d//c.bin", "d/c.metadata.txt"), ("d/./c.bin", "d/c.metadata.txt"), ("//c.bin", "/c.metadata.txt"), ("./c.bin", "./c.metadata.txt"),
                    ("d/s/.//c.bin/", "d/s/c.metadata.txt"), ("/c.bin", "/c.metadata.txt")]:
        assert vc.metadata_path(p) == want, p
    for bad in ["..", "/", "a/..", ".", ""]:
        with pytest.raises(VdfError):
            vc.metadata_path(bad)


def test_cache_files_carry_their_sidecar_and_refuse_other_crop_modes(tmp_path):
    """A cache the app loads = the bincode file + <stem>.metadata.txt; a reader checks the sidecar against its own hashing mode before it mixes
    the entries with hashes of this engine (a Cropdetect::None cache holds other hashes for letterboxed files)."""
    rng = np.random.default_rng(3)
    h = hg.random_hashes(rng, 50)
    d = rng.integers(1, 5000, size=50).astype(np.uint32)
    paths = [f"/v/{i}.mp4" for i in range(50)]
    cp = tmp_path / "hashes.bin"
    vc.write_cache_files(cp, h, d, paths, cropdetect="letterbox", skip_forward_amount=15.0)
    assert (tmp_path / "hashes.metadata.txt").read_text() == "Unix,FfmpegBackend,Letterbox,15,1"
    c = vc.load_cache_files(cp, "letterbox", 15.0)
    assert np.array_equal(c["hashes"], h) and c["paths"] == paths
    with pytest.raises(vc.CacheMetadataError, match="crop mismatch"):
        vc.load_cache_files(cp, "none", 15.0)
    with pytest.raises(vc.CacheMetadataError, match="skip_forward_amount mismatch"):
        vc.load_cache_files(cp, "letterbox", 0.0)
    os.remove(tmp_path / "hashes.metadata.txt")
    with pytest.raises(FileNotFoundError, match="Cache exists but metadata is absent"):  # the app: error! + exit(1)
        vc.load_cache_files(cp, "letterbox", 15.0)


# ---- multi-threaded decoder -------------------------------------------------------------------------------------------------
def _varint(v):
    if v < 251:
        return bytes([v])
    if v < 1 << 16:
        return bytes([251]) + v.to_bytes(2, "little")
    if v < 1 << 32:
        return bytes([252]) + v.to_bytes(4, "little")
    return bytes([253]) + v.to_bytes(8, "little")


def _entry_len(path: str, words, dur, secs=0, nanos=0):
    """bytes of one Ok entry as the encoder writes it (key, mtime, variant, 16 words, src_path, duration)"""
    pb = path.encode()
    return 2 * (len(_varint(len(pb))) + len(pb)) + len(_varint(secs)) + len(_varint(nanos)) + 1 + sum(len(_varint(int(w))) for w in words) + len(_varint(int(dur)))


def _mixed_cache(n, rng, bait=False):
    """n Ok entries through the encoder, with Err entries (by hand from the bincode rules) spliced in between, hash words that take the short
    varint forms, and - bait - paths that CONTAIN the byte pattern the decoder resynchronises on."""
    h = hg.random_hashes(rng, n)
    h[::5, 3] = 9
    h[1::7, 15] = 70000
    h[2::11] = 0
    d = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    paths = [f"/lib/{i % 23}/é{i}.mkv" for i in range(n)]
    if bait:
        fake = "\x00" + "".join("ý" + "abcdefg" for _ in range(16))  # (UTF-8: not the raw 0xFD pattern, see below for the byte form)
        for i in range(0, n, 97):
            paths[i] = paths[i] + fake
    body = vc.encode_cache(h, d, paths)
    assert body[:1] != b"" and n >= 251
    head_len = len(_varint(n))
    at = head_len  # split the encoder's body at the entry boundaries
    ends = [_entry_len(paths[i], h[i], d[i]) for i in range(n)]
    n_err = 0
    out = bytearray()
    for i in range(n):
        out += body[at:at + ends[i]]
        at += ends[i]
        if i % 50 == 7:
            kind = (i // 50) % 3
            pb = f"/bad/{i}".encode()
            out += _varint(len(pb)) + pb + _varint(i) + _varint(5) + bytes([1, kind])
            if kind == 1:
                out += _varint(3) + b"msg"
            n_err += 1
    assert at == len(body)
    return _varint(n + n_err) + bytes(out), h, d, paths, n_err


@pytest.mark.parametrize("bait", [False, True])
def test_parallel_decode_equals_the_sequential_one(bait):
    rng = np.random.default_rng(17)
    data, h, d, paths, n_err = _mixed_cache(6000, rng, bait)
    ref = vc.decode_cache(data, n_threads=1)
    assert ref["n_err"] == n_err and ref["n_entries"] == 6000 + n_err and ref["paths"] == paths
    assert np.array_equal(ref["hashes"], h) and np.array_equal(ref["durations"], d)
    from vid_dup_finder_lib_amd import _capi

    before = _capi.load().vdf_cache_decode_fallbacks()
    for nt in (2, 5, 16, 64):
        c = vc.decode_cache(data, n_threads=nt)
        for k in ("hashes", "durations", "mtime_secs", "mtime_nanos"):
            assert np.array_equal(c[k], ref[k]), (nt, k)
        assert c["paths"] == paths and (c["n_entries"], c["n_err"], c["n_key_differs"]) == (ref["n_entries"], n_err, 0)
    assert _capi.load().vdf_cache_decode_fallbacks() == before  # well-formed caches decode in parallel: no fallback
    for cut in (len(data) // 3, len(data) - 1):  # malformed input fails on every thread count
        for nt in (1, 4):
            with pytest.raises(VdfError):
                vc.decode_cache(data[:cut], n_threads=nt)
    with pytest.raises(VdfError):
        vc.decode_cache(data + b"\x00", n_threads=4)


def test_a_false_resynchronisation_point_falls_back_to_the_sequential_decode():
    """The raw pattern (variant 0, then 253 at stride 9) planted inside a VidProc message: a worker that starts there parses garbage or ends
    off the next range's start; the decoder must notice and decode front to back."""
    rng = np.random.default_rng(23)
    n = 3000
    h = hg.random_hashes(rng, n)
    d = np.arange(n, dtype=np.uint32)
    paths = [f"/p/{i}" for i in range(n)]
    body = vc.encode_cache(h, d, paths)
    head = len(_varint(n))
    fake = bytes([0]) + b"".join(bytes([253]) + bytes(8) for _ in range(16)) + bytes([2]) + b"zz" + bytes([9])  # looks like hash + path + duration
    msg = b"x" * 40 + fake * 100 + b"y" * 40
    err_entry = _varint(4) + b"/err" + _varint(1) + _varint(2) + bytes([1, 1]) + _varint(len(msg)) + msg
    # an Err entry with that message after every 200th entry
    ends = [_entry_len(paths[i], h[i], d[i]) for i in range(n)]
    out, at, n_err = bytearray(), head, 0
    for i in range(n):
        out += body[at:at + ends[i]]
        at += ends[i]
        if i % 100 == 50:
            out += err_entry
            n_err += 1
    data = _varint(n + n_err) + bytes(out)
    from vid_dup_finder_lib_amd import _capi

    ref = vc.decode_cache(data, n_threads=1)
    assert ref["n_err"] == n_err and np.array_equal(ref["hashes"], h)
    before = _capi.load().vdf_cache_decode_fallbacks()
    for nt in (3, 8, 32):
        c = vc.decode_cache(data, n_threads=nt)
        assert np.array_equal(c["hashes"], h) and np.array_equal(c["durations"], d) and c["paths"] == paths and c["n_err"] == n_err
    assert _capi.load().vdf_cache_decode_fallbacks() > before  # the bait was taken at least once (90 % of the file's bytes are bait)


# ---- cache bytes -> groups on the GPU ---------------------------------------------------------------------------------------
def _planted_cache(rng, n, tie_paths=False):
    words, dur = hg.planted_set(rng, n, n_clusters=max(n // 40, 4), durations="windowed")
    perm = rng.permutation(len(dur))  # a HashMap's order: arbitrary
    words, dur = words[perm], dur[perm]
    if tie_paths:  # equal durations + the separator traps: the path order decides the greedy grouping
        dur[:] = 100
        stems = ["a/b", "a.b", "a b", "a/b/c", "a//b", "ab", "./a", "/a", "../a", "a"]
        paths = [f"{stems[i % len(stems)]}/v{i // len(stems):05d}.mp4" for i in range(len(dur))]
    else:
        paths = [f"/lib/{i % 7}/v{i}.mp4" for i in range(len(dur))]
    return words, dur, paths


@pytest.mark.gpu
@pytest.mark.parametrize("tie_paths", [False, True])
def test_search_cache_equals_the_per_entry_route_and_the_oracle(engine, tie_paths):
    """search_cache(bytes) = decode -> vdf_path_ranks -> upload -> vdf_sort_order_device -> gather -> search(), against
    vdf.search(video_hashes_from_cache(bytes)) (one Python object and one key tuple per entry) and the oracle."""
    import vid_dup_finder_lib_amd as vdf
    from oracle import vdf_oracle as orc

    rng = np.random.default_rng(5)
    words, dur, paths = _planted_cache(rng, 1500, tie_paths)
    data = vc.encode_cache(words, dur, paths)
    got = vc.search_cache(data, 0.35, engine=engine)
    slow = vdf.search(vc.video_hashes_from_cache(data), 0.35, engine=engine)
    want = orc.search(words, dur, paths, 0.35)
    assert [list(g.duplicates()) for g in got] == [list(g.duplicates()) for g in slow] == want and len(want) >= 4
    assert all(g.reference() is None for g in got)


@pytest.mark.gpu
def test_search_cache_with_selections_and_references(engine):
    """The app's --files / --with-refs filters (app_fns.rs:428-447) as index selections: members and references come back as indices into the
    cache's own arrays."""
    import vid_dup_finder_lib_amd as vdf
    from oracle import vdf_oracle as orc

    rng = np.random.default_rng(8)
    words, dur, paths = _planted_cache(rng, 2400)
    cache = vc.decode_cache(vc.encode_cache(words, dur, paths))
    idx = rng.permutation(len(dur))
    cand, refs = np.sort(idx[:1700]), idx[1700:2100]  # references in an arbitrary order
    got = vc.search_cache(cache, 0.35, engine=engine, cand_idx=cand)
    want = orc.search(words[cand], dur[cand], [paths[i] for i in cand], 0.35)
    assert [list(g.duplicates()) for g in got] == want and len(want) >= 2
    got_r = vc.search_cache(cache, 0.35, engine=engine, cand_idx=cand, ref_idx=refs)
    hashes = vc.video_hashes_from_cache(vc.encode_cache(words, dur, paths))
    by_path = {h.src_path(): h for h in hashes}
    slow = vdf.search_with_references([by_path[paths[i]] for i in refs], [by_path[paths[i]] for i in cand], 0.35, engine=engine)
    assert [(g.reference(), list(g.duplicates())) for g in got_r] == [(g.reference(), list(g.duplicates())) for g in slow] and len(slow) >= 2
    assert vc.search_cache(cache, 0.35, engine=engine, cand_idx=np.zeros(0, np.uint64)) == []
    with pytest.raises(VdfError):
        vc.search_cache_arrays(cache, 0.35, engine, cand_idx=[len(dur)])


@pytest.mark.gpu
def test_search_cache_on_a_multi_device_context():
    import vid_dup_finder_lib_amd as vdf
    from oracle import vdf_oracle as orc

    rng = np.random.default_rng(9)
    words, dur, paths = _planted_cache(rng, 3000, True)
    cache = vc.decode_cache(vc.encode_cache(words, dur, paths))
    eng = vdf.Engine(devices=[0, 0, 0])
    try:
        got = vc.search_cache(cache, 0.35, engine=eng)
        assert [list(g.duplicates()) for g in got] == orc.search(words, dur, paths, 0.35)
        refs = rng.permutation(3000)[:300]
        got_r = vc.search_cache(cache, 0.35, engine=eng, ref_idx=refs)
        one = vdf.Engine(0)
        want_r = vc.search_cache(cache, 0.35, engine=one, ref_idx=refs)
        one.close()
        assert [(g.reference(), list(g.duplicates())) for g in got_r] == [(g.reference(), list(g.duplicates())) for g in want_r] and len(want_r) >= 2
    finally:
        eng.close()
