"""Search::sort's (duration, src_path) order computed from the PATHS on the device (csrc/sort_order.hip: launch_path_duration_order,
behind vdf_sort_order_paths and vdf_search_cache_entries) against Python's sort by (duration, rust_path_key) - the mirror's restatement of
std::path's component order (search_algorithm.rs:55-61; PathBuf: Ord).

For plain paths (what a directory walk produces) the component order is the byte order with '/' below everything and a path that ends
before one that goes on; the device sorts 8-byte words of the re-coded bytes, last word first.  The traps: bytes below '/' (0x01, ' ',
'!', '-', '.') next to a separator, names that are prefixes of each other, dot-names that are not special, bytes >= 0x80 and 0xFF, absolute
against relative, long shared prefixes (skipped words), lengths on both sides of every 8-byte word, exact duplicates (stability decides),
equal durations and distinct ones; and a single non-plain path, or an over-long one, sends the call down the host's road - same order."""
import numpy as np
import pytest

from vid_dup_finder_lib_amd import cache as vc
from vid_dup_finder_lib_amd import rust_path_key

pytestmark = pytest.mark.gpu


def want_order(durations, paths):
    keys = [rust_path_key(p) for p in paths]
    return np.array(sorted(range(len(paths)), key=lambda i: (int(durations[i]), keys[i], i)), np.uint32)


def trap_paths(rng, n):
    alphabet = np.frombuffer(b"\x01 !-.0Aa~\x80\xff", np.uint8)

    def name(lo, hi):
        while True:
            b = bytes(alphabet[rng.integers(len(alphabet), size=int(rng.integers(lo, hi)))])
            if b not in (b".", b".."):
                return b

    roots = [b"", b"/", b"/mnt/library/videos/", b"mnt/library/videos/", b"/mnt/library/videos.old/", b"/mnt/library/videos/x/"]
    dirs = [name(1, 4) for _ in range(12)]
    paths = []
    for _ in range(n):
        depth = int(rng.integers(0, 4))
        comps = [dirs[int(rng.integers(len(dirs)))] for _ in range(depth)] + [name(1, 20)]
        paths.append(roots[int(rng.integers(len(roots)))] + b"/".join(comps))
    return paths + paths[: n // 10] + [b"", b"/", b"/mnt", b"/mnt/library/videos", b"mnt"]


@pytest.mark.parametrize("n", [1, 7, 300, 6000, 70000])
@pytest.mark.parametrize("dur_kind", ["equal", "few", "distinct"])
def test_device_order_is_the_sort_by_duration_and_component_order(engine, n, dur_kind):
    rng = np.random.default_rng(n * 3 + len(dur_kind))
    paths = trap_paths(rng, n)
    m = len(paths)
    dur = {"equal": np.full(m, 77, np.uint32), "few": rng.integers(5, 9, size=m).astype(np.uint32),
           "distinct": rng.integers(0, 2**32, size=m, dtype=np.uint64).astype(np.uint32)}[dur_kind]
    got, used = vc.sort_order_paths(engine, dur, paths)
    assert used, "plain paths must take the device road"
    assert np.array_equal(got, want_order(dur, paths))


def test_shared_prefixes_word_boundaries_and_long_paths(engine):
    rng = np.random.default_rng(5)
    base = b"/srv/media/library_of_everything/"  # 33 shared bytes: four whole words are skipped, the fifth begins inside the shared part
    paths = []
    for L in range(0, 70):  # every length across eight word boundaries
        for _ in range(6):
            paths.append(base + bytes(rng.choice(np.frombuffer(b"ab/", np.uint8), size=L)).replace(b"//", b"/a").strip(b"/"))
    paths = [p if not p.endswith(b"/") else p + b"x" for p in paths]
    paths += [base + b"x" * 990, base + b"x" * 989 + b"y", base + b"x" * 500]  # up to 1023 bytes: 128 words
    dur = rng.integers(10, 13, size=len(paths)).astype(np.uint32)
    got, used = vc.sort_order_paths(engine, dur, paths)
    assert used and np.array_equal(got, want_order(dur, paths))
    # one byte over the limit: the host's road, the same order
    paths.append(base + b"z" * 1000)
    dur = np.append(dur, np.uint32(11))
    got, used = vc.sort_order_paths(engine, dur, paths)
    assert not used and np.array_equal(got, want_order(dur, paths))


def test_one_path_that_is_not_plain_takes_the_hosts_road(engine):
    rng = np.random.default_rng(9)
    paths = trap_paths(rng, 2000)
    dur = rng.integers(0, 50, size=len(paths)).astype(np.uint32)
    for bad in (b"/mnt//library/./videos/a", b"a/../b", b"trailing/", b"./x", b"nul\x00byte"):
        q = list(paths)
        q[17] = bad
        got, used = vc.sort_order_paths(engine, dur, q)
        assert not used, bad
        assert np.array_equal(got, want_order(dur, q)), bad


def test_the_cache_search_agrees_on_both_roads(engine, monkeypatch):
    """vdf_search_cache_entries with the device order against the same call with VDF_NO_DEVICE_PATH_ORDER (host ranks): identical groups -
    with durations that tie everywhere, so that the path order decides every window and every group's member order."""
    import hashgen as hg
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(31)
    n = 6000
    words, _ = hg.planted_set(rng, n, n_clusters=150, max_copies=6, max_flips=200, durations="zero")
    dur = rng.integers(100, 104, size=n).astype(np.uint32)
    paths = [f"/v/d{i % 13}/{'c' if i % 3 else 'c.'}{i * 7919 % 5000}.mkv" for i in range(n)]
    data = vc.encode_cache(words, dur, paths)
    got = vc.search_cache(data, 0.35, engine=engine)
    monkeypatch.setenv("VDF_NO_DEVICE_PATH_ORDER", "1")
    monkeypatch.setenv("VDF_SEARCH_BACKEND", engine.backend)
    host = vdf.Engine(0)
    try:
        want = vc.search_cache(data, 0.35, engine=host)
    finally:
        host.close()
    assert got == want and len(got) > 100
