"""HIP search path (through the C ABI) vs the CPU oracle: bit-exact MatchGroup index lists."""
import numpy as np
import pytest

import hashgen as hg
from oracle import vdf_oracle as orc

pytestmark = pytest.mark.gpu
SCALE = 1000.0


def _both_self(engine, words, dur, tol_int):
    got = engine.search_self_sorted(words, dur, tol_int)
    want = orc.search_self_sorted(words, dur, tol_int)
    assert got == want
    return got


def _both_refs(engine, cw, cd, rw, rd, tol_int):
    got = engine.search_refs_sorted(cw, cd, rw, rd, tol_int)
    want = orc.search_refs_sorted(cw, cd, rw, rd, tol_int)
    assert got == want
    return got


def test_reference_scenario_known_group(engine):
    rng = np.random.default_rng(1)
    members = np.stack(hg.HashesWithDistanceSet(1, 50, 201, 100, rng).all_members(rng))
    groups = _both_self(engine, members, np.zeros(len(members), np.uint32), 200)
    assert [len(g) for g in groups] == [50]


def test_reference_scenario_duration(engine):
    rng = np.random.default_rng(2)
    short = hg.HashesWithDistanceSet(1, 100, 201, 100, rng).groups[0].members(rng)
    words = np.stack(list(short) + list(short[:50]))
    dur = np.array([50] * 100 + [250] * 50, np.uint32)
    perm = rng.permutation(150)
    w, d, _ = hg.sort_by_duration(words[perm], dur[perm])
    groups = _both_self(engine, w, d, 200)
    assert sorted(len(g) for g in groups) == [50, 100]


def test_reference_scenario_distance(engine):
    rng = np.random.default_rng(3)
    allh = np.stack(hg.HashesWithDistanceSet(2, 100, 150, 50, rng).all_members(rng))
    groups = _both_self(engine, allh, np.zeros(len(allh), np.uint32), 100)
    assert sorted(len(g) for g in groups) == [100, 110]


def test_reference_scenario_refs(engine):
    rng = np.random.default_rng(4)
    gs = hg.HashesWithDistanceSet(5, 100, 150, 50, rng)
    cands = np.stack(gs.all_members(rng))
    zeros = np.zeros(len(cands), np.uint32)
    r1 = _both_refs(engine, cands, zeros, gs.groups[3].start_hash[None], np.zeros(1, np.uint32), 50)
    assert [(r, len(m)) for r, m in r1] == [(0, 130)]
    starts = np.stack([gs.groups[0].start_hash, gs.groups[4].start_hash])
    r2 = _both_refs(engine, cands, zeros, starts, np.zeros(2, np.uint32), 50)
    assert [(r, len(m)) for r, m in r2] == [(0, 100), (1, 140)]


def test_empty_and_tiny(engine):
    assert engine.search_self_sorted(np.zeros((0, 16), np.uint64), np.zeros(0, np.uint32), 1000) == []
    one = hg.random_hashes(np.random.default_rng(0), 1)
    assert engine.search_self_sorted(one, np.zeros(1, np.uint32), 1024) == []
    two = np.concatenate([one, one])
    assert _both_self(engine, two, np.zeros(2, np.uint32), 0) == [[1, 0]]
    assert engine.search_refs_sorted(one, np.zeros(1, np.uint32), np.zeros((0, 16), np.uint64), np.zeros(0, np.uint32), 5) == []


@pytest.mark.parametrize("n,seed", [(700, 10), (3000, 11), (20000, 12)])
@pytest.mark.parametrize("durations", ["zero", "windowed"])
def test_planted_sets_match_oracle(engine, n, seed, durations):
    rng = np.random.default_rng(seed)
    words, dur = hg.planted_set(rng, n, n_clusters=max(4, n // 50), durations=durations)
    w, d, _ = hg.sort_by_duration(words, dur)
    for tol in (350, 0, 120):
        groups = _both_self(engine, w, d, tol)
        if tol == 350:
            assert len(groups) > 0
    stats = engine.last_stats()
    assert stats["pairs"] == orc.pairs_self(d)


def test_padding_bits_count(engine):
    """hash_with_spatial_distance may set bits 1000..1023; the kernel must not mask them (video_hash.rs:311-317)."""
    base = np.zeros((1, 16), np.uint64)
    other = base.copy()
    other[0, 15] = np.uint64(0xFFFFFF) << np.uint64(40)  # 24 padding bits
    w = np.concatenate([base, other])
    d = np.zeros(2, np.uint32)
    assert _both_self(engine, w, d, 23) == []
    assert _both_self(engine, w, d, 24) == [[1, 0]]


def test_all_identical_and_hit_buffer_overflow(engine):
    """Adversarial: every pair is a hit.  Forces the hit-buffer overflow protocol (rows replayed in chunks with
    the consumption bitmap fed back); the result must still be the single group the reference builds."""
    n = 3000
    one = hg.random_hashes(np.random.default_rng(5), 1)
    w = np.repeat(one, n, axis=0)
    d = np.zeros(n, np.uint32)
    engine.set_hit_capacity(5000)
    try:
        groups = _both_self(engine, w, d, 0)
        assert len(groups) == 1 and len(groups[0]) == n
        assert engine.last_stats()["n_launches"] > 1
        # clustered + overflow
        rng = np.random.default_rng(6)
        words, dur = hg.planted_set(rng, 4000, n_clusters=40, max_copies=60, max_flips=100)
        w2, d2, _ = hg.sort_by_duration(words, dur)
        engine.set_hit_capacity(700)
        _both_self(engine, w2, d2, 350)
    finally:
        engine.set_hit_capacity(1 << 24)


def test_duration_window_edges(engine):
    """One-sided x1.1 window with truncation: (f64(d) * 1.1) as u32 (search_algorithm.rs:99)."""
    one = hg.random_hashes(np.random.default_rng(7), 1)
    durs = np.array([10, 11, 12, 100, 109, 110, 111, 1000, 1100, 1101, 4000000000, 4294967295], np.uint32)
    w = np.repeat(one, len(durs), axis=0)
    _both_self(engine, w, durs, 0)
    # +-5% windows for references: (d*0.95) as u32 .. (d*1.05) as u32, both truncating
    cd = np.arange(0, 400, dtype=np.uint32)
    cw = np.repeat(one, len(cd), axis=0)
    rd = np.array([0, 1, 19, 20, 21, 100, 199, 200, 399, 1000], np.uint32)
    rw = np.repeat(one, len(rd), axis=0)
    _both_refs(engine, cw, cd, rw, rd, 0)


@pytest.mark.parametrize("n_cand,n_ref,seed", [(2500, 300, 20), (12000, 900, 21)])
def test_refs_planted_match_oracle(engine, n_cand, n_ref, seed):
    rng = np.random.default_rng(seed)
    words, dur = hg.planted_set(rng, n_cand, n_clusters=n_cand // 40, durations="windowed")
    cw, cd, _ = hg.sort_by_duration(words, dur)
    # references: half near-copies of candidates (matching durations), half fresh
    pick = rng.choice(n_cand, size=n_ref // 2, replace=False)
    rw = cw[pick].copy()
    rd = cd[pick].copy()
    for i in range(len(rw)):
        flips = rng.choice(1024, size=int(rng.integers(0, 340)), replace=False)
        bits = np.unpackbits(rw[i].view(np.uint8), bitorder="little")
        bits[flips] ^= 1
        rw[i] = np.packbits(bits, bitorder="little").view(np.uint64)
    fresh = hg.random_hashes(rng, n_ref - len(rw))
    fd = np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=len(fresh)))).astype(np.uint32)
    rw = np.concatenate([rw, fresh])
    rd = np.concatenate([rd, fd])
    perm = rng.permutation(len(rw))
    res = _both_refs(engine, cw, cd, rw[perm], rd[perm], 350)
    assert len(res) > 0
    engine.set_hit_capacity(16)  # every hit is output: the buffer is resized exactly once
    try:
        _both_refs(engine, cw, cd, rw[perm], rd[perm], 350)
    finally:
        engine.set_hit_capacity(1 << 24)


def test_public_api_matches_oracle(engine):
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(30)
    words, dur = hg.planted_set(rng, 1200, n_clusters=30, durations="windowed")
    paths = [f"dir{int(rng.integers(0, 5))}/v{i}.mp4" if i % 3 else f"dir.{i}/v.mp4" for i in range(len(words))]
    hashes = [vdf.VideoHash(words[i], paths[i], int(dur[i])) for i in range(len(words))]
    got = vdf.search(hashes, 0.35, engine=engine)
    want = orc.search(words, dur, paths, 0.35)
    assert [list(g.duplicates()) for g in got] == want
    assert all(g.reference() is None and g.len() >= 2 for g in got)
    refs = hashes[:40]
    got_r = vdf.search_with_references(refs, hashes[20:], 0.2, engine=engine)
    want_r = orc.search_with_references(words[:40], dur[:40], paths[:40], words[20:], dur[20:], paths[20:], 0.2)
    assert [(g.reference(), list(g.duplicates())) for g in got_r] == want_r
    assert vdf.search([], 1.0, engine=engine) == []


@pytest.mark.parametrize("plain", [True, False])
def test_public_api_sorts_large_sets_through_the_engine(engine, plain):
    """From 2048 hashes on, search() / search_with_references() take Search::sort's order from the library (vdf_sort_order_paths: device for
    plain paths, the native component comparator otherwise) instead of a Python key per entry: the same order, the same groups."""
    import vid_dup_finder_lib_amd as vdf
    from vid_dup_finder_lib_amd import api

    rng = np.random.default_rng(31 + plain)
    words, dur = hg.planted_set(rng, 6000, n_clusters=80, durations="windowed")
    n = len(words)

    def path(i):
        d = int(rng.integers(0, 7))
        if plain:
            return [f"/m/dir{d}/v{i}.mp4", f"/m/dir.{d}/v{i}.mp4", f"/m/dir{d}/sub/v{i % 50}.mp4", f"m/dir{d} x/v{i}.mp4"][i % 4]
        return [f"/m//dir{d}/./v{i}.mp4", f"../m/dir{d}/v{i}.mp4", f"./m/dir{d}/v{i % 50}.mp4", f"/m/dir{d}/v{i}.mp4/"][i % 4]

    paths = [path(i) for i in range(n)]
    hashes = [vdf.VideoHash(words[i], paths[i], int(dur[i])) for i in range(n)]
    assert api.sort_order(hashes, engine) == api.sort_order(hashes)  # the library's order = the Python keys' (equal paths: input order)
    got = vdf.search(hashes, 0.35, engine=engine)
    assert [list(g.duplicates()) for g in got] == orc.search(words, dur, paths, 0.35) and len(got) >= 40
    got_r = vdf.search_with_references(hashes[:60], hashes[30:], 0.25, engine=engine)
    want_r = orc.search_with_references(words[:60], dur[:60], paths[:60], words[30:], dur[30:], paths[30:], 0.25)
    assert [(g.reference(), list(g.duplicates())) for g in got_r] == want_r


@pytest.mark.parametrize("n", [1, 31, 257, 700])
def test_large_tolerances_match_everything(engine, n):
    """tolerance >= 0.5: the fp4 backend's threshold 1024 - 2 tol goes <= 0, so even its zero padding rows/columns
    pass the fast test; the window checks must still reject them.  tolerance 1.0 matches every pair."""
    rng = np.random.default_rng(900 + n)
    w = hg.random_hashes(rng, n)
    d = np.sort(rng.integers(100, 140, size=n).astype(np.uint32))
    for tol in (512, 700, 1024, 5000):
        _both_self(engine, w, d, tol)
    rw = hg.random_hashes(rng, 9)
    rd = rng.integers(90, 150, size=9).astype(np.uint32)
    for tol in (600, 1024):
        _both_refs(engine, w, d, rw, rd, tol)


def test_groups_max_distance_matches_bruteforce(engine):
    """the app's Sorting::Distance key (search_output.rs:43-60)."""
    rng = np.random.default_rng(77)
    words, dur = hg.planted_set(rng, 3000, n_clusters=60, max_copies=40, max_flips=300)
    w, d, _ = hg.sort_by_duration(words, dur)
    groups = engine.search_self_sorted(w, d, 350)
    assert len(groups) > 10
    got = engine.groups_max_distance(w, groups)
    want = [max(orc.hamming(w[a], w[b]) for i, a in enumerate(g) for b in g[i + 1:]) for g in groups]
    assert got.tolist() == want
    # with references: contained_paths = duplicates then the reference
    pick = [g[0] for g in groups[:20]]
    rw = w[pick].copy()
    rw[:, 0] ^= np.uint64(0xFFFF)
    res = engine.search_refs_sorted(w, d, rw, d[pick], 350)
    got_r = engine.groups_max_distance(w, [m for _, m in res], ref_hashes=rw, ref_index=[r for r, _ in res])
    want_r = []
    for r, m in res:
        hs = [w[x] for x in m] + [rw[r]]
        want_r.append(max(orc.hamming(hs[i], hs[j]) for i in range(len(hs)) for j in range(i + 1, len(hs))))
    assert got_r.tolist() == want_r
    assert engine.groups_max_distance(w, []).tolist() == []


@pytest.mark.parametrize("step", [0, 7, 10, 11, 12, 13, 16])
@pytest.mark.parametrize("tol", [0, 120, 350, 600])
@pytest.mark.parametrize("backend", ["mfma", "valu"])
def test_early_exit_step_never_changes_results(backend, step, tol, monkeypatch):
    """The MFMA kernel may stop a 32 x 32 block after 64 (step + 1) bits when every partial distance already exceeds the
    tolerance.  Any step must give the oracle's groups: steps where nearly every block exits (small tol), steps where none
    does (tol 600), clustered data whose blocks contain hits, and the disabled test (16)."""
    import vid_dup_finder_lib_amd as vdf

    monkeypatch.setenv("VDF_SEARCH_BACKEND", backend)
    monkeypatch.setenv("VDF_MFMA_PRUNE_STEP", str(step))
    eng = vdf.Engine(0)
    try:
        rng = np.random.default_rng(1000 * step + tol)
        words, dur = hg.planted_set(rng, 3000, n_clusters=60, max_copies=6, max_flips=min(tol + 40, 700),
                                    durations="zero" if tol % 2 else "log")
        words, dur, _ = hg.sort_by_duration(words, dur)
        got = eng.search_self_sorted(words, dur, tol)
        want = orc.search_self_sorted(words, dur, tol)
        assert got == want
        st = eng.last_stats()
        assert st["early_exit_bits"] in (0, 448, 576, 704, 768, 832, 896)
        assert st["pairs_early_exit"] <= st["pairs_computed"]
        if step == 16:
            assert st["early_exit_bits"] == 0 and st["pairs_early_exit"] == 0
    finally:
        eng.close()


@pytest.mark.parametrize("n", [1, 2, 511, 512, 513, 1023, 1025, 4095, 4097, 8193])
def test_tile_and_chunk_boundaries(engine, n):
    """Database sizes around the 512-row workgroup tile, the 128-column LDS stage and the 4096-column minimum chunk."""
    rng = np.random.default_rng(n)
    words, dur = hg.planted_set(rng, n, n_clusters=max(1, n // 40), max_flips=360, durations="log" if n % 2 else "zero")
    words, dur, _ = hg.sort_by_duration(words, dur)
    _both_self(engine, words, dur, 350)


def test_chunk_boundary_65536(engine):
    """One candidate chunk is 65536 columns by default: a database just over it, all durations equal (every row tile
    crosses the chunk boundary), checked against the oracle on the rows around the boundary and by planted recovery."""
    n = 65536 + 700
    rng = np.random.default_rng(65536)
    words, dur = hg.planted_set(rng, n, n_clusters=300, max_flips=340, durations="zero")
    got = engine.search_self_sorted(words, dur, 350)
    want = orc.search_self_sorted(words, dur, 350)
    assert got == want


@pytest.mark.parametrize("cand_capacity", [8, 64, 1000])
def test_suspect_queue_overflow_goes_through_the_overflow_protocol(cand_capacity, monkeypatch):
    """The matrix-core backend queues suspect pairs for an exact second pass.  With a queue far too small for dense
    near-duplicates (forced here) a launch drops suspects and reports the smallest row that lost one through overflow_row;
    the caller-side protocol (vdf_search_self) must still return exactly the reference's groups."""
    import vid_dup_finder_lib_amd as vdf

    monkeypatch.setenv("VDF_CAND_CAPACITY", str(cand_capacity))
    monkeypatch.setenv("VDF_SEARCH_BACKEND", "mfma")
    eng = vdf.Engine(0)
    try:
        rng = np.random.default_rng(cand_capacity)
        words, dur = hg.planted_set(rng, 2500, n_clusters=25, max_copies=40, max_flips=120, durations="windowed")
        w, d, _ = hg.sort_by_duration(words, dur)
        assert eng.search_self_sorted(w, d, 350) == orc.search_self_sorted(w, d, 350)
        assert eng.last_stats()["n_launches"] > 1
        pick = rng.choice(len(d), size=64, replace=False)
        # references have no consumption, every hit is output: the library reruns the launch with a larger queue
        assert eng.search_refs_sorted(w, d, w[pick], d[pick], 300) == orc.search_refs_sorted(w, d, w[pick], d[pick], 300)
    finally:
        eng.close()


def test_suspect_queue_slots_beyond_a_smaller_launch_are_not_read_stale(monkeypatch):
    """One context, three searches: dense near-duplicates (the suspect queue overflows and is re-run x4 larger until it
    fits), then a tiny search (small queue again), then the dense one once more.  The slots the first search dirtied beyond
    the second search's capacity must be empty again before the third reads them: a stale suspect that is a true hit would
    be emitted twice, and search_with_references outputs every hit (duplicate group members, inflated n_hits)."""
    import vid_dup_finder_lib_amd as vdf

    monkeypatch.setenv("VDF_CAND_CAPACITY", "64")
    monkeypatch.setenv("VDF_SEARCH_BACKEND", "mfma")
    eng = vdf.Engine(0)
    try:
        rng = np.random.default_rng(77)
        words, dur = hg.planted_set(rng, 3000, n_clusters=20, max_copies=60, max_flips=100, durations="zero")
        w, d, _ = hg.sort_by_duration(words, dur)
        pick = rng.choice(len(d), size=256, replace=False)
        want_refs = orc.search_refs_sorted(w, d, w[pick], d[pick], 350)
        want_self = orc.search_self_sorted(w, d, 350)
        tiny_w, tiny_d = w[:40].copy(), d[:40].copy()
        want_tiny = orc.search_refs_sorted(tiny_w, tiny_d, tiny_w[:4], tiny_d[:4], 350)
        for _ in range(3):
            assert eng.search_refs_sorted(w, d, w[pick], d[pick], 350) == want_refs
            assert eng.search_refs_sorted(tiny_w, tiny_d, tiny_w[:4], tiny_d[:4], 350) == want_tiny
            assert eng.search_self_sorted(w, d, 350) == want_self
            assert eng.search_self_sorted(tiny_w, tiny_d, 350) == orc.search_self_sorted(tiny_w, tiny_d, 350)
    finally:
        eng.close()


def test_pinned_database_reuses_its_expansion_and_never_goes_stale(engine):
    """vdf_ctx_pin_database: searches against the pinned candidate database skip the operand expansion (matrix-core backend) -
    same results as unpinned, for changing reference sets and tolerances (another tolerance = another tested prefix = a fresh
    expansion); once the promise is withdrawn a changed database is seen again.  Also drives the speculative device-side sort
    of the hit list (second and later calls with >= 16 k hits)."""
    import torch

    rng = np.random.default_rng(123)
    words, dur = hg.planted_set(rng, 30_000, n_clusters=600, max_copies=40, max_flips=200, durations="windowed")
    w, d, _ = hg.sort_by_duration(words, dur)
    tw = torch.from_numpy(w.view(np.int64).copy()).cuda()
    td = torch.from_numpy(d.view(np.int32).copy()).cuda()
    torch.cuda.synchronize()

    def refs(rw, rd, tol):
        a = torch.from_numpy(rw.view(np.int64).copy()).cuda()
        b = torch.from_numpy(rd.view(np.int32).copy()).cuda()
        torch.cuda.synchronize()
        hits, n = engine.search_refs_device(tw.data_ptr(), td.data_ptr(), len(d), a.data_ptr(), b.data_ptr(), len(rd), tol)
        assert n == len(hits)
        from vid_dup_finder_lib_amd import engine as ve
        return ve.groups_from_ref_hits(hits)

    engine.pin_database(tw.data_ptr(), len(d))
    try:
        for k, tol in enumerate([350, 350, 350, 200, 350]):
            pick = np.random.default_rng(k).choice(len(d), size=9000, replace=False)
            got = refs(w[pick], d[pick], tol)
            assert got == orc.search_refs_sorted(w, d, w[pick], d[pick], tol), (k, tol)
            assert tol < 350 or sum(len(m) for _, m in got) > (1 << 14)  # long enough for the speculative device sort
    finally:
        engine.pin_database(0, 0)
    # promise withdrawn: overwrite part of the database in place and search again
    w2 = w.copy()
    w2[:5000] = hg.random_hashes(np.random.default_rng(9), 5000)
    tw.copy_(torch.from_numpy(w2.view(np.int64)))
    torch.cuda.synchronize()
    pick = np.random.default_rng(77).choice(len(d), size=2000, replace=False)
    assert refs(w2[pick], d[pick], 350) == orc.search_refs_sorted(w2, d, w2[pick], d[pick], 350)


@pytest.mark.parametrize("n", [1, 2, 1000, 70_000, 300_001])
def test_device_sort_order_is_search_sort(engine, n):
    """vdf_sort_order_device = Search::sort (search_algorithm.rs:55-61: stable sort_by_key on (duration, src_path)) for entries that
    stay in HBM: durations with many ties, path ranks with ties (equal paths keep input order), and without ranks (all paths equal:
    stable by duration); vdf_apply_order_device gathers hashes and durations into that order.  Checked against the oracle's
    sort_order on real path strings (Rust's component-wise PathBuf order) and numpy's stable lexsort."""
    import torch

    rng = np.random.default_rng(n)
    dur = rng.integers(0, 40 if n < 100_000 else 7200, size=n).astype(np.uint32)
    if n > 10:
        dur[rng.choice(n, size=n // 10, replace=False)] = np.uint32(4_000_000_000)  # large keys: every radix digit is live
    names = [f"dir{int(rng.integers(0, 7))}/sub.{int(rng.integers(0, 3))}/v{int(rng.integers(0, max(2, n // 3)))}.mp4" for _ in range(n)]
    order_by_path = sorted(set(names), key=orc.rust_path_key)
    rank_of = {p: i for i, p in enumerate(order_by_path)}
    rank = np.array([rank_of[p] for p in names], np.uint32)
    want = np.array(orc.sort_order(dur.tolist(), names), np.int64) if n <= 70_000 else np.lexsort((np.arange(n), rank, dur))
    assert np.array_equal(want, np.lexsort((np.arange(n), rank, dur)))
    w = hg.random_hashes(rng, n)
    t_d = torch.from_numpy(dur.view(np.int32).copy()).cuda()
    t_r = torch.from_numpy(rank.view(np.int32).copy()).cuda()
    t_w = torch.from_numpy(w.view(np.int64).copy()).cuda()
    perm = torch.zeros(n, dtype=torch.int32, device="cuda")
    w_out, d_out = torch.zeros_like(t_w), torch.zeros_like(t_d)
    torch.cuda.synchronize()
    engine.sort_order_device(t_d.data_ptr(), n, perm.data_ptr(), d_path_rank=t_r.data_ptr())
    engine.apply_order_device(t_w.data_ptr(), t_d.data_ptr(), perm.data_ptr(), n, w_out.data_ptr(), d_out.data_ptr())
    torch.cuda.synchronize()
    got = perm.cpu().numpy().view(np.uint32).astype(np.int64)
    assert np.array_equal(got, want)
    assert np.array_equal(w_out.cpu().numpy().view(np.uint64), w[want]) and np.array_equal(d_out.cpu().numpy().view(np.uint32), dur[want])
    engine.sort_order_device(t_d.data_ptr(), n, perm.data_ptr())  # no ranks: all paths equal
    torch.cuda.synchronize()
    assert np.array_equal(perm.cpu().numpy().view(np.uint32).astype(np.int64), np.argsort(dur, kind="stable"))
