"""The duplicate-dense workload of bench.py's `dup_heavy` leg against the oracle: 10 % of the hashes in clusters of 2..200
near-copies sharing a duration - the product's own case (a duplicate finder that finds a lot).  It exercises what the sparse
random databases never do at volume: the suspect queue, wave-aggregated hit appends, the device-side hit sort, the host
replay of the greedy consumption (search_algorithm.rs:147-161) over millions of pairs."""
import os
import sys

import numpy as np
import pytest

from oracle import vdf_oracle as orc

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_dup_heavy_generator_prefix_matches_the_oracle(engine):
    import bench

    w, d, n_clusters, cluster_pairs = bench.make_dup_heavy(50_000)
    assert n_clusters > 20 and cluster_pairs > 200_000
    want = orc.search_self_sorted(w, d, 350)
    got = engine.search_self_sorted(w, d, 350)
    assert got == want
    assert len(got) == n_clusters and sum(len(g) for g in got) == 5000
    st = engine.last_stats()
    assert st["n_hits"] >= cluster_pairs  # the device evaluates the whole thresholded adjacency ...
    tm = engine.last_timing()
    # ... and, on one device, drops the pairs of rows that can never become targets of the greedy replay before they are
    # sorted, downloaded and replayed: a cluster of s mutual duplicates sends down s - 1 pairs, not s (s - 1) / 2
    assert 0 < st["n_hits"] - tm["hits_filtered"] < 3 * 5000


def test_replay_filter_never_changes_the_groups(monkeypatch):
    """The device-side filter of replay-irrelevant hits against the unfiltered path and the oracle, on data built to stress
    its argument: chains (a ~ b ~ c with a !~ c: b is consumed by a, so c must stay a target), clusters whose smallest member
    is consumed by an EARLIER unrelated target, duration windows that cut clusters, all-identical runs."""
    import vid_dup_finder_lib_amd as vdf
    import hashgen as hg

    rng = np.random.default_rng(11)
    n = 24_000
    w = hg.random_hashes(rng, n)
    d = np.sort(rng.integers(100, 130, size=n).astype(np.uint32))

    def near(src, k):
        bits = np.unpackbits(w[src].view(np.uint8), bitorder="little")
        bits[rng.choice(1024, size=k, replace=False)] ^= 1
        return np.packbits(bits, bitorder="little").view(np.uint64)

    for c in range(300):  # chains: each link 200 bits from the previous one, so links two apart are ~360 > 350 apart
        at = np.sort(rng.choice(n, size=int(rng.integers(3, 40)), replace=False))
        for a, b in zip(at[:-1], at[1:]):
            w[b] = near(a, 200)
    for c in range(60):  # dense balls (everything within 2 x 150 of everything) spread over the duration range
        at = rng.choice(n, size=int(rng.integers(50, 300)), replace=False)
        for b in at[1:]:
            w[b] = near(at[0], int(rng.integers(0, 150)))
    w[5000:5400] = w[5000]  # identical run
    want = orc.search_self_sorted(w, d, 350)
    for no_filter in ("0", "1"):
        monkeypatch.setenv("VDF_NO_HIT_FILTER", no_filter)
        for backend in ("mfma", "valu"):
            monkeypatch.setenv("VDF_SEARCH_BACKEND", backend)
            eng = vdf.Engine(0)
            try:
                assert eng.search_self_sorted(w, d, 350) == want, (no_filter, backend)
                st, tm = eng.last_stats(), eng.last_timing()
                assert st["n_hits"] > (1 << 16)
                assert (tm["hits_filtered"] > 0) == (no_filter == "0")
            finally:
                eng.close()
    monkeypatch.setenv("VDF_NO_HIT_FILTER", "0")
    two = vdf.Engine(devices=[0, 0])  # two slots: each sees only its own rows' hits, so they OR their bitmaps (multi.cpp: LocalExchange)
    try:
        assert two.search_self_sorted(w, d, 350) == want
        assert all(two.device_timing(k)["hits_filtered"] > 0 for k in range(2))
    finally:
        two.close()
    monkeypatch.setenv("VDF_NO_HIT_FILTER", "1")
    two = vdf.Engine(devices=[0, 0])
    try:
        assert two.search_self_sorted(w, d, 350) == want and two.last_timing()["hits_filtered"] == 0
    finally:
        two.close()


@pytest.mark.parametrize("slots", [2, 3])
def test_sharded_launch_filters_like_one_device(slots):
    """The replay filter of a SHARDED launch (row tiles dealt over the slots of a multi-GPU context; the device list repeats GPU 0
    here): whether a row can become a target is a property of the complete hit set, so the slots OR their has-incoming and covered
    bitmaps between the filter's steps.  Same groups as the oracle, every slot drops hits, and what comes down is the s - 1 pairs
    per cluster a single device sends - not the s (s - 1) / 2 adjacency (consumption: search_algorithm.rs:141-161)."""
    import torch

    import bench
    import vid_dup_finder_lib_amd as vdf

    w, d, n_clusters, cluster_pairs = bench.make_dup_heavy(200_000)
    want = orc.search_self_sorted(w, d, 350)
    eng = vdf.Engine(devices=[0] * slots)
    try:
        assert eng.search_self_sorted(w, d, 350) == want and len(want) == n_clusters
        st, tm = eng.last_stats(), eng.last_timing()
        assert st["n_hits"] == cluster_pairs and st["n_launches"] == 1
        per = [eng.device_timing(k)["hits_filtered"] for k in range(slots)]
        assert all(p > 0 for p in per) and sum(per) == tm["hits_filtered"]
        downloaded = st["n_hits"] - tm["hits_filtered"]
        assert downloaded == 20_000 - n_clusters  # every cluster: its smallest member's s - 1 hits
        assert downloaded <= 0.02 * cluster_pairs
        # the shards entry point (what bench.py's dup_heavy leg and a Rust caller with resident hashes use)
        cut = [len(d) * k // slots for k in range(slots + 1)]
        tw = [torch.from_numpy(w[a:b].view(np.int64)).cuda() for a, b in zip(cut[:-1], cut[1:])]
        td = [torch.from_numpy(d[a:b].view(np.int32)).cuda() for a, b in zip(cut[:-1], cut[1:])]
        torch.cuda.synchronize()
        assert eng.search_self_shards([t.data_ptr() for t in tw], [t.data_ptr() for t in td], [len(t) for t in td], 350) == want
        assert eng.last_timing()["hits_filtered"] == tm["hits_filtered"]
        # a hit buffer too small for the adjacency: nobody filters (the lists are incomplete), the overflow protocol takes over
        eng.set_hit_capacity(100_000)
        assert eng.search_self_sorted(w, d, 350) == want and eng.last_stats()["n_launches"] > 1
    finally:
        eng.close()


def test_dense_hits_take_the_device_sort_and_the_overflow_protocol(engine):
    """More than 2^17 hits in one launch: the list is sorted on the device before it comes down; with a small hit buffer the
    same database goes through the overflow protocol instead.  Same groups either way, and through the shards entry point
    of a one-device multi-GPU context (what the bench leg calls)."""
    import torch

    import bench
    import vid_dup_finder_lib_amd as vdf

    w, d, n_clusters, cluster_pairs = bench.make_dup_heavy(30_000, seed=7, frac=0.2)
    want = orc.search_self_sorted(w, d, 350)
    assert engine.search_self_sorted(w, d, 350) == want
    assert engine.last_stats()["n_hits"] > (1 << 17)
    engine.set_hit_capacity(50_000)
    try:
        assert engine.search_self_sorted(w, d, 350) == want
        assert engine.last_stats()["n_launches"] > 1
    finally:
        engine.set_hit_capacity(1 << 24)
    old = os.environ.get("VDF_SEARCH_BACKEND")
    os.environ["VDF_SEARCH_BACKEND"] = engine.backend
    try:
        one = vdf.Engine(devices=[0])
    finally:
        if old is None:
            os.environ.pop("VDF_SEARCH_BACKEND", None)
        else:
            os.environ["VDF_SEARCH_BACKEND"] = old
    try:
        tw = torch.from_numpy(w.view(np.int64)).cuda()
        td = torch.from_numpy(d.view(np.int32)).cuda()
        torch.cuda.synchronize()
        assert one.search_self_shards([tw.data_ptr()], [td.data_ptr()], [len(d)], 350) == want
        tm = one.last_timing()
        assert tm["total_ms"] >= tm["replay_ms"] >= 0 and tm["stream_ms"] > 0
        if engine.backend == "mfma":
            assert 0 < tm["suspects"] <= tm["suspect_capacity"]
    finally:
        one.close()
    # references against the dense database: every hit is output
    pick = np.random.default_rng(3).choice(len(d), size=500, replace=False)
    assert engine.search_refs_sorted(w, d, w[pick], d[pick], 350) == orc.search_refs_sorted(w, d, w[pick], d[pick], 350)


def test_dup_heavy_at_the_bench_size(engine):
    """The bench leg's own database (1 M hashes, 1015 clusters, 6.59 M thresholded pairs) through the whole host-level search()
    - stream, suspect queue, wave-staged appends, replay filter, device sort, replay: structure at full size (disjoint groups,
    every member within tolerance of its group's target, sizes), and the oracle's literal search_self on the same generator's
    400 k database (2e9 windowed comparisons on one host thread: ~5 s; the full size needs 35 s)."""
    if engine.backend != "mfma":
        pytest.skip("one backend is enough at this size (the VALU backend runs the 50 k and 30 k cases above)")
    import bench

    w, d, n_clusters, cluster_pairs = bench.make_dup_heavy(1_000_000)
    got = engine.search_self_sorted(w, d, 350)
    st, tm = engine.last_stats(), engine.last_timing()
    assert st["n_hits"] == cluster_pairs == 6_590_299 and st["n_launches"] == 1
    assert tm["hits_filtered"] > 6_000_000
    seen = np.zeros(len(d), bool)
    for g in got[::37]:
        t = g[-1]
        assert all(orc.hamming(w[t], w[m]) <= 350 for m in g[:-1]) and g[:-1] == sorted(g[:-1]) and all(m > t for m in g[:-1])
    for g in got:
        assert not seen[g].any()
        seen[g] = True
    assert seen.sum() == 100_000 and len(got) == n_clusters
    targets = [g[-1] for g in got]
    assert targets == sorted(targets, reverse=True)  # ret.reverse(), search_algorithm.rs:167
    w4, d4, n4, pairs4 = bench.make_dup_heavy(400_000)
    got4 = engine.search_self_sorted(w4, d4, 350)
    assert engine.last_stats()["n_hits"] == pairs4 and engine.last_timing()["hits_filtered"] > 0
    assert got4 == orc.search_self_sorted(w4, d4, 350) and len(got4) == n4
