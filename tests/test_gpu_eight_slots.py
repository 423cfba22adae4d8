"""G = 8 before the driver's 8-GPU node does it: an eight-slot context on ONE GPU (device list {0} x 8) runs exactly the code an
eight-GPU node runs - eight worker threads (multi.cpp), row tiles dealt round-robin over eight shards, the eight-way merge of the
sorted per-shard hit lists and the eight-way bitmap exchange of the replay filter (api.cpp: search_self_resident; LocalExchange) -
with plain device copies where the node has RCCL over xGMI.  What this cannot show is the scaling CURVE (one GPU's throughput is
shared by the slots); DESIGN.md section 5 says so.  Partition being replaced: search_algorithm.rs:81-171 (one thread, one pass)."""
import os
import sys

import numpy as np
import pytest
import torch

import hashgen as hg
from oracle import vdf_oracle as orc

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.fixture(scope="module")
def eight():
    import vid_dup_finder_lib_amd as vdf

    for k in [k for k in os.environ if k.startswith("VDF_")]:
        del os.environ[k]
    eng = vdf.Engine(devices=[0] * 8)
    yield eng
    eng.close()


def _shards(w, d, cuts):
    tw = [torch.from_numpy(w[a:b].view(np.int64).copy()).cuda() for a, b in zip(cuts[:-1], cuts[1:])]
    td = [torch.from_numpy(d[a:b].view(np.int32).copy()).cuda() for a, b in zip(cuts[:-1], cuts[1:])]
    torch.cuda.synchronize()
    return tw, td


def _ptrs(ts):
    return [t.data_ptr() if t.numel() else 0 for t in ts]


def test_ten_million_on_eight_slots_equals_one_device(eight):
    """BASELINE configs[3] at full size (10 M hashes, 5e13 pairs) sharded 8 ways: same groups as the single-device run, ONE launch,
    and the round-robin deal of row tiles gives every slot the same share of the admitted pairs to within 0.2 %."""
    import vid_dup_finder_lib_amd as vdf

    n = 10_000_000
    rng = np.random.default_rng(20250614)
    words = hg.random_hashes(rng, n)
    planted = 0
    for s in range(0, n - 4, 99_991):
        bits = np.unpackbits(words[s].view(np.uint8), bitorder="little")
        bits[rng.choice(1000, size=int(rng.integers(0, 341)), replace=False)] ^= 1
        words[s + 1] = np.packbits(bits, bitorder="little").view(np.uint64)
        planted += 1
    dur = np.zeros(n, np.uint32)
    cuts = [n * k // 8 for k in range(9)]
    tw, td = _shards(words, dur, cuts)
    got = eight.search_self_shards(_ptrs(tw), _ptrs(td), [len(t) for t in td], 350)
    st = eight.last_stats()
    assert st["n_launches"] == 1 and st["pairs"] == n * (n - 1) // 2
    per = [eight.device_stats(k)["pairs"] for k in range(8)]
    assert sum(per) == st["pairs"] and max(per) / min(per) < 1.002, per
    del tw, td
    one = vdf.Engine(0)
    try:
        d_w = torch.from_numpy(words.view(np.int64)).cuda()
        d_d = torch.zeros(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        hits, n_hits, overflow = one.search_self_device(d_w.data_ptr(), d_d.data_ptr(), n, 350)
        assert overflow == 0xFFFFFFFF
    finally:
        one.close()
    from vid_dup_finder_lib_amd import engine as ve

    want = ve.finish_self(ve.replay_self(n, hits))  # the host replay of the single device's adjacency (search_algorithm.rs:141-161)
    assert len(hits) == planted and got == want and len(got) == planted


def test_dup_heavy_on_eight_slots(eight):
    """The duplicate-dense database of bench.py's dup_heavy leg: 200 k hashes against the oracle, 1 M against one device; the eight-way
    bitmap exchange lets every slot drop the rows that cannot become targets - what comes down is < 2 % of the adjacency."""
    import bench
    import vid_dup_finder_lib_amd as vdf

    w, d, n_clusters, cluster_pairs = bench.make_dup_heavy(200_000)
    want = orc.search_self_sorted(w, d, 350)
    assert eight.search_self_sorted(w, d, 350) == want and len(want) == n_clusters
    st, tm = eight.last_stats(), eight.last_timing()
    per = [eight.device_timing(k)["hits_filtered"] for k in range(8)]
    assert st["n_hits"] == cluster_pairs and all(p > 0 for p in per) and sum(per) == tm["hits_filtered"]
    assert st["n_hits"] - tm["hits_filtered"] <= 0.02 * cluster_pairs
    w, d, n_clusters, cluster_pairs = bench.make_dup_heavy(1_000_000)
    one = vdf.Engine(0)
    try:
        want = one.search_self_sorted(w, d, 350)
    finally:
        one.close()
    cuts = [len(d) * k // 8 for k in range(9)]
    tw, td = _shards(w, d, cuts)
    got = eight.search_self_shards(_ptrs(tw), _ptrs(td), [len(t) for t in td], 350)
    assert got == want and len(got) == n_clusters
    st, tm = eight.last_stats(), eight.last_timing()
    assert st["n_hits"] == cluster_pairs and st["n_launches"] == 1
    assert (st["n_hits"] - tm["hits_filtered"]) / st["n_hits"] <= 0.02


def test_references_over_eight_uneven_shards_one_empty(eight):
    """search_with_references with candidates AND references in eight uneven shards, an empty one among each (a GPU that hashed
    nothing): reference groups in reference input order, exactly the oracle's."""
    rng = np.random.default_rng(808)
    n = 40_000
    words, dur = hg.planted_set(rng, n, n_clusters=300, max_copies=5, durations="windowed")
    w, d, _ = hg.sort_by_duration(words, dur)
    sizes = [9000, 0, 3000, 7001, 999, 12000, 1, 7999]
    assert sum(sizes) == n
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    tw, td = _shards(w, d, cuts)
    pick = rng.choice(n, size=1000, replace=False)
    rw, rd = w[pick].copy(), d[pick].copy()
    rsizes = [0, 300, 5, 200, 95, 1, 399, 0]
    rcut = np.concatenate([[0], np.cumsum(rsizes)])
    trw, trd = _shards(rw, rd, rcut)
    got = eight.search_refs_shards(_ptrs(tw), _ptrs(td), sizes, _ptrs(trw), _ptrs(trd), rsizes, 350)
    assert got == orc.search_refs_sorted(w, d, rw, rd, 350)
    assert eight.search_self_shards(_ptrs(tw), _ptrs(td), sizes, 350) == orc.search_self_sorted(w, d, 350)


def test_a_failed_bitmap_exchange_degrades_to_the_unfiltered_list(monkeypatch):
    """the replay filter's exchange is an optimisation - when it cannot be carried out (no librccl on a multi-GPU node, a failed
    collective; here forced with VDF_TEST_EXCHANGE_FAIL) every slot goes on with its unfiltered list and the search returns the same groups."""
    import bench
    import vid_dup_finder_lib_amd as vdf

    w, d, n_clusters, cluster_pairs = bench.make_dup_heavy(100_000)
    want = orc.search_self_sorted(w, d, 350)
    monkeypatch.setenv("VDF_TEST_EXCHANGE_FAIL", "1")
    eng = vdf.Engine(devices=[0, 0, 0, 0])
    try:
        assert eng.search_self_sorted(w, d, 350) == want
        st, tm = eng.last_stats(), eng.last_timing()
        assert st["n_hits"] == cluster_pairs > (1 << 16) and tm["hits_filtered"] == 0
    finally:
        eng.close()
    monkeypatch.delenv("VDF_TEST_EXCHANGE_FAIL")
    eng = vdf.Engine(devices=[0, 0, 0, 0])
    try:
        assert eng.search_self_sorted(w, d, 350) == want and eng.last_timing()["hits_filtered"] > 0
    finally:
        eng.close()


def test_the_root_cause_of_a_failed_slot_is_what_the_caller_reads(eight):
    """when one slot of a sharded launch fails, the others leave with "another device ... failed"; the call reports the status
    and message of the slot that failed on its own.  Unsorted durations fail on every slot alike (the same check), so the message is that one."""
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(3)
    w = hg.random_hashes(rng, 5000)
    d = rng.integers(10, 1000, size=5000).astype(np.uint32)  # NOT sorted
    with pytest.raises(vdf.VdfError) as ei:
        eight.search_self_sorted(w, d, 350)
    assert "another device" not in str(ei.value) or "ascending" in str(ei.value) or "sorted" in str(ei.value), str(ei.value)
