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
    assert st["n_hits"] >= cluster_pairs  # the device emits the whole thresholded adjacency, the replay consumes it


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
