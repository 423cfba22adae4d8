"""The single-process multi-GPU context of the C ABI (vdf_ctx_create_multi): one host thread + stream per listed device
inside the library.  The GPU box has one MI355X, so the device list repeats device 0 ({0, 0}, {0, 0, 0}): every slot is a
full sub-context with its own thread, stream and scratch, the row tiles are dealt round-robin over the slots and the
all-gather of the *_shards calls is replaced by device-to-device copies (RCCL refuses two ranks on one device).
Results must be identical to the single-device context and to the oracle.  VDF_FORCE_RCCL exercises the real
librccl path (dlopen, ncclCommInitAll, ncclAllGather on the library's stream) with a world of one."""
import os

import numpy as np
import pytest

import hashgen as hg
from oracle import vdf_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _groups(offsets, members):
    return [[int(m) for m in members[int(offsets[i]):int(offsets[i + 1])]] for i in range(len(offsets) - 1)]


@pytest.fixture(scope="module", params=["mfma", "valu"])
def engines(request):
    import vid_dup_finder_lib_amd as vdf

    old = os.environ.get("VDF_SEARCH_BACKEND")
    os.environ["VDF_SEARCH_BACKEND"] = request.param
    try:
        one, two, three = vdf.Engine(0), vdf.Engine(devices=[0, 0]), vdf.Engine(devices=[0, 0, 0])
    finally:
        if old is None:
            os.environ.pop("VDF_SEARCH_BACKEND", None)
        else:
            os.environ["VDF_SEARCH_BACKEND"] = old
    yield one, two, three
    for e in (one, two, three):
        e.close()


def test_device_list_is_reported(engines):
    one, two, three = engines
    assert one.n_devices == 1 and two.n_devices == 2 and three.devices == [0, 0, 0]


def test_golden_search_on_multi_context(engines):
    z = np.load(os.path.join(G, "search_golden.npz"))
    w, d = z["hashes"], z["durations"]
    for eng in engines:
        assert eng.search_self_sorted(w, d, 350) == _groups(z["self350_offsets"], z["self350_members"])
        assert eng.search_self_sorted(w, d, 100) == _groups(z["self100_offsets"], z["self100_members"])
        refs = eng.search_refs_sorted(w, d, z["ref_hashes"], z["ref_durations"], 350)
        assert [r for r, _ in refs] == [int(x) for x in z["refs350_index"]]
        assert [m for _, m in refs] == _groups(z["refs350_offsets"], z["refs350_members"])


@pytest.mark.parametrize("n,durations", [(5000, "windowed"), (20000, "zero"), (1537, "windowed")])
def test_planted_sets_match_single_context_and_oracle(engines, n, durations):
    one, two, three = engines
    rng = np.random.default_rng(n)
    words, dur = hg.planted_set(rng, n, n_clusters=n // 50, max_copies=5, durations=durations)
    w, d, _ = hg.sort_by_duration(words, dur)
    want = orc.search_self_sorted(w, d, 350)
    assert one.search_self_sorted(w, d, 350) == want
    for eng in (two, three):
        assert eng.search_self_sorted(w, d, 350) == want
        st = eng.last_stats()
        per = [eng.device_stats(k) for k in range(eng.n_devices)]
        assert st["pairs"] == sum(p["pairs"] for p in per) == one.last_stats()["pairs"]  # the slots partition the triangle
        assert all(p["pairs"] > 0 for p in per)
    pick = rng.choice(n, size=333, replace=False)
    rw, rd = w[pick].copy(), d[pick].copy()
    want_r = orc.search_refs_sorted(w, d, rw, rd, 300)
    for eng in engines:
        assert eng.search_refs_sorted(w, d, rw, rd, 300) == want_r


def test_overflow_protocol_across_slots(engines):
    """All-identical hashes: O(n^2) hits against a 2000-entry buffer per slot -> many rounds, consumption bitmap fed back
    to every slot, still the reference's single group."""
    one, two, three = engines
    n = 3000
    w = np.tile(hg.random_hashes(np.random.default_rng(1), 1), (n, 1))
    d = np.zeros(n, np.uint32)
    want = orc.search_self_sorted(w, d, 0)
    assert len(want) == 1 and len(want[0]) == n
    for eng in (one, two, three):
        eng.set_hit_capacity(2000)
        try:
            assert eng.search_self_sorted(w, d, 0) == want
            assert eng.last_stats()["n_launches"] > 1
        finally:
            eng.set_hit_capacity(1 << 24)
    # clustered: several groups straddle the overflow rows
    rng = np.random.default_rng(2)
    words, dur = hg.planted_set(rng, 4000, n_clusters=30, max_copies=60, max_flips=100, durations="windowed")
    w, d, _ = hg.sort_by_duration(words, dur)
    want = orc.search_self_sorted(w, d, 350)
    for eng in (two, three):
        eng.set_hit_capacity(300)
        try:
            assert eng.search_self_sorted(w, d, 350) == want
        finally:
            eng.set_hit_capacity(1 << 24)


def test_unsorted_input_is_rejected(engines):
    import vid_dup_finder_lib_amd as vdf

    w = hg.random_hashes(np.random.default_rng(3), 100)
    d = np.arange(100, dtype=np.uint32)[::-1].copy()
    for eng in engines:
        with pytest.raises(vdf.VdfError) as ei:
            eng.search_self_sorted(w, d, 350)
        assert ei.value.code == -5
        with pytest.raises(vdf.VdfError):
            eng.search_refs_sorted(w, d, w[:3], d[:3], 350)


def test_hashing_fans_out_over_the_slots(engines):
    one, two, three = engines
    rng = np.random.default_rng(4)
    frames = rng.integers(0, 256, size=(101, 17, 48, 80), dtype=np.uint8)
    want = orc.hash_clips(frames)
    for eng in engines:
        got, dc = eng.hash_frames(frames, want_dontcare=True)
        assert np.array_equal(got, want)
    lb = np.zeros((7, 16, 90, 160), np.uint8)
    lb[:, :, 20:70, :] = rng.integers(30, 256, size=(7, 16, 50, 160), dtype=np.uint8)
    h1, c1 = one.hash_frames_letterbox(lb)
    for eng in (two, three):
        h2, c2 = eng.hash_frames_letterbox(lb)
        assert np.array_equal(h1, h2) and np.array_equal(c1, c2)
    assert (c1[:, 2] == 20).all()


def test_device_pointer_calls_need_a_single_device_context(engines):
    import torch

    import vid_dup_finder_lib_amd as vdf

    _, two, _ = engines
    t = torch.zeros((64, 16), dtype=torch.int64, device="cuda")
    dd = torch.zeros(64, dtype=torch.int32, device="cuda")
    with pytest.raises(vdf.VdfError) as ei:
        two.search_self_device(t.data_ptr(), dd.data_ptr(), 64, 350)
    assert ei.value.code == -5 and "single-device" in str(ei.value)


@pytest.mark.parametrize("sizes", [(2500, 2500), (4000, 1000), (0, 5000), (1700, 1600, 1700)])
def test_shard_calls_match_the_host_calls(engines, sizes):
    """Database shards resident in HBM (as after hashing on the GPUs), uneven and empty shards included."""
    import torch

    one, two, three = engines
    eng = two if len(sizes) == 2 else three
    n = sum(sizes)
    rng = np.random.default_rng(n + len(sizes))
    words, dur = hg.planted_set(rng, n, n_clusters=80, max_copies=5, durations="windowed")
    w, d, _ = hg.sort_by_duration(words, dur)
    want = orc.search_self_sorted(w, d, 350)
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    tw = [torch.from_numpy(w[a:b].view(np.int64).copy()).cuda() for a, b in zip(cuts[:-1], cuts[1:])]
    td = [torch.from_numpy(d[a:b].view(np.int32).copy()).cuda() for a, b in zip(cuts[:-1], cuts[1:])]
    torch.cuda.synchronize()
    got = eng.search_self_shards([t.data_ptr() if t.numel() else 0 for t in tw], [t.data_ptr() if t.numel() else 0 for t in td],
                                 sizes, 350)
    assert got == want
    pick = rng.choice(n, size=120, replace=False)
    rw, rd = w[pick].copy(), d[pick].copy()
    rcut = np.linspace(0, len(rd), len(sizes) + 1).astype(int)
    trw = [torch.from_numpy(rw[a:b].view(np.int64).copy()).cuda() for a, b in zip(rcut[:-1], rcut[1:])]
    trd = [torch.from_numpy(rd[a:b].view(np.int32).copy()).cuda() for a, b in zip(rcut[:-1], rcut[1:])]
    torch.cuda.synchronize()
    refs = eng.search_refs_shards([t.data_ptr() if t.numel() else 0 for t in tw], [t.data_ptr() if t.numel() else 0 for t in td], sizes,
                                  [t.data_ptr() for t in trw], [t.data_ptr() for t in trd], [len(t) for t in trd], 300)
    assert refs == orc.search_refs_sorted(w, d, rw, rd, 300)
    # per-slot hashing of resident clips
    frames = [torch.randint(0, 256, (5 + 3 * k, 16, 64, 64), dtype=torch.uint8, device="cuda") for k in range(len(sizes))]
    outs = [torch.zeros((f.shape[0], 16), dtype=torch.int64, device="cuda") for f in frames]
    torch.cuda.synchronize()
    eng.hash_frames_shards([f.data_ptr() for f in frames], [f.shape[0] for f in frames], 16, 64, 64, [o.data_ptr() for o in outs])
    for f, o in zip(frames, outs):
        assert np.array_equal(o.cpu().numpy().view(np.uint64), orc.hash_clips(f.cpu().numpy()))


def test_real_rccl_path_with_a_world_of_one(monkeypatch):
    """VDF_FORCE_RCCL: a one-device multi context takes the librccl route for the replication (dlopen + ncclCommInitAll +
    grouped ncclAllGather on the library's stream) - the code the 8-GPU node runs, minus the peers."""
    import torch

    import vid_dup_finder_lib_amd as vdf

    monkeypatch.setenv("VDF_FORCE_RCCL", "1")
    eng = vdf.Engine(devices=[0])
    try:
        rng = np.random.default_rng(77)
        words, dur = hg.planted_set(rng, 6000, n_clusters=100, max_copies=4, durations="windowed")
        w, d, _ = hg.sort_by_duration(words, dur)
        tw = torch.from_numpy(w.view(np.int64).copy()).cuda()
        td = torch.from_numpy(d.view(np.int32).copy()).cuda()
        torch.cuda.synchronize()
        assert eng.search_self_shards([tw.data_ptr()], [td.data_ptr()], [len(d)], 350) == orc.search_self_sorted(w, d, 350)
    finally:
        eng.close()
