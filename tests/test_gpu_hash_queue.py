"""concurrent per-clip submitters share batched launches; every submitted clip's hash (and crop box) is
compared with the CPU ORACLE's from_frames / Cropdetect::Letterbox + from_frames for that clip (and, as a second check, with
the batch API of the same library)."""
import threading

import numpy as np
import pytest

from oracle import vdf_oracle as orc

pytestmark = pytest.mark.gpu


def _oracle_letterbox(frames):
    """(hashes [n, 16], crops [n, 4]) of the oracle's crop_video_frames(Letterbox) + from_frames, clip by clip."""
    hs, cs = [], []
    for clip in frames:
        rc, h, _, crop = orc.hash_clip_letterbox(clip)
        assert rc == 0
        hs.append(h)
        cs.append(tuple(int(x) for x in crop))
    return np.stack(hs), cs


@pytest.mark.parametrize("letterbox", [False, True])
def test_many_threads_one_batch(engine, letterbox):
    from vid_dup_finder_lib_amd.engine import HashQueue

    rng = np.random.default_rng(1)
    n, h, w = 96, 48, 64
    frames = rng.integers(40, 220, size=(n, 16, h, w), dtype=np.uint8)
    if letterbox:
        frames[:, :, :5, :] = 16
        frames[::3, :, :, -7:] = 200
    want = engine.hash_frames_letterbox(frames) if letterbox else (engine.hash_frames(frames), np.zeros((n, 4), np.uint32))
    if letterbox:
        o_hash, o_crop = _oracle_letterbox(frames)
        assert any(c != (0, 0, 0, 0) for c in o_crop)  # the bars are really detected
    else:
        o_hash, o_crop = orc.hash_clips(frames), [(0, 0, 0, 0)] * n
    q = HashQueue(engine, w, h, max_batch=32, max_wait_us=20000, letterbox=letterbox)
    got = [None] * n
    errs = []

    def worker(ids):
        try:
            for i in ids:
                got[i] = q.submit(frames[i])
        except Exception as e:  # pragma: no cover
            errs.append(e)

    threads = [threading.Thread(target=worker, args=(range(t, n, 12),)) for t in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errs and all(g is not None for g in got)
    for i in range(n):
        assert np.array_equal(got[i][0], o_hash[i]), i        # the oracle, clip by clip
        assert got[i][1] == o_crop[i], i
        assert np.array_equal(got[i][0], want[0][i])          # and the batch API
        assert got[i][1] == tuple(int(x) for x in want[1][i])
    n_batches, n_clips = q.stats()
    assert n_clips == n and n_batches < n  # concurrent callers really were batched together
    q.close()


def test_single_caller_does_not_wait_for_a_full_batch(engine):
    from vid_dup_finder_lib_amd.engine import HashQueue

    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, size=(3, 16, 32, 32), dtype=np.uint8)
    q = HashQueue(engine, 32, 32, max_batch=1024, max_wait_us=1000)
    o_hash = orc.hash_clips(frames)
    for i in range(3):
        hsh, crop = q.submit(frames[i])
        assert np.array_equal(hsh, o_hash[i]) and crop == (0, 0, 0, 0)
        assert np.array_equal(hsh, engine.hash_frames(frames[i:i + 1])[0])
    assert q.stats() == (3, 3)
    with pytest.raises(ValueError):
        q.submit(frames[0][:, :16, :])
    q.close()


def test_full_hd_callers_overlap_batches(engine):
    """64 threads, one 1080p clip each (33 MB per clip: the staging copies dominate), 16 clips per batch: the queue's second
    slot must collect - and launch - while the first batch is still on the GPU (in_flight_max >= 2), and every caller gets
    exactly the batch API's hash."""
    from vid_dup_finder_lib_amd.engine import HashQueue

    rng = np.random.default_rng(3)
    n, h, w = 64, 1080, 1920
    base = rng.integers(0, 256, size=(4, 16, h, w), dtype=np.uint8)
    frames = [np.roll(base[i % 4], shift=i, axis=2) for i in range(n)]  # 64 distinct clips without 2 GB of RNG output
    want = np.concatenate([engine.hash_frames(np.stack(frames[i:i + 8])) for i in range(0, n, 8)])
    o_hash = np.concatenate([orc.hash_clips(np.stack(frames[i:i + 8])) for i in range(0, n, 8)])  # ~4 s of host work
    q = HashQueue(engine, w, h, max_batch=16, max_wait_us=200000)
    got = [None] * n
    errs = []
    start = threading.Barrier(n)

    def worker(i):
        try:
            start.wait()
            got[i] = q.submit(frames[i])
        except Exception as e:  # pragma: no cover
            errs.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs and all(g is not None for g in got)
    for i in range(n):
        assert np.array_equal(got[i][0], o_hash[i]), i
        assert np.array_equal(got[i][0], want[i]), i
    n_batches, n_clips = q.stats()
    assert n_clips == n and n_batches <= n // 4
    assert q.in_flight_max() >= 2
    q.close()


def test_queue_on_a_multi_gpu_context_spreads_its_slots():
    """A queue created on a multi-GPU context gets two slots per listed device (here: device 0 twice = 4 slots)."""
    import vid_dup_finder_lib_amd as vdf
    from vid_dup_finder_lib_amd.engine import HashQueue

    eng = vdf.Engine(devices=[0, 0])
    try:
        rng = np.random.default_rng(5)
        n = 40
        frames = rng.integers(0, 256, size=(n, 16, 72, 96), dtype=np.uint8)
        want = orc.hash_clips(frames)
        assert np.array_equal(eng.hash_frames(frames), want)
        q = HashQueue(eng, 96, 72, max_batch=4, max_wait_us=50000)
        got = [None] * n
        start = threading.Barrier(n)

        def worker(i):
            start.wait()
            got[i] = q.submit(frames[i])

        ts = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert all(g is not None and np.array_equal(g[0], want[i]) for i, g in enumerate(got))
        assert q.stats()[1] == n
        q.close()
    finally:
        eng.close()


@pytest.mark.parametrize("slots", ["1", "4"])
def test_queue_slots_knob_and_many_more_callers_than_a_batch(engine, monkeypatch, slots):
    """VDF_QUEUE_SLOTS (read when a queue is made): one slot (every batch waits for the one before) and four; 48 callers against
    batches of 4, so most of them sleep for a free slot at any time (the wake-ups of round 6: a reopening slot wakes as many
    sleepers as it has room for) - every clip against the oracle, letterbox boxes included, and a large frame size that crosses
    the link in several pieces."""
    from vid_dup_finder_lib_amd.engine import HashQueue

    monkeypatch.setenv("VDF_QUEUE_SLOTS", slots)
    rng = np.random.default_rng(int(slots))
    for (h, w, n, letterbox) in ((40, 56, 240, True), (360, 640, 48, False)):
        frames = rng.integers(40, 220, size=(n, 16, h, w), dtype=np.uint8)
        if letterbox:
            frames[::2, :, :6, :] = 20
            frames[1::3, :, :, :9] = 200
            o_hash, o_crop = _oracle_letterbox(frames)
        else:
            o_hash, o_crop = orc.hash_clips(frames), [(0, 0, 0, 0)] * n
        q = HashQueue(engine, w, h, max_batch=4, max_wait_us=300, letterbox=letterbox)
        got, errs = [None] * n, []

        def worker(ids):
            try:
                for i in ids:
                    got[i] = q.submit(frames[i])
            except Exception as e:  # pragma: no cover
                errs.append(e)

        threads = [threading.Thread(target=worker, args=(range(t, n, 48),)) for t in range(48)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=180)
        assert not errs and all(g is not None for g in got)
        for i in range(n):
            assert np.array_equal(got[i][0], o_hash[i]) and got[i][1] == o_crop[i], (h, w, i)
        n_batches, n_clips = q.stats()
        assert n_clips == n and n_batches >= n // 4
        assert q.in_flight_max() <= int(slots)
        q.close()
