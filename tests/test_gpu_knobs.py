"""Every VDF_* switch the library reads (measurement and test knobs, csrc/api.cpp: create_single; multi.cpp) changes HOW a result is
computed, never the result: a fresh engine per setting runs a small pass over both halves of the hot path - host-frame hashing (packed and
strided), 64 x 64 / 640-wide / letterboxed clips, search() and search_with_references() - against the oracle.  A knob that ships without a
test is a wrong-answer risk nobody would notice; tests/test_knob_coverage.py (CPU) checks that every getenv in csrc/ appears in a test."""
import os

import numpy as np
import pytest

import hashgen as hg
from oracle import vdf_oracle as orc

pytestmark = pytest.mark.gpu

# (environment, backend the knob belongs to)
KNOBS = [
    ({}, "mfma"),
    ({"VDF_CHUNK_COLS": "192"}, "valu"),
    ({"VDF_ROWS_PER_LANE": "1"}, "valu"),
    ({"VDF_ROWS_PER_LANE": "4"}, "valu"),
    ({"VDF_COPY_THREADS": "3", "VDF_HOST_DIRECT": "0"}, "mfma"),
    ({"VDF_HOST_CHUNK_MB": "1", "VDF_HOST_DIRECT": "0"}, "mfma"),
    ({"VDF_HASH_NO_PERSISTENT": "1"}, "mfma"),
    ({"VDF_HASH_WGS_PER_CU": "2"}, "mfma"),
    ({"VDF_HASH_WGS_PER_CU": "4"}, "mfma"),
    ({"VDF_LB_NC16": "1"}, "mfma"),
    ({"VDF_MFMA_CHUNK_COLS": "4096"}, "mfma"),
    ({"VDF_MFMA_CHUNK_COLS": "512"}, "mfma"),
    ({"VDF_MFMA_GROUP": "3"}, "mfma"),
    ({"VDF_MFMA_MIN_WGS": "64"}, "mfma"),
    ({"VDF_MFMA_REFS_ROWS": "512"}, "mfma"),
    ({"VDF_MFMA_SELF_ROWS": "256"}, "mfma"),
    ({"VDF_MFMA_PRUNE_STEP": "14"}, "mfma"),
    ({"VDF_MFMA_PRUNE_STEP": "16"}, "mfma"),
    ({"VDF_WAVESTREAM_NW": "4"}, "mfma"),
    ({"VDF_WAVESTREAM_NW": "6"}, "mfma"),
    ({"VDF_NO_WAVESTREAM": "1"}, "mfma"),
    ({"VDF_NO_ROWCROP": "1"}, "mfma"),
    ({"VDF_ROWCROP_ALL": "1"}, "mfma"),
    ({"VDF_NO_BOXSTREAM": "1"}, "mfma"),
    ({"VDF_NO_SMALLCROP": "1"}, "mfma"),
    ({"VDF_NO_LB_FUSED": "1"}, "mfma"),
    ({"VDF_LB_HOST_PLAN": "1"}, "mfma"),
    ({"VDF_NO_DEVICE_PATH_ORDER": "1"}, "mfma"),  # exercised by tests/test_gpu_path_order.py (the cache route on both roads)
    ({"VDF_NO_HIT_FILTER": "1"}, "mfma"),
    ({"VDF_CAND_CAPACITY": "64"}, "mfma"),
    ({"VDF_SPIN_WAIT": "1"}, "mfma"),        # the host hashing path's waits spin (round 6: they sleep after a short poll)
    ({"VDF_NO_LINK_TURNS": "1"}, "mfma"),    # bulk host-to-device transfers without the per-GPU turn (hash_host.cpp)
    ({"VDF_RESIZE_MODE": "4"}, "mfma"),
]


@pytest.fixture(scope="module")
def workload():
    rng = np.random.default_rng(404)
    w = {}
    w["small"] = rng.integers(0, 256, size=(40, 16, 64, 64), dtype=np.uint8)
    w["small_want"] = orc.hash_clips(w["small"])
    strided = rng.integers(0, 256, size=(6, 19, 48, 80), dtype=np.uint8)  # 19 frames per clip: only the first 16 count, strided input
    w["strided"] = strided
    w["strided_want"] = orc.hash_clips(strided[:, :16])
    wide = (rng.integers(0, 256, size=(5, 16, 360, 640), dtype=np.uint8) // 3 + 40).astype(np.uint8)
    w["wide"] = wide
    w["wide_want"] = orc.hash_clips(wide)
    # letterboxed 720 x 1280: top / bottom bars, side bars (the side walk's 32-strip form needs 512 rows), none
    lb = (rng.integers(60, 200, size=(12, 16, 720, 1280), dtype=np.uint8))
    for c in range(12):
        if c % 3 == 0:
            lb[c, :, :90] = 3
            lb[c, :, -88:] = 4
        elif c % 3 == 1:
            lb[c, :, :, :160] = 2
            lb[c, :, :, -160:] = 2
    w["lb"] = lb
    res = [orc.hash_clip_letterbox(c) for c in lb]  # (rc, hash, coefs, crop) per clip
    assert all(r[0] == 0 for r in res)
    w["lb_want"] = np.stack([r[1] for r in res])
    w["lb_crops"] = np.array([r[3] for r in res], np.uint32)
    assert {tuple(c) for c in w["lb_crops"]} == {(0, 0, 90, 88), (160, 160, 0, 0), (0, 0, 0, 0)}  # the three shapes are what the oracle sees
    # letterboxed small frames (64 x 64 and 90 x 160): one workgroup per clip since round 5 (VDF_NO_SMALLCROP: one per frame)
    for key, (hh, ww) in (("lbs", (64, 64)), ("lbm", (90, 160))):
        f = rng.integers(60, 200, size=(20, 16, hh, ww), dtype=np.uint8)
        f[::2, :, :hh // 8] = 16
        f[::2, :, -(hh // 7):] = 18
        f[1::4, :, :, :ww // 9] = 15
        res = [orc.hash_clip_letterbox(c) for c in f]
        assert all(r[0] == 0 for r in res) and any(r[3] != (0, 0, 0, 0) for r in res)
        w[key], w[key + "_want"], w[key + "_crops"] = f, np.stack([r[1] for r in res]), np.array([r[3] for r in res], np.uint32)
    words, dur = hg.planted_set(rng, 3000, n_clusters=60, durations="windowed")
    order = np.argsort(dur, kind="stable")
    w["words"], w["dur"] = words[order], dur[order]
    w["self_want"] = {t: orc.search_self_sorted(w["words"], w["dur"], t) for t in (350, 400)}
    w["refs"] = rng.permutation(3000)[:300]
    w["refs_want"] = orc.search_refs_sorted(w["words"], w["dur"], w["words"][w["refs"]], w["dur"][w["refs"]], 350)
    # a dense cluster: the replay filter and the suspect-queue overflow protocol have work
    dense = hg.random_hashes(rng, 2000)
    for i in range(1, 600):
        dense[i] = dense[0]
        dense[i, int(rng.integers(16))] ^= np.uint64(1) << np.uint64(int(rng.integers(40)))
    w["dense"], w["dense_dur"] = dense, np.full(2000, 77, np.uint32)
    w["dense_want"] = orc.search_self_sorted(dense, w["dense_dur"], 350)
    return w


@pytest.mark.parametrize("env,backend", KNOBS, ids=[",".join(f"{k}={v}" for k, v in e.items()) or "defaults" for e, _ in KNOBS])
def test_a_knob_never_changes_a_result(env, backend, workload, monkeypatch):
    import vid_dup_finder_lib_amd as vdf

    for k in [k for k in os.environ if k.startswith("VDF_")]:
        monkeypatch.delenv(k)
    monkeypatch.setenv("VDF_SEARCH_BACKEND", backend)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = workload
    eng = vdf.Engine(0)  # the switches are read when the context is made
    try:
        assert np.array_equal(eng.hash_frames(w["small"]), w["small_want"])
        assert np.array_equal(eng.hash_frames(w["strided"]), w["strided_want"])
        assert np.array_equal(eng.hash_frames(w["wide"]), w["wide_want"])
        for key in ("lb", "lbs", "lbm"):
            got, crops = eng.hash_frames_letterbox(w[key])
            assert np.array_equal(crops, w[key + "_crops"]) and np.array_equal(got, w[key + "_want"]), key
        for tol in (350, 400):
            assert eng.search_self_sorted(w["words"], w["dur"], tol) == w["self_want"][tol]
        assert eng.search_refs_sorted(w["words"], w["dur"], w["words"][w["refs"]], w["dur"][w["refs"]], 350) == w["refs_want"]
        eng.set_hit_capacity(1 << 16)  # the dense cluster's 180 k pairs do not fit: overflow protocol too
        assert eng.search_self_sorted(w["dense"], w["dense_dur"], 350) == w["dense_want"]
    finally:
        eng.close()


def test_force_rccl_on_a_one_device_multi_context(workload, monkeypatch):
    """VDF_FORCE_RCCL (multi.cpp) is read when a multi-GPU context is made: a world of one goes through librccl."""
    import torch

    import vid_dup_finder_lib_amd as vdf

    monkeypatch.setenv("VDF_FORCE_RCCL", "1")
    w = workload
    eng = vdf.Engine(devices=[0])
    try:
        h = torch.from_numpy(w["words"].view(np.int64)).cuda()
        d = torch.from_numpy(w["dur"].astype(np.int32)).cuda()
        torch.cuda.synchronize()
        got = eng.search_self_shards([h.data_ptr()], [d.data_ptr()], [len(w["dur"])], 350)
        assert got == w["self_want"][350]
    finally:
        eng.close()
