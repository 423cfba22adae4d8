"""Race hunters in the suite: a bounded form of tools/diff_sweep*.py.  Every case is the FIRST launch of a FRESH
context (cold instruction cache, nothing allocated: the conditions under which round 5's hand-over race showed - 1 - 2 clips in 30 000, in
a first launch only, under 800 green parity tests) on a LARGE batch, and is compared on the device with the same batch through another
kernel family of the same library:

  hash        default dispatch and every forced form (VDF_RESIZE_MODE 3 / 5 / 6, stream / wave-stream knobs) that accepts the size,
              against the whole-line kernels (VDF_RESIZE_MODE=4), over the sizes that select each resize family;
  cropped     random boxes (one box for the batch / a box per clip, full-width and with side bars, boxes of one chunk) through the default
              dispatch, VDF_ROWCROP_ALL and mode 5, against the whole-line cropped kernel;
  letterbox   detect + crop + hash: the fused small-frame kernel, the device-box route (VDF_NO_LB_FUSED) and round 5's host-planned
              route (VDF_LB_HOST_PLAN) against each other, boxes included;
  search      the two Hamming backends (fp4 Gram matrix on the matrix cores / XOR + popcount on the VALU) against each other: same hits.

Parity with the ORACLE is the business of the other -m gpu files; what this one adds is volume at first launch.  >= 300 cases."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES_RUN = {"hash": 0, "cropped": 0, "letterbox": 0, "search": 0}
CLOCK = {}  # "t0": when this file's first case began (pytest imports every file before it runs the first test)


def fresh_engine(env):
    """A new context with exactly these switches (the library reads them when the context is made)."""
    import vid_dup_finder_lib_amd as vdf

    saved = {k: os.environ.pop(k) for k in [k for k in os.environ if k.startswith("VDF_")]}
    os.environ.update(env)
    try:
        return vdf.Engine(0)
    finally:
        for k in env:
            os.environ.pop(k, None)
        os.environ.update(saved)


def first_launch(env, fn):
    import vid_dup_finder_lib_amd as vdf

    CLOCK.setdefault("t0", time.perf_counter())
    eng = fresh_engine(env)
    try:
        out = fn(eng)
        torch.cuda.synchronize()
        return out
    except vdf.VdfError as e:
        return e
    finally:
        eng.close()


def batch(g, w, h, budget_mb):
    n = int(max(48, min(30000, budget_mb * 1_000_000 // (16 * w * h))))
    fr = torch.randint(0, 256, (n, 16, h, w), dtype=torch.uint8, device="cuda", generator=g)
    torch.cuda.synchronize()
    return n, fr


# sizes by the resize family the default dispatch gives them (csrc/resize_dispatch.cpp): persistent one-tile (64 x 64 and
# smaller), fused per-clip (width off a multiple of 16), tiled <= 256 x 128, short-wide streams, chunk / per-wave / K-split streams, whole-line
HASH_SIZES = [(64, 64), (64, 48), (48, 36), (32, 32), (47, 33), (80, 48), (96, 64), (128, 72), (128, 128), (100, 60), (160, 90), (176, 144),
              (256, 128), (256, 144), (224, 126), (320, 180), (426, 240), (480, 270), (640, 360), (854, 480), (1024, 576), (1280, 720),
              (1366, 768), (1920, 1080), (2560, 1440), (3840, 2160), (100, 300), (1920, 64), (720, 576), (1440, 1080)]
HASH_VARIANTS = [("default", {}), ("mode5", {"VDF_RESIZE_MODE": "5"}), ("mode6", {"VDF_RESIZE_MODE": "6"}), ("mode3", {"VDF_RESIZE_MODE": "3"}),
                 ("no_wavestream", {"VDF_NO_WAVESTREAM": "1"}), ("no_persistent", {"VDF_HASH_NO_PERSISTENT": "1"})]


@pytest.mark.parametrize("rep", range(3))
@pytest.mark.parametrize("w,h", HASH_SIZES, ids=[f"{w}x{h}" for w, h in HASH_SIZES])
def test_hash_families_agree_on_a_first_launch(w, h, rep):
    g = torch.Generator(device="cuda")
    g.manual_seed(w * 4099 + h + 1_000_003 * rep)
    n, fr = batch(g, w, h, 1200 if w * h <= 8192 else 300)

    def call(eng):
        out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        eng.hash_frames_device(fr.data_ptr(), n, 16, w, h, out.data_ptr())
        return out

    ref = first_launch({"VDF_RESIZE_MODE": "4"}, call)
    assert torch.is_tensor(ref), ref
    for name, env in HASH_VARIANTS:
        if name == "mode3" and h > 128:
            continue
        out = first_launch(env, call)
        if not torch.is_tensor(out):
            continue  # the forced form refuses this size
        bad = torch.nonzero((out != ref).any(dim=1)).flatten()
        assert len(bad) == 0, f"{w}x{h} n={n} {name}: {len(bad)} clips differ from the whole-line kernels, first {bad[:6].tolist()}"
        CASES_RUN["hash"] += 1


CROP_SIZES = [(64, 48), (128, 96), (256, 128), (426, 240), (640, 360), (720, 576), (854, 480), (1024, 576), (1280, 720), (1366, 768),
              (1920, 1080), (2560, 1440)]


@pytest.mark.parametrize("rep", range(3))
@pytest.mark.parametrize("kind", ["one_box", "per_clip"])
@pytest.mark.parametrize("w,h", CROP_SIZES, ids=[f"{w}x{h}" for w, h in CROP_SIZES])
def test_cropped_dispatch_agrees_on_a_first_launch(w, h, kind, rep):
    rng = np.random.default_rng(w * 31 + h + (kind == "per_clip") + 977 * rep)
    g = torch.Generator(device="cuda")
    g.manual_seed(w * 7 + h + 1_000_003 * rep)
    n, fr = batch(g, w, h, 1000 if w * h <= 8192 else 250)

    def box():
        t, b = (int(rng.integers(0, max(1, h // 2))) for _ in range(2))
        if t + b >= h:
            b = 0
        l, r = (0, 0) if rng.random() < 0.45 else tuple(int(rng.integers(0, max(1, w // 3))) for _ in range(2))
        if rng.random() < 0.35:  # short boxes: few blocks, ONE chunk (where round 5's race lived)
            keep = int(rng.integers(1, 70))
            if h - t - b > keep:
                b = h - t - keep
        return (l, r, t, b)

    crops = np.zeros((n, 4), np.uint32)
    if kind == "one_box":
        crops[:] = box()
    else:
        pool = [box() for _ in range(6)] + [(0, 0, 0, 0)]
        crops[:] = np.array(pool, np.uint32)[rng.integers(0, len(pool), n)]

    def call(eng):
        out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        eng.hash_frames_cropped_device(fr.data_ptr(), n, 16, w, h, crops, out.data_ptr())
        return out

    ref = first_launch({"VDF_RESIZE_MODE": "4", "VDF_NO_ROWCROP": "1", "VDF_NO_SMALLCROP": "1"}, call)
    assert torch.is_tensor(ref), ref
    for name, env in (("default", {}), ("rowcrop_all", {"VDF_ROWCROP_ALL": "1"}), ("mode5", {"VDF_RESIZE_MODE": "5"}), ("no_boxstream", {"VDF_NO_BOXSTREAM": "1"})):
        out = first_launch(env, call)
        if not torch.is_tensor(out):
            continue
        bad = torch.nonzero((out != ref).any(dim=1)).flatten()
        assert len(bad) == 0, f"{w}x{h} n={n} {kind} {name} box0={tuple(int(v) for v in crops[0])}: {len(bad)} clips differ, first {bad[:6].tolist()}"
        CASES_RUN["cropped"] += 1


LB_SIZES = [(64, 64), (48, 36), (64, 40), (160, 90), (256, 128), (640, 360), (1280, 720), (854, 480)]


@pytest.mark.parametrize("pattern", ["none", "top_bottom", "side", "mixed_noisy"])
@pytest.mark.parametrize("w,h", LB_SIZES, ids=[f"{w}x{h}" for w, h in LB_SIZES])
def test_letterbox_routes_agree_on_a_first_launch(w, h, pattern):
    g = torch.Generator(device="cuda")
    g.manual_seed(w * 13 + h * 5 + len(pattern))
    n, fr = batch(g, w, h, 600 if w * h <= 8192 else 200)
    bt, bs = max(1, h // 8), max(1, w // 8)
    if pattern in ("top_bottom", "mixed_noisy"):
        sl = slice(None) if pattern == "top_bottom" else slice(0, None, 3)
        fr[sl, :, :bt] = 16
        fr[sl, :, h - bt:] = 17
    if pattern in ("side", "mixed_noisy"):
        sl = slice(None) if pattern == "side" else slice(1, None, 3)
        fr[sl, :, :, :bs] = 16
        fr[sl, :, :, w - bs:] = 16
    if pattern == "mixed_noisy":  # what a lossy codec leaves of a bar: the strip tests' undecided cases, and one black probe frame in 500
        nz = torch.randint(0, 5, fr.shape, dtype=torch.uint8, device="cuda", generator=g)
        fr = torch.where(fr <= 17, fr + nz, fr)
        fr[::500, 0] = 16
    torch.cuda.synchronize()

    def call(eng):
        out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
        dcr = torch.full((n, 4), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        eng.hash_frames_letterbox_device(fr.data_ptr(), n, 16, w, h, out.data_ptr(), d_crops=dcr.data_ptr())
        torch.cuda.synchronize()  # the call only queued work on the library's stream: torch's stream must not read the results early
        return torch.cat([out, dcr.to(torch.int64)], dim=1)

    ref = first_launch({"VDF_LB_HOST_PLAN": "1", "VDF_NO_SMALLCROP": "1", "VDF_RESIZE_MODE": "4", "VDF_NO_ROWCROP": "1"}, call)
    assert torch.is_tensor(ref), ref
    for name, env in (("default", {}), ("no_fused", {"VDF_NO_LB_FUSED": "1"}), ("host_plan", {"VDF_LB_HOST_PLAN": "1"})):
        out = first_launch(env, call)
        assert torch.is_tensor(out), out
        bad = torch.nonzero((out != ref).any(dim=1)).flatten()
        assert len(bad) == 0, f"{w}x{h} n={n} {pattern} {name}: {len(bad)} clips differ (hash or box), first {bad[:6].tolist()}"
        CASES_RUN["letterbox"] += 1


@pytest.mark.parametrize("case", range(48))
def test_search_backends_agree_on_a_first_launch(case):
    import hashgen as hg

    rng = np.random.default_rng(9000 + case)
    n = int(rng.choice([3000, 20_000, 70_000, 150_000]))
    tol = int(rng.choice([100, 250, 350, 380, 420]))
    words, dur = hg.planted_set(rng, n, n_clusters=max(10, n // 200), max_copies=int(rng.integers(2, 40)), max_flips=min(tol + 30, 500),
                                durations=str(rng.choice(["windowed", "zero"])))
    w, d, _ = hg.sort_by_duration(words, dur)
    d_w = torch.from_numpy(w.view(np.int64)).cuda()
    d_d = torch.from_numpy(d.view(np.int32)).cuda()
    torch.cuda.synchronize()
    res = {}
    for backend in ("mfma", "valu"):
        def call(eng):
            hits, n_hits, overflow = eng.search_self_device(d_w.data_ptr(), d_d.data_ptr(), n, tol, capacity=1 << 23)
            return hits, n_hits, overflow, eng.last_stats()["pairs"]
        res[backend] = first_launch({"VDF_SEARCH_BACKEND": backend}, call)
        assert isinstance(res[backend], tuple), res[backend]
        CASES_RUN["search"] += 1
    a, b = res["mfma"], res["valu"]
    assert a[1] == b[1] and a[2] == b[2] == 0xFFFFFFFF and a[3] == b[3], (a[1:], b[1:])
    assert np.array_equal(a[0], b[0]), f"n={n} tol={tol}: the backends' hit lists differ"


def test_the_sweep_was_big_enough_and_quick_enough():
    """(runs last in file order) at least 300 first-launch cases (about 1100 today), in the time a round's GPU budget can afford on every run.
    On the library before ce37e43 (tools/build_variant_fast.sh race dct_hash.hip -DVDF_ABL_NO_ONE_CHUNK_BARRIER) the sweep failed 2 of 3 runs
    at a third of this volume (64 x 48 through the chunk kernel: 2 - 3 clips of 24 414; gpurun_out/r6l) - and tests/test_isa_barriers.py
    fails on it every time."""
    total = sum(CASES_RUN.values())
    took = time.perf_counter() - CLOCK["t0"]
    print(f"diff sweep: {CASES_RUN}, {total} cases in {took:.0f} s")
    assert total >= 300 and all(v > 0 for v in CASES_RUN.values()), CASES_RUN
    assert took < 90, "the sweep must stay cheap enough to run in every round"
