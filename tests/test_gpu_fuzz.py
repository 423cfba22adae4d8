"""Randomised differential tests: HIP path (both Hamming backends, all resize kernels) vs the CPU oracle."""
import numpy as np
import pytest

import hashgen as hg
from oracle import vdf_oracle as orc

import os

pytestmark = pytest.mark.gpu
# VDF_FUZZ_SEEDS=N widens every case list for a soak run (default sizes keep the suite fast)
_SOAK = int(os.environ.get("VDF_FUZZ_SEEDS", "0"))


def _durations(rng, n):
    kind = rng.integers(0, 5)
    if kind == 0:
        return np.zeros(n, np.uint32)
    if kind == 1:
        return rng.integers(0, 12, size=n).astype(np.uint32)                      # many ties, tiny windows
    if kind == 2:
        return np.floor(np.exp(rng.uniform(0, np.log(2e5), size=n))).astype(np.uint32)
    if kind == 3:
        return rng.integers(4_000_000_000, 2**32, size=n, dtype=np.uint64).astype(np.uint32)  # saturating casts
    return rng.choice(np.array([7, 8, 100, 109, 110, 111, 1000, 1100], np.uint32), size=n)


def _clustered(rng, n):
    """Random hashes with a random cluster structure: centres + members at random radii."""
    w = hg.random_hashes(rng, n)
    n_centres = int(rng.integers(1, max(2, n // 8 + 1)))
    centres = rng.choice(n, size=min(n_centres, n), replace=False)
    for i in range(n):
        if rng.random() < 0.6:
            c = int(rng.choice(centres))
            if c != i:
                bits = np.unpackbits(w[c].view(np.uint8), bitorder="little")
                bits[rng.choice(1024, size=int(rng.integers(0, 420)), replace=False)] ^= 1
                w[i] = np.packbits(bits, bitorder="little").view(np.uint64)
    return w


@pytest.mark.parametrize("seed", range(max(40, _SOAK)))
def test_search_fuzz(engine, seed):
    rng = np.random.default_rng(10_000 + seed)
    n = int(rng.integers(1, 1400))
    w = _clustered(rng, n)
    d = _durations(rng, n)
    w, d, _ = hg.sort_by_duration(w, d)
    tol = int(rng.choice([0, 1, 50, 200, 350, 351, 500, 512, 700, 1024]))
    assert engine.search_self_sorted(w, d, tol) == orc.search_self_sorted(w, d, tol)
    n_ref = int(rng.integers(1, 300))
    rw = np.concatenate([w[rng.choice(n, size=n_ref // 2 + 1)], hg.random_hashes(rng, n_ref)])[:n_ref]
    rd = np.concatenate([d[rng.choice(n, size=n_ref // 2 + 1)], _durations(rng, n_ref)])[:n_ref]
    assert engine.search_refs_sorted(w, d, rw, rd, tol) == orc.search_refs_sorted(w, d, rw, rd, tol)


@pytest.mark.parametrize("seed", range(max(6, _SOAK // 8)))
def test_search_fuzz_with_tiny_hit_buffer(engine, seed):
    rng = np.random.default_rng(20_000 + seed)
    n = int(rng.integers(200, 1200))
    w = _clustered(rng, n)
    d = _durations(rng, n)
    w, d, _ = hg.sort_by_duration(w, d)
    engine.set_hit_capacity(int(rng.integers(1, 400)))
    try:
        assert engine.search_self_sorted(w, d, 400) == orc.search_self_sorted(w, d, 400)
        rw, rd = w[: n // 3], d[: n // 3]
        assert engine.search_refs_sorted(w, d, rw, rd, 400) == orc.search_refs_sorted(w, d, rw, rd, 400)
    finally:
        engine.set_hit_capacity(1 << 24)


@pytest.mark.parametrize("seed", range(max(24, _SOAK // 2)))
def test_hash_fuzz_frame_sizes(seed):
    """Random frame sizes from 1 x 1 (up-scaling) to a few hundred pixels, random content statistics, every resize
    kernel that accepts the size."""
    import os

    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(30_000 + seed)
    h = int(rng.integers(1, 40)) if seed % 3 == 0 else int(rng.integers(1, 260))
    w = int(rng.integers(1, 40)) if seed % 3 == 1 else int(rng.integers(1, 330))
    kind = rng.integers(0, 3)
    if kind == 0:
        frames = rng.integers(0, 256, size=(3, 17, h, w), dtype=np.uint8)
    elif kind == 1:
        frames = rng.choice(np.array([0, 255], np.uint8), size=(3, 17, h, w))               # saturating content
    else:
        frames = (rng.integers(0, 30, size=(3, 17, 1, 1)) + rng.integers(0, 4, size=(3, 17, h, w)) * 60).astype(np.uint8)
    want, coefs = orc.hash_clips_with_coefs(frames)
    care = np.abs(coefs) >= 1e-6
    wb = np.unpackbits(want.view(np.uint8), bitorder="little").reshape(3, 1024)[:, :1000]
    for mode in (0, 1, 3, 4):
        os.environ["VDF_RESIZE_MODE"] = str(mode)
        try:
            eng = vdf.Engine(0)
        finally:
            os.environ.pop("VDF_RESIZE_MODE", None)
        try:
            got = eng.hash_frames(frames)
        except vdf.VdfError as e:
            assert mode in (3, 4) and e.code == -2, (mode, h, w, str(e))  # forced MFMA mode on tables that need the fallback
            continue
        finally:
            eng.close()
        gb = np.unpackbits(got.view(np.uint8), bitorder="little").reshape(3, 1024)[:, :1000]
        assert not (gb != wb).any(), (mode, h, w)


@pytest.mark.parametrize("seed", range(max(10, _SOAK // 20)))
def test_hash_fuzz_large_frames(seed):
    """Decoder-sized frames (129..900 rows, up to 1500 wide, odd sizes included): the whole-line per-frame kernel (auto and
    forced), the linear-stream kernel and the scalar kernel against the oracle."""
    import os

    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(70_000 + seed)
    h, w = int(rng.integers(129, 900)), int(rng.integers(17, 1500))
    if seed % 4 == 0:
        w = (w + 127) // 128 * 128  # line-aligned pitch
    elif seed % 4 == 1:
        w = max(256, w // 16 * 16)  # multiple of 16: the linear-stream kernel unless it is also a multiple of 128
    frames = rng.integers(0, 256, size=(2, 16, h, w), dtype=np.uint8)
    if seed % 3 == 0:
        frames = (frames // 16 * 16 + rng.integers(0, 3, size=(2, 16, 1, 1))).astype(np.uint8)  # banded content
    want, coefs = orc.hash_clips_with_coefs(frames)
    care = np.abs(coefs) >= 1e-6
    wb = np.unpackbits(want.view(np.uint8), bitorder="little").reshape(2, 1024)[:, :1000]
    for mode in (0, 1, 4, 5):
        os.environ["VDF_RESIZE_MODE"] = str(mode)
        try:
            eng = vdf.Engine(0)
        finally:
            os.environ.pop("VDF_RESIZE_MODE", None)
        try:
            got = eng.hash_frames(frames)
        except vdf.VdfError as e:
            assert mode in (4, 5) and e.code == -2, (mode, h, w, str(e))
            continue
        finally:
            eng.close()
        gb = np.unpackbits(got.view(np.uint8), bitorder="little").reshape(2, 1024)[:, :1000]
        assert not (gb != wb).any(), (mode, h, w)


@pytest.mark.parametrize("seed", range(max(30, _SOAK // 4)))
def test_cropped_hash_fuzz(seed):
    """vdf_hash_frames_u8_cropped_device with random frame sizes (pitches of every alignment class, 64..1100 columns, 129..420
    rows) and random per-clip crop boxes (some clips uncropped, some full-width, some starting off a dword) through the
    default kernels and the forced cropped-stream / whole-line ones: hashes equal to the oracle's on the cropped copies."""
    import os

    import torch

    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(90_000 + seed)
    w = int(rng.choice([64, 96, 176, 200, 240, 272, 320, 426, 480, 500, 600, 640, 720, 854, 960, 1024, 1100]))
    if seed % 5 == 0:
        w = int(rng.integers(64, 1100))
    h = int(rng.integers(129, 420))
    if (w * h) % 16 and seed % 2:
        h += (16 - h % 16) % 16  # most sizes whose frames end on 16-byte boundaries (the stream kernels' condition), some that do not
    n = 10
    frames = rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
    crops = np.zeros((n, 4), np.uint32)
    for c in range(n):
        kind = rng.integers(0, 4)
        if kind == 0:
            continue
        t, b = int(rng.integers(0, h // 4)), int(rng.integers(0, h // 4))
        l, r = (0, 0) if kind == 1 else (int(rng.integers(0, w // 5)), int(rng.integers(0, w // 5)))
        crops[c] = (l, r, t, b)
    want = np.stack([orc.hash_clip(np.ascontiguousarray(frames[c][:, crops[c][2]:h - crops[c][3], crops[c][0]:w - crops[c][1]]))[1]
                     for c in range(n)])
    d = torch.from_numpy(frames).cuda()
    for mode in (0, 5, 4):
        os.environ["VDF_RESIZE_MODE"] = str(mode)
        try:
            eng = vdf.Engine(0)
        finally:
            os.environ.pop("VDF_RESIZE_MODE", None)
        try:
            out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            eng.hash_frames_cropped_device(d.data_ptr(), n, 16, w, h, crops, out.data_ptr())
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint64)
        finally:
            eng.close()
        assert np.array_equal(got, want), (mode, h, w, np.nonzero((got != want).any(axis=1))[0].tolist(), crops.tolist())


def _bar(rng, shape, axis_len_first):
    """One bar of a random style: what a bar can look like after an encoder - and what only looks like one."""
    style = int(rng.integers(0, 8))
    base = int(rng.choice([0, 1, 16, 17, 128, 235, 250, 255])) if rng.random() < 0.5 else int(rng.integers(0, 256))
    if style == 0:    # one value
        return np.full(shape, base, np.uint8)
    if style == 1:    # one value per strip (gradient along the walk) or per position inside the strips
        ramp = (base + np.arange(shape[axis_len_first], dtype=np.int64) * int(rng.integers(1, 4))) % 256
        idx = [None] * len(shape)
        idx[axis_len_first] = slice(None)
        return np.broadcast_to(ramp.astype(np.uint8)[tuple(idx)], shape).copy()
    if style in (2, 3):  # noise inside a window of k levels: k <= 17 is the range rule's ground (16 accepted for sure, 17 by the count or not)
        k = int(rng.choice([2, 4, 8, 16, 17, 18, 24, 40]))
        lo = min(base, 256 - k)
        return rng.integers(lo, lo + k, size=shape).astype(np.uint8)
    if style in (4, 5):  # a clean or slightly noisy bar with outliers (logo pixels, subtitles): below / around / above the 10 % mark
        k = int(rng.choice([1, 3]))
        lo = min(base, 256 - k)
        out = rng.integers(lo, lo + k, size=shape).astype(np.uint8)
        p = float(rng.choice([0.001, 0.05, 0.095, 0.105, 0.2]))
        mask = rng.random(shape) < p
        out[mask] = ((out[mask].astype(np.int64) + 100) % 256).astype(np.uint8)
        return out
    if style == 6:    # two far-apart values: never a bar
        return rng.choice(np.array([base, (base + 90) % 256], np.uint8), size=shape)
    return rng.integers(0, 256, size=shape).astype(np.uint8)  # picture


@pytest.mark.parametrize("seed", range(max(24, _SOAK // 2)))
def test_letterbox_detect_fuzz(seed):
    """vdf_cropdetect_letterbox_device against the oracle on random frames with random bars of random styles on all four edges (one value, a
    ramp, noise inside windows of 2 ... 40 levels, outliers around the 10 % mark, two-valued, picture), dark / flat / busy pictures (a dark picture
    continues a dark bar: the walk goes on into it), frames 0 and 8 differing, every alignment class of width and buffer base, all three
    side-walk kernels (H < 256, < 512, >= 512: with the aligned probes) - the round-5 shortcuts (range rule, probes) decide nothing the count would not."""
    import torch

    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(70_000 + seed)
    hk = seed % 3
    h = int(rng.integers(17, 256)) if hk == 0 else int(rng.integers(256, 512)) if hk == 1 else int(rng.integers(512, 900))
    w = int(rng.choice([64, 128, 192, 256, 320, 384, 512, 640, 704, 768, 896, 1024, 1280])) if seed % 2 else int(rng.integers(40, 1100))
    n = 12 if h * w < 400_000 else 6
    frames = np.empty((n, 16, h, w), np.uint8)
    for c in range(n):
        pic = int(rng.integers(0, 4))
        if pic == 0:
            f = rng.integers(0, 256, size=(16, h, w), dtype=np.uint8)
        elif pic == 1:    # dark picture: within the tolerance of a black bar
            f = rng.integers(0, 30, size=(16, h, w), dtype=np.uint8)
        elif pic == 2:    # flat picture with a little texture
            f = (int(rng.integers(0, 236)) + rng.integers(0, 20, size=(16, h, w))).astype(np.uint8)
        else:             # smooth gradient both ways
            f = ((np.arange(h)[:, None] * 255 // h + np.arange(w)[None, :] * 255 // w) // 2).astype(np.uint8)[None].repeat(16, 0)
        for fr in (0, 8):
            if fr == 8 and rng.random() < 0.5:
                f[8] = f[0]   # the same bars in both probed frames (the usual case); else frame 8 gets its own below
                continue
            t, b = (int(rng.integers(0, h // 3)) if rng.random() < 0.6 else 0 for _ in range(2))
            l, r = (int(rng.integers(0, w // 3)) if rng.random() < 0.6 else 0 for _ in range(2))
            if l: f[fr, :, :l] = _bar(rng, (h, l), 1)
            if r: f[fr, :, w - r:] = _bar(rng, (h, r), 1)
            if t: f[fr, :t, :] = _bar(rng, (t, w), 0)
            if b: f[fr, h - b:, :] = _bar(rng, (b, w), 0)
        if rng.random() < 0.08:
            f[0] = int(rng.integers(0, 256))  # a uniform probe frame: the walkers meet
        frames[c] = f
    offset = int(rng.choice([0, 0, 16, 64, 128, 1, 3]))
    buf = torch.zeros(frames.size + 256, dtype=torch.uint8, device="cuda")
    buf[offset:offset + frames.size] = torch.from_numpy(frames).cuda().reshape(-1)
    crops = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    eng = vdf.Engine(0)
    try:
        torch.cuda.synchronize()
        eng.cropdetect_letterbox_device(buf.data_ptr() + offset, n, 16, w, h, crops.data_ptr())
        torch.cuda.synchronize()
    finally:
        eng.close()
    got = crops.cpu().numpy().astype(np.uint32)
    want = np.array([orc.cropdetect_letterbox(c) for c in frames], np.uint32)
    assert np.array_equal(got, want), (h, w, offset, [(i, g, x) for i, (g, x) in enumerate(zip(got.tolist(), want.tolist())) if g != x])


@pytest.mark.parametrize("seed", range(max(12, _SOAK // 10)))
def test_hash_fuzz_wide_frames(seed):
    """The wide kernels numerically fuzzed: 1500..4200 columns, 129..2200 rows, through the device entry point with random
    base alignment and frame / clip strides (packed and aligned, 16-byte-aligned padding, odd padding off a 3..15-byte base):
    auto dispatch, the whole-line kernel (4), the linear-stream kernel wherever it applies (5: up to 1984 columns) and its
    K-split form wherever it applies (6: 2048..4096 columns, a multiple of 16) - whole hash words against the oracle."""
    import torch

    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(110_000 + seed)
    if seed % 3 == 0:
        w = int(rng.integers(94, 257)) * 16          # multiple of 16: 1504 .. 4096 (stream / K-split eligible)
    elif seed % 3 == 1:
        w = int(rng.integers(1500, 4201))            # anything, odd widths included (whole-line kernel; > 4096 too)
    else:
        w = int(rng.integers(12, 33)) * 128          # line-aligned pitches 1536 .. 4096
    h = int(rng.integers(129, 2201)) if seed % 4 else int(rng.choice([1080, 1440, 2160, 1152, 1200]))
    n, nf = 2, int(rng.choice([16, 17]))
    frames = rng.integers(0, 256, size=(n, nf, h, w), dtype=np.uint8)
    if seed % 5 == 0:
        frames = (frames // 32 * 32 + rng.integers(0, 3, size=(n, nf, 1, 1))).astype(np.uint8)  # banded content
    want = orc.hash_clips(np.ascontiguousarray(frames[:, :16]))
    layout = (seed // 3) % 3 if seed < 9 else int(rng.integers(0, 3))  # seeds 0..8: every layout meets every width class
    if layout == 0:
        base, pad_f, pad_c = 0, 0, 0
    elif layout == 1:
        base, pad_f, pad_c = 16, 16 * int(rng.integers(0, 9)), 16 * int(rng.integers(0, 9))
        pad_f += (16 - (w * h + pad_f) % 16) % 16  # every frame starts on a 16-byte boundary
    else:
        base, pad_f, pad_c = int(rng.integers(3, 16)), int(rng.integers(1, 200)), int(rng.integers(1, 200))
    fs = w * h + pad_f
    cs = nf * fs + pad_c
    buf = np.full(base + (n - 1) * cs + (nf - 1) * fs + w * h, 0xAB, np.uint8)  # ends with the last byte of the last frame
    for c in range(n):
        for f in range(nf):
            o = base + c * cs + f * fs
            buf[o:o + w * h] = frames[c, f].reshape(-1)
    d_buf = torch.from_numpy(buf).cuda()
    for mode in (0, 4, 5, 6):
        os.environ["VDF_RESIZE_MODE"] = str(mode)
        try:
            eng = vdf.Engine(0)
        finally:
            os.environ.pop("VDF_RESIZE_MODE", None)
        try:
            d_out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            eng.hash_frames_device(d_buf.data_ptr() + base, n, nf, w, h, d_out.data_ptr(), frame_stride=fs, clip_stride=cs)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy().view(np.uint64)
        except vdf.VdfError as e:
            assert e.code == -2, (mode, h, w, str(e))  # a forced MFMA mode on tables that need the scalar fallback
            continue
        finally:
            eng.close()
        assert np.array_equal(got, want), (mode, h, w, layout, base, pad_f, pad_c)


_SOAK_FAMILIES = [  # name, rows, columns, clips, letterboxed
    ("K-split 2048", 300, 2048, 24, False), ("K-split 3840", 200, 3840, 12, False),
    ("stream 422 (shifted)", 240, 422, 40, False), ("stream 480", 270, 480, 40, False), ("stream 500 (re-pitched)", 288, 500, 40, False),
    # one block stream per wave: 4, 5, 6 and 8 waves per workgroup, every addressing mode
    ("wave-stream 1920", 300, 1920, 24, False), ("wave-stream 1792 (re-pitched)", 200, 1792, 24, False), ("wave-stream 1440, 5 waves", 200, 1440, 40, False),
    ("wave-stream 1366 (shifted), 5 waves", 200, 1366, 40, False), ("wave-stream 1536 (re-pitched), 5 waves", 200, 1536, 24, False),
    ("wave-stream 1280 (re-pitched), 6 waves", 360, 1280, 24, False), ("wave-stream 1152, 6 waves", 324, 1152, 32, False),
    ("wave-stream 1001 (shifted), 6 waves", 208, 1001, 32, False), ("wave-stream 640, 8 waves", 360, 640, 40, False),
    ("wave-stream 854 (shifted), 8 waves", 480, 854, 24, False), ("wave-stream 768 (re-pitched), 8 waves", 432, 768, 32, False),
    ("wave-stream 1950 (shifted), 3 waves", 200, 1950, 24, False), ("whole-line 2400 (odd stride)", 131, 2401, 12, False),
    ("whole-line 2000", 200, 2000, 20, False), ("persistent 64", 64, 64, 600, False), ("fused 128", 128, 128, 200, False),
    ("cropped stream 854", 480, 854, 24, True), ("cropped stream 480", 270, 480, 40, True), ("cropped whole-line 1280", 360, 1280, 16, True),
    # top / bottom bars only: the ROWCROP instantiations of the stream kernels (per-clip first row, height, vertical table)
    ("row-cropped stream 1280", 360, 1280, 24, "rows"), ("row-cropped stream 854 (shifted)", 300, 854, 24, "rows"),
    ("row-cropped stream 640", 360, 640, 40, "rows"), ("row-cropped wave-stream 1920", 300, 1920, 24, "rows"),
    ("row-cropped wave-stream 1366 (shifted)", 200, 1366, 32, "rows"), ("row-cropped K-split 3840", 200, 3840, 12, "rows"),    # side bars from a few fixed widths: boxes that share their column range go through the per-wave kernel group by group
    ("box groups 1280", 360, 1280, 32, "boxes"), ("box groups 1001 (shifted)", 300, 1001, 32, "boxes"),
]


@pytest.mark.parametrize("name,h,w,n,letterbox", _SOAK_FAMILIES, ids=[f[0] for f in _SOAK_FAMILIES])
def test_stream_kernels_soak(name, h, w, n, letterbox):
    """Race hunt, bounded: every kernel family of the hash path is launched >= 20 times on FRESH engines (fresh allocations,
    cold tables and caches - the persistent kernels' barrier / LDS-DMA bugs of round 2 showed up in about one launch in ten)
    and every launch must reproduce the oracle's hashes (and the scalar kernel's, VDF_RESIZE_MODE=1)."""
    import vid_dup_finder_lib_amd as vdf

    def fresh(mode):
        os.environ["VDF_RESIZE_MODE"] = str(mode)
        try:
            return vdf.Engine(0)
        finally:
            os.environ.pop("VDF_RESIZE_MODE", None)

    rng = np.random.default_rng(len(name) * 1000 + w)
    frames = rng.integers(40, 220, size=(n, 16, h, w), dtype=np.uint8)
    if letterbox:
        for c in range(n):
            t, b = int(rng.integers(0, h // 5)), int(rng.integers(0, h // 5))
            l, r = (int(rng.integers(0, w // 6)), int(rng.integers(0, w // 6))) if c % 2 and letterbox != "rows" else (0, 0)
            if letterbox == "boxes":
                l, r = [(w // 8, w // 8), (w // 8 + 1, w // 8 + 2), (0, 0)][c % 3]
            if t:
                frames[c, :, :t] = 16
            if b:
                frames[c, :, h - b:] = 16
            if l:
                frames[c, :, :, :l] = 16
            if r:
                frames[c, :, :, w - r:] = 16
        want = np.stack([orc.hash_clip_letterbox(frames[c])[1] for c in range(n)])
        call = lambda e: e.hash_frames_letterbox(frames)[0]  # noqa: E731
    else:
        want = orc.hash_clips(frames)
        call = lambda e: e.hash_frames(frames)  # noqa: E731
    e = fresh(1)
    try:
        assert np.array_equal(call(e), want), "scalar kernel vs oracle"
    finally:
        e.close()
    reps = max(10, _SOAK // 4)
    bad = []
    for rep in range(reps):
        e = fresh(0)
        try:
            for k in range(2):
                got = call(e)
                if not np.array_equal(got, want):
                    bad.append((rep, k, np.nonzero((got != want).any(axis=1))[0].tolist()))
        finally:
            e.close()
    assert not bad, f"{name}: {len(bad)} bad launches of {2 * reps}: {bad[:5]}"
