"""The fused small-frame letterbox kernel (csrc/dct_hash.hip: letterbox_resize_dct_hash_small_kernel; frames of at most 64 x 64 - detect,
crop, resize, DCT and hash of a clip in one persistent workgroup) against the oracle's cropdetect_letterbox + crop + from_frames
(video_frames_gray.rs:38-128,201-210; video_hash_builder.rs:188-204), clip by clip: boxes AND whole hash words.

What the other letterbox tests do not reach: every width / height from 1 to 64 (widths off a multiple of 4 and of 16, frames of a single
row or column, 1 x 1), batches long enough that every persistent workgroup loops many times, the clips near the buffer's end (which leave the
fused launch for the careful route: 1 x 1 frames make that FOUR clips), strip contents on both sides of each of the strip test's three
verdicts (range accept / coarse-bin reject / histogram), probe frames that disagree, uniform probe frames (converging edges -> no crop)."""
import numpy as np
import pytest
import torch

from oracle import vdf_oracle as orc

pytestmark = pytest.mark.gpu


def make_clips(rng, n, w, h):
    """n clips of 16 x h x w with a mix of bar shapes and strip contents."""
    kind = rng.integers(0, 4, size=n)  # 0 iid noise, 1 smooth, 2 few grey levels (histogram cases), 3 near-flat with outliers
    fr = rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for c in np.nonzero(kind == 1)[0]:
        fr[c] = np.clip(100 + 60 * np.sin(xx / max(w, 2) * 4 + c) + 40 * np.cos(yy / max(h, 2) * 3) + rng.normal(0, 3, (16, h, w)), 0, 255)
    for c in np.nonzero(kind == 2)[0]:
        levels = rng.choice(256, size=int(rng.integers(2, 5)), replace=False)
        fr[c] = levels[rng.integers(0, len(levels), size=(16, h, w))]
    for c in np.nonzero(kind == 3)[0]:
        base = int(rng.integers(0, 240))
        fr[c] = np.clip(base + rng.integers(0, int(rng.integers(8, 40)), size=(16, h, w)), 0, 255)  # ranges on both sides of the tolerance 16
        mask = rng.random((16, h, w)) < rng.choice([0.02, 0.09, 0.11, 0.3])         # outliers on both sides of 10 %
        fr[c][mask] = rng.integers(0, 256, size=int(mask.sum()))
    for c in range(n):
        shape = rng.integers(0, 8)
        t, b = (int(rng.integers(0, max(1, h // 3) + 1)) for _ in range(2))
        l, r = (int(rng.integers(0, max(1, w // 3) + 1)) for _ in range(2))
        val = int(rng.integers(0, 256))
        noise = int(rng.choice([0, 0, 3, 16, 17, 30]))  # clean bars, codec noise inside / at / over the tolerance
        def bar(sl):
            blk = fr[c][sl]
            blk[...] = np.clip(val + (rng.integers(0, noise + 1, size=blk.shape) if noise else 0), 0, 255)
        if shape in (1, 3, 5):
            if t: bar((slice(None), slice(0, t)))
            if b: bar((slice(None), slice(h - b, h)))
        if shape in (2, 3, 5):
            if l: bar((slice(None), slice(None), slice(0, l)))
            if r: bar((slice(None), slice(None), slice(w - r, w)))
        if shape == 4:   # a uniform probe frame (a fade-in): every strip is letterbox, the edges converge, that frame says "no crop"
            fr[c, int(rng.choice([0, 8]))] = val
        if shape == 5:   # the two probes disagree: bars only in frame 0 .. 7
            fr[c, 8:] = rng.integers(0, 256, size=(8, h, w))
        if shape == 6:   # both probes uniform, different values
            fr[c, 0], fr[c, 8] = val, 255 - val
        if shape == 7 and t + b < h:  # a blemish inside a bar
            if t: bar((slice(None), slice(0, t)))
            fr[c, 0, 0, int(rng.integers(0, w))] = 255 - val
    return fr


SIZES = [(64, 64), (1, 1), (1, 64), (64, 1), (2, 3), (3, 2), (5, 64), (64, 5), (15, 15), (16, 16), (17, 17), (31, 33), (33, 31), (47, 64),
         (48, 36), (50, 50), (62, 64), (63, 63), (64, 40), (40, 64), (13, 57), (57, 13), (16, 64), (64, 16), (36, 20), (20, 36), (60, 44), (7, 7)]


@pytest.mark.parametrize("w,h", SIZES, ids=[f"{w}x{h}" for w, h in SIZES])
def test_fused_kernel_matches_the_oracle_clip_by_clip(w, h):
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(w * 131 + h)
    n = 2600 if w * h >= 1024 else 3400  # > 768 persistent workgroups x 3 iterations; the last clip(s) take the careful route
    fr = make_clips(rng, n, w, h)
    res = [orc.hash_clip_letterbox(c) for c in fr]
    assert all(r[0] == 0 for r in res)
    want_h = np.stack([r[1] for r in res])
    want_c = np.array([r[3] for r in res], np.uint32)
    assert len({tuple(c) for c in want_c}) >= (4 if min(w, h) >= 5 else 1)
    eng = vdf.Engine(0)
    try:
        d = torch.from_numpy(fr).cuda()
        out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
        dcr = torch.full((n, 4), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for rep in range(2):  # the first call builds the tables of this frame size; the second runs with everything resident
            out.zero_(); dcr.fill_(-1); torch.cuda.synchronize()
            eng.hash_frames_letterbox_device(d.data_ptr(), n, 16, w, h, out.data_ptr(), d_crops=dcr.data_ptr())
            torch.cuda.synchronize()
            got_c = dcr.cpu().numpy().astype(np.uint32)
            bad = np.nonzero((got_c != want_c).any(axis=1))[0]
            assert len(bad) == 0, f"{w}x{h} rep {rep}: {len(bad)} boxes differ, first clip {bad[0]}: {got_c[bad[0]]} want {want_c[bad[0]]}"
            got_h = out.cpu().numpy().view(np.uint64)
            bad = np.nonzero((got_h != want_h).any(axis=1))[0]
            assert len(bad) == 0, f"{w}x{h} rep {rep}: {len(bad)} hashes differ, first clip {bad[0]} box {want_c[bad[0]]}"
        # the host-frames entry point (batches, pinned staging) and the one that returns the boxes to the host
        hh, cc = eng.hash_frames_letterbox(fr[:500])
        assert np.array_equal(cc, want_c[:500]) and np.array_equal(hh, want_h[:500])
        out.zero_(); torch.cuda.synchronize()
        crops = eng.hash_frames_letterbox_device(d.data_ptr(), n, 16, w, h, out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(crops, want_c) and np.array_equal(out.cpu().numpy().view(np.uint64), want_h)
    finally:
        eng.close()


def test_overlapping_and_repeated_clips():
    """clip_stride below a clip's size (overlapping windows over one long frame sequence) and clip_stride 0 (one clip n times): every clip
    whose loads could pass the buffer's end takes the careful route - with stride 0 that is all of them."""
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(5)
    w = h = 64
    seq = make_clips(rng, 4, w, h).reshape(64, h, w)  # 64 frames
    seq[:, :6] = 20
    d = torch.from_numpy(seq).cuda()
    eng = vdf.Engine(0)
    try:
        for stride_frames, n in ((1, 49), (3, 17), (0, 9)):
            out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
            dcr = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            eng.hash_frames_letterbox_device(d.data_ptr(), n, 16, w, h, out.data_ptr(), d_crops=dcr.data_ptr(), frame_stride=w * h,
                                             clip_stride=stride_frames * w * h)
            torch.cuda.synchronize()
            for c in range(n):
                rc, hw, _, crop = orc.hash_clip_letterbox(seq[c * stride_frames:c * stride_frames + 16])
                assert rc == 0 and tuple(int(x) for x in dcr[c].cpu()) == crop and np.array_equal(out[c].cpu().numpy().view(np.uint64), hw), (stride_frames, c)
    finally:
        eng.close()
