"""Device letterbox crop detection + cropped hashing vs the oracle, which is pinned by the
reference's own known-answer tests (tests/test_oracle_letterbox.py)."""
import numpy as np
import pytest
import torch

from oracle import vdf_oracle as orc

pytestmark = pytest.mark.gpu


def _bits(words):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little").reshape(len(words), 1024)[:, :1000]


def _letterboxed(rng, n, h, w, max_bar=0.3, smooth=True):
    """Random clips with random black/grey bars (noisy within +-5), some frames differing between 0 and 8."""
    frames = rng.integers(40, 220, size=(n, 16, h, w), dtype=np.uint8)
    if smooth:
        frames = (frames // 4 + 60).astype(np.uint8) + rng.integers(0, 60, size=(n, 16, 1, 1), dtype=np.uint8)
    for c in range(n):
        l, r = (int(rng.integers(0, int(w * max_bar))) for _ in range(2))
        t, b = (int(rng.integers(0, int(h * max_bar))) for _ in range(2))
        base = int(rng.integers(0, 40))
        noise = lambda shape: (base + rng.integers(0, 6, size=shape)).astype(np.uint8)
        if t: frames[c, :, :t, :] = noise((16, t, w))
        if b: frames[c, :, h - b:, :] = noise((16, b, w))
        if l: frames[c, :, :, :l] = noise((16, h, l))
        if r: frames[c, :, :, w - r:] = noise((16, h, r))
        if c % 5 == 0 and t > 2:  # frame 8 has a narrower top bar: the union must take the minimum
            frames[c, 8, t - 2:t, :] = rng.integers(100, 200, size=(2, w), dtype=np.uint8)
        if c % 7 == 0:
            frames[c, 0] = 17  # uniform frame 0: converging edges -> that frame contributes "no crop"
    return frames


@pytest.mark.parametrize("h,w", [(3, 3), (6, 5), (40, 56), (64, 64), (90, 160), (217, 131)])
def test_reference_kats_and_random_frames_match_oracle(engine, h, w):
    rng = np.random.default_rng(h * 1000 + w)
    if (h, w) == (3, 3):  # the reference's own 3x3 cases (tol 16 on the product path)
        cases = [[255] * 9, [0] * 9, [127, 127, 127, 127, 0, 127, 127, 127, 127], [120, 130, 120, 130, 0, 130, 120, 130, 120],
                 [0, 0, 0, 0, 127, 0, 0, 0, 0], [127, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 200, 0, 0, 120, 0, 0, 100],
                 [0, 0, 0, 0, 127, 0, 0, 0, 127]]
        frames = np.stack([np.tile(np.array(c, np.uint8).reshape(1, 3, 3), (16, 1, 1)) for c in cases])
    elif (h, w) == (6, 5):
        pix = [0, 0, 0, 0, 0, 0, 255, 255, 255, 0, 0, 255, 255, 255, 0, 0, 255, 255, 255, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
        frames = np.tile(np.array(pix, np.uint8).reshape(1, 1, 6, 5), (1, 16, 1, 1))
    else:
        frames = _letterboxed(rng, 24, h, w)
    d = torch.from_numpy(frames).cuda()
    crops = torch.zeros((len(frames), 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    engine.cropdetect_letterbox_device(d.data_ptr(), len(frames), 16, w, h, crops.data_ptr())
    torch.cuda.synchronize()
    got = crops.cpu().numpy().astype(np.uint32)
    want = np.array([orc.cropdetect_letterbox(c) for c in frames], np.uint32)
    assert np.array_equal(got, want)
    if (h, w) == (6, 5):
        assert tuple(got[0]) == (1, 1, 1, 2)  # video_frames_gray.rs:444-459


@pytest.fixture(params=[0, 5])
def stream_mode(request, monkeypatch):
    """0 = the defaults (linear-stream cropped kernel for pitches that are not a multiple of 128), 5 = that kernel wherever it fits."""
    monkeypatch.setenv("VDF_RESIZE_MODE", str(request.param))
    return request.param


@pytest.mark.parametrize("h,w", [(40, 56), (64, 64), (90, 160), (217, 131), (120, 256), (270, 480), (300, 200), (136, 333),
                                 (360, 640), (480, 854), (576, 720), (300, 500), (1080, 1920), (333, 1366), (240, 426),
                                 (426, 240), (320, 176), (200, 96)])
def test_letterbox_hash_matches_oracle(h, w, stream_mode):
    """crop_video_frames(Letterbox) + from_frames: same crop, same hash bits (don't-care rule) as hashing the cropped
    copies on the CPU; the device reads the crop box in place.  Frames of 256..1984 columns and more than 128 rows go
    through the linear-stream cropped kernel (per-clip boxes, band tables, every row-start alignment), the rest through
    the whole-line cropped kernel."""
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(7 * h + w)
    frames = _letterboxed(rng, 20 if h * w < 500_000 else 6, h, w)
    engine = vdf.Engine(0)  # a fresh engine: it reads VDF_RESIZE_MODE when it is created
    try:
        hashes, crops, dc = engine.hash_frames_letterbox(frames, want_dontcare=True)
    finally:
        engine.close()
    n_cropped = 0
    for c in range(len(frames)):
        rc, want, coefs, crop = orc.hash_clip_letterbox(frames[c], want_coefs=True)
        assert rc == 0 and tuple(int(x) for x in crops[c]) == crop
        care = np.abs(coefs) >= 1e-6
        assert not ((_bits(hashes[c:c + 1])[0] != _bits(want[None])[0])).any()
        n_cropped += any(crop)
    assert n_cropped >= len(frames) // 2


def test_uncropped_clips_take_the_fast_path_and_agree(engine):
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(10, 16, 64, 64), dtype=np.uint8)  # noise: no letterbox
    hashes, crops = engine.hash_frames_letterbox(frames)
    assert not crops.any()
    assert np.array_equal(hashes, engine.hash_frames(frames))


def test_cropped_device_entry_point_and_errors(engine):
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(4)
    frames = rng.integers(0, 256, size=(6, 16, 50, 70), dtype=np.uint8)
    crops = np.array([[0, 0, 0, 0], [3, 4, 5, 6], [10, 0, 0, 7], [0, 20, 9, 0], [1, 1, 1, 1], [30, 30, 20, 20]], np.uint32)
    d = torch.from_numpy(frames).cuda()
    out = torch.zeros((6, 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    engine.hash_frames_cropped_device(d.data_ptr(), 6, 16, 70, 50, crops, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    for c in range(6):
        l, r, t, b = (int(x) for x in crops[c])
        rc, want, coefs = orc.hash_clip(np.ascontiguousarray(frames[c][:, t:50 - b, l:70 - r]), want_coefs=True)
        care = np.abs(coefs) >= 1e-6
        assert rc == 0 and not ((_bits(got[c:c + 1])[0] != _bits(want[None])[0])).any()
    bad = crops.copy()
    bad[2] = [40, 30, 0, 0]  # l + r >= w: Crop::from_edge_offsets asserts (crop.rs:21-22)
    with pytest.raises(vdf.VdfError) as ei:
        engine.hash_frames_cropped_device(d.data_ptr(), 6, 16, 70, 50, bad, out.data_ptr())
    assert ei.value.code == -5
    with pytest.raises(vdf.VdfError) as ei:
        engine.hash_frames_letterbox(frames[:, :9])
    assert ei.value.code == -1  # NotEnoughFrames


@pytest.mark.parametrize("h,w,crop", [(300, 720, (5, 3, 10, 7)), (426, 240, (5, 3, 60, 29)), (320, 176, (2, 13, 71, 13)),
                                      (300, 720, (6, 2, 0, 0)), (300, 720, (0, 0, 31, 17))])
def test_cropped_box_whose_padded_width_equals_the_pitch(engine, h, w, crop):
    """A crop box narrower than the frame whose LDS pitch (width + 3, rounded up to an odd multiple of 16) comes out equal to
    the frame's pitch takes the linear DMA loop of the cropped stream kernel although it does not start on a dword: the
    copy must start at the dword below the first pixel (LDS-DMA does not simply ignore the low address bits; found at
    240 and 176 wide, could have hit 720 wide)."""
    rng = np.random.default_rng(h + w)
    frames = rng.integers(0, 256, size=(24, 16, h, w), dtype=np.uint8)
    crops = np.tile(np.array(crop, np.uint32), (24, 1))
    crops[::3] = 0  # a third of the clips uncropped: mixed boxes in one launch
    d = torch.from_numpy(frames).cuda()
    out = torch.zeros((24, 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    engine.hash_frames_cropped_device(d.data_ptr(), 24, 16, w, h, crops, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    for c in range(24):
        l, r, t, b = (int(x) for x in crops[c])
        rc, want, coefs = orc.hash_clip(np.ascontiguousarray(frames[c][:, t:h - b, l:w - r]), want_coefs=True)
        assert rc == 0 and np.array_equal(got[c], want), c


def test_gen_hashes_mirrors_the_builder_default(engine):
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(11)
    frames = _letterboxed(rng, 6, 48, 64)
    hs = vdf.gen_hashes(frames, [f"v{i}" for i in range(6)], list(range(6)), engine=engine)  # default: Letterbox
    for i, vh in enumerate(hs):
        rc, want, coefs, _ = orc.hash_clip_letterbox(frames[i], want_coefs=True)
        care = np.abs(coefs) >= 1e-6
        assert not ((_bits(vh.hash[None])[0] != _bits(want[None])[0])).any()
        assert vh.src_path() == f"v{i}" and vh.duration() == i
    plain = vdf.gen_hashes(frames, ["p"] * 6, [0] * 6, cropdetect=vdf.Cropdetect.NONE, engine=engine)
    assert np.array_equal(np.stack([p.hash for p in plain]), engine.hash_frames(frames))
    with pytest.raises(vdf.NotEnoughFrames):
        vdf.gen_hashes(frames[:, :10], ["p"] * 6, [0] * 6, engine=engine)


def test_cropdetect_letterbox_mirror_returns_crops(engine):
    """api.cropdetect_letterbox: the reference's function (video_frames_gray.rs:201-210) and return type (crop.rs) for a batch."""
    import vid_dup_finder_lib_amd as vdf

    pix = [0, 0, 0, 0, 0, 0, 255, 255, 255, 0, 0, 255, 255, 255, 0, 0, 255, 255, 255, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    kat = np.tile(np.array(pix, np.uint8).reshape(1, 1, 6, 5), (1, 16, 1, 1))
    (c,) = vdf.cropdetect_letterbox(kat, engine=engine)
    assert c == vdf.Crop.from_edge_offsets((5, 6), 1, 1, 1, 2) and c.as_view_args() == (1, 1, 3, 3)  # video_frames_gray.rs:444-459
    rng = np.random.default_rng(12)
    frames = _letterboxed(rng, 10, 48, 64)
    crops = vdf.cropdetect_letterbox(frames, engine=engine)
    for i, c in enumerate(crops):
        l, r, t, b = (int(v) for v in orc.cropdetect_letterbox(frames[i]))
        assert c == vdf.Crop.from_edge_offsets((64, 48), l, r, t, b)
        x, y, bw, bh = c.as_view_args()  # the box's view is what the reference crops out and hashes (video_hash_builder.rs:197-201)
        rc, want, _ = orc.hash_clip(np.ascontiguousarray(frames[i][:, y:y + bh, x:x + bw]), want_coefs=True)
        assert rc == 0 and np.array_equal(engine.hash_frames_letterbox(frames[i:i + 1])[0][0], want)
    with pytest.raises(vdf.NotEnoughFrames):
        vdf.cropdetect_letterbox(frames[:, :10], engine=engine)


@pytest.mark.parametrize("base,pad_f,pad_c", [(5, 29, 77), (4, 28, 76)])
@pytest.mark.parametrize("h,w", [(64, 64), (90, 160), (270, 480), (360, 640)])
def test_letterbox_strided_misaligned_device_buffers(engine, h, w, base, pad_f, pad_c):
    """The device letterbox entry point with padded frame/clip strides, 17 frames per clip and a base pointer 5 bytes off
    alignment (whole-line cropped kernel) or 4 bytes off with strides that are multiples of 4 (the linear-stream cropped
    kernel still applies: its DMA needs dword-aligned frame bases only): same crops and hashes as the packed host path."""
    rng = np.random.default_rng(11 * h + w)
    n, nf = 6, 17
    frames = _letterboxed(rng, n, h, w)
    frames = np.concatenate([frames, frames[:, :nf - frames.shape[1]]], axis=1) if frames.shape[1] < nf else frames[:, :nf]
    fs = w * h + pad_f
    cs = nf * fs + pad_c
    buf = np.full(base + (n - 1) * cs + (nf - 1) * fs + w * h, 0x55, np.uint8)
    for c in range(n):
        for f in range(nf):
            o = base + c * cs + f * fs
            buf[o:o + w * h] = frames[c, f].reshape(-1)
    d_buf = torch.from_numpy(buf).cuda()
    d_out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    crops = engine.hash_frames_letterbox_device(d_buf.data_ptr() + base, n, nf, w, h, d_out.data_ptr(), frame_stride=fs,
                                                clip_stride=cs)
    torch.cuda.synchronize()
    want_h, want_c = engine.hash_frames_letterbox(np.ascontiguousarray(frames))
    assert np.array_equal(crops, want_c)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want_h)


@pytest.mark.parametrize("h,w", [(1080, 1920), (900, 1600), (322, 1440), (200, 1680),  # per-wave form at the frame's own pitch
                                 (360, 640), (720, 1280), (270, 480), (176, 320), (594, 1056),  # chunk form, rows as they are (W % 16 == 0)
                                 (576, 1024), (432, 768), (300, 1536), (260, 1792),  # chunk form, re-pitched rows
                                 (480, 852), (240, 500), (300, 1364),  # W % 4 == 0, not 16
                                 (480, 854), (768, 1366), (333, 999), (240, 426), (201, 1001),  # rows off a dword boundary
                                 (400, 2048), (288, 2560), (432, 3840), (256, 4096), (300, 2064)])  # K-split form
def test_top_bottom_bars_take_the_stream_kernels(engine, monkeypatch, h, w):
    """Full-width crop boxes (top / bottom bars only) stream like shorter frames: the linear-stream kernels the uncropped call
    takes at that width, as ROWCROP instantiations with a per-clip first row, height and vertical table.  Boxes of every block
    count (fewer than four 16-row blocks = waves without work, heights off a multiple of 16, one row, odd first rows = chunks that
    start off 4- and 16-byte boundaries), clips without bars among them; equal to the oracle on the cropped copies, to the
    general cropped kernels (VDF_NO_ROWCROP, and VDF_NO_WAVESTREAM where only the per-wave form has a ROWCROP instantiation)."""
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(h * 7 + w)
    tb = [(0, 0), (h // 8, h // 8), (1, 0), (0, 1), (h // 2 - 20, h // 2 - 21), (h - 17, 0), (0, h - 33), (h // 3, 5), (7, h // 3),
          (h - 1, 0), (h // 2 - 32, h // 2 - 32), (16, 16), (3, 2), (h // 5 | 1, h // 7)]
    if (w * h) % 16:
        pytest.skip("frames that do not end on a 16-byte boundary stay on the general kernels")
    n = len(tb)
    frames = rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
    crops = np.array([(0, 0, t, b) for t, b in tb], np.uint32)
    want = np.stack([orc.hash_clip(np.ascontiguousarray(frames[c][:, tb[c][0]:h - tb[c][1]]))[1] for c in range(n)])
    d = torch.from_numpy(frames).cuda()
    out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    engine.hash_frames_cropped_device(d.data_ptr(), n, 16, w, h, crops, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    out2 = torch.zeros_like(out)
    # VDF_ROWCROP_ALL: also at the widths where the dispatch keeps the general kernels because they measured faster (768, 1024, 1536, 2048)
    for env in ("VDF_ROWCROP_ALL", "VDF_NO_ROWCROP", "VDF_NO_WAVESTREAM"):
        monkeypatch.setenv(env, "1")
        out2.zero_()
        engine.hash_frames_cropped_device(d.data_ptr(), n, 16, w, h, crops, out2.data_ptr())
        torch.cuda.synchronize()
        monkeypatch.delenv(env)
        assert torch.equal(out, out2), env
    # one clip with a side bar sends the whole call down the general cropped path: same hashes for the others
    crops2 = crops.copy()
    crops2[3] = (8, 0, 0, 1)
    engine.hash_frames_cropped_device(d.data_ptr(), n, 16, w, h, crops2, out2.data_ptr())
    torch.cuda.synchronize()
    keep = [c for c in range(n) if c != 3]
    assert torch.equal(out[keep], out2[keep])
    bad = crops.copy()
    bad[5] = (0, 0, h - 10, 10)
    with pytest.raises(vdf.VdfError) as ei:
        engine.hash_frames_cropped_device(d.data_ptr(), n, 16, w, h, bad, out2.data_ptr())
    assert ei.value.code == -5


@pytest.mark.parametrize("h,w", [(120, 200), (64, 17), (70, 33), (300, 64), (1080, 1920), (131, 250)])
def test_side_bars_walk_in_batches_of_columns(engine, h, w):
    """Pillarboxed clips: from the second column strip on the device judges a batch of strips at a time (column_strips: sixteen).  Bars of 1, 15,
    16, 17, 18, 32, 33 ... columns on either side, strips that fail in the middle of a batch (just over 10 % outliers, values just outside
    +-16 of the mode, a tie between two modes), noisy bars, frames narrower than a batch, a uniform frame (every strip passes, the edges
    converge): the same crops as the oracle's strip-by-strip take_while."""
    rng = np.random.default_rng(h * 31 + w)
    widths = [0, 1, 2, 15, 16, 17, 18, 31, 32, 33, 34, 47, 48, 49, 63, 64, 65]
    cases = []
    for k, lw in enumerate(widths):
        rw = widths[(k * 7 + 3) % len(widths)]
        if lw + rw + 2 > w:
            lw, rw = lw % max(1, w // 3), rw % max(1, w // 3)
        f = rng.integers(60, 200, size=(16, h, w), dtype=np.uint8)
        base = int(rng.integers(0, 230))
        if lw:
            f[:, :, :lw] = base + rng.integers(0, 12, size=(16, h, lw))
        if rw:
            f[:, :, w - rw:] = base + rng.integers(0, 12, size=(16, h, rw))
        kind = k % 6
        col = lw // 2 if lw > 2 else None
        if col is not None:
            if kind == 1:    # just over 10 % outliers in one column of the left bar: the walk stops there
                f[:, : h // 10 + 1, col] = 255 - base // 2
            elif kind == 2:  # exactly 10 % or just under: still letterbox (> 0.9 is strict on the other side)
                f[:, : max(0, (h - 1) // 10), col] = 255 - base // 2
            elif kind == 3:  # values at mode +- 16 and +- 17 in one column
                f[:, :, col] = base
                f[:, : h // 20 + 1, col] = min(255, base + 16)
                f[:, h // 20 + 1: h // 8 + 2, col] = min(255, base + 17)
            elif kind == 4:  # two values with the same count: the LAST maximum is the mode (Iterator::max_by_key)
                f[:, : h // 2, col] = base
                f[:, h // 2: 2 * (h // 2), col] = min(255, base + 20)
        if rw > 3 and kind == 5:
            f[:, : h // 9 + 1, w - 1 - rw // 3] = 255 - base // 2
        cases.append(f)
    cases.append(np.full((16, h, w), 77, np.uint8))  # uniform: converging edges -> no crop
    g = np.full((16, h, w), 20, np.uint8)
    g[:, :, w // 2] = 200  # one bright column in the middle: the edges meet it from both sides
    cases.append(g)
    frames = np.stack(cases)
    d = torch.from_numpy(frames).cuda()
    crops = torch.zeros((len(frames), 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    engine.cropdetect_letterbox_device(d.data_ptr(), len(frames), 16, w, h, crops.data_ptr())
    torch.cuda.synchronize()
    got = crops.cpu().numpy().astype(np.uint32)
    want = np.array([orc.cropdetect_letterbox(c) for c in frames], np.uint32)
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0]
    assert (want[:, :2] > 16).any() or w < 40  # the batches were really walked


@pytest.mark.parametrize("h,w", [(1080, 1920), (720, 1280), (480, 854), (300, 1001), (576, 1024), (432, 3840), (144, 2000), (400, 700)])
def test_boxes_that_share_their_column_range_take_the_per_wave_kernel(engine, monkeypatch, h, w):
    """Pillarboxed clips whose boxes share a column range (x0, width) form groups, and a group of four or more goes through the per-wave
    stream kernel with the frame's pitch, a column offset and the box's band table (one launch per range) while the clips with full-width
    boxes stream next to them and odd ones out take the gather kernel.  Ranges on and off dword boundaries (MODE 1 / 2 of the DMA),
    widths off a multiple of 4 and of 16, per-clip rows inside a range, boxes narrower than the kernel takes (< 513 columns: gather):
    equal to the oracle on the cropped copies and to the gather kernel alone (VDF_NO_BOXSTREAM)."""
    rng = np.random.default_rng(h * 13 + w)
    ranges = [(w // 8, w // 8), (w // 8 + 1, w // 8 - 1), (16, 0), (3, 5), (w // 2, 8), (0, w // 5)]
    crops = []
    for k, (l, r) in enumerate(ranges):
        for j in range(5):  # five clips per range, each with its own rows
            crops.append((l, r, (j * 7) % (h // 4), (j * 11 + k) % (h // 4)))
    crops += [(0, 0, 0, 0), (0, 0, h // 8, h // 8), (0, 0, 0, 3), (w // 3, 1, 2, 0), (5, w // 4, 0, 0)]  # full-width boxes and two odd ones out
    order = rng.permutation(len(crops))
    crops = np.array([crops[i] for i in order], np.uint32)
    n = len(crops)
    frames = rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
    want = np.stack([orc.hash_clip(np.ascontiguousarray(frames[c][:, crops[c][2]:h - crops[c][3], crops[c][0]:w - crops[c][1]]))[1] for c in range(n)])
    d = torch.from_numpy(frames).cuda()
    out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    engine.hash_frames_cropped_device(d.data_ptr(), n, 16, w, h, crops, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0].tolist()
    monkeypatch.setenv("VDF_NO_BOXSTREAM", "1")
    out2 = torch.zeros_like(out)
    engine.hash_frames_cropped_device(d.data_ptr(), n, 16, w, h, crops, out2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


@pytest.mark.parametrize("h,w", [(1080, 1920), (300, 1001), (96, 250), (64, 64)])
def test_uniform_bars_are_accepted_from_their_loads_and_blemished_ones_are_not(engine, h, w):
    """Round 5: a batch of strips whose pixels stay inside a window of `tolerance` (a bar: one value per strip, as a rule) is accepted without
    a histogram - rows by their exact min / max, columns by staying within tolerance / 2 of row 0's pixel.  Everything around that shortcut
    against the oracle: clean bars of one value, of one value PER STRIP (a gradient across the bar), bars with a single blemish in the first / a
    middle / the last row group or in the tail bytes of a row, blemishes that add up to just under and just over 10 % of a strip, noisy bars
    inside +-16, a bar whose rows are constant but differ from each other."""
    rng = np.random.default_rng(h * 7 + w)
    bar_w, bar_h = max(w // 8, 3), max(h // 9, 3)
    cases = []

    def clip(edit):
        f = rng.integers(60, 200, size=(16, h, w), dtype=np.uint8)
        edit(f)
        cases.append(f)

    def sides(f, v=16):
        f[:, :, :bar_w] = v
        f[:, :, w - bar_w:] = v

    def rows(f, v=16):
        f[:, :bar_h] = v
        f[:, h - bar_h:] = v

    clip(lambda f: sides(f))
    clip(lambda f: rows(f))
    clip(lambda f: (sides(f), rows(f)))
    def very_wide(f):  # bars of 0.4 w: several wide probes (columns_narrow) in a row, then counted batches, then single strips
        bw = int(w * 0.4)
        f[:, :, :bw] = 17
        f[:, :, w - bw + 3:] = 17
    clip(very_wide)
    def wide_with_a_late_blemish(f):  # clean for the first probe, one odd pixel inside the second: the walk must go on counting from there
        bw = int(w * 0.4)
        f[:, :, :bw] = 17
        f[:, h - 1, min(bw - 2, 5 * (w // 32) + 3)] = 200
    clip(wide_with_a_late_blemish)
    clip(lambda f: f.__setitem__((slice(None), slice(None), slice(0, bar_w)), np.arange(bar_w, dtype=np.uint8)[None, None, :] * 3 + 5))  # one value per column
    clip(lambda f: f.__setitem__((slice(None), slice(0, bar_h)), (np.arange(bar_h, dtype=np.uint8) * 5 + 3)[None, :, None]))  # one value per row
    for at in (0, h // 2, h - 1):  # a single blemish somewhere in an otherwise constant side bar: still > 90 %
        def e(f, at=at):
            sides(f)
            f[:, at, bar_w // 2] = 250
            f[:, at, w - 1] = 251
        clip(e)
    for at in (0, w // 2, w - 1):  # and in a top / bottom bar, incl. the last (tail) bytes of a row
        def e(f, at=at):
            rows(f)
            f[:, 1, at] = 250
            f[:, h - 1, at] = 3
        clip(e)
    for frac in (0.099, 0.101):  # blemishes that add up to just under / just over 10 % of the second column and of the second row
        def e(f, frac=frac):
            sides(f)
            rows(f)
            k_col = int(np.floor(h * frac)) if frac < 0.1 else int(np.ceil(h * frac))
            k_row = int(np.floor(w * frac)) if frac < 0.1 else int(np.ceil(w * frac))
            f[:, bar_h:bar_h + k_col, 1] = 255   # column 1: k_col pixels far from the mode
            f[:, 1, bar_w:bar_w + k_row] = 255   # row 1
        clip(e)
    def noisy(f):
        f[:, :, :bar_w] = rng.integers(8, 25, size=(16, h, bar_w), dtype=np.uint8)   # inside +-16 of any mode it can have
        f[:, :bar_h] = rng.integers(100, 117, size=(16, bar_h, w), dtype=np.uint8)
    clip(noisy)
    def striped(f):  # rows constant, each with its own value: the column walk leaves the shortcut at the second row group
        f[:, :, :bar_w] = (np.arange(h, dtype=np.uint8) % 7 + 10)[None, :, None]
    clip(striped)
    frames = np.stack(cases)
    d = torch.from_numpy(frames).cuda()
    crops = torch.zeros((len(frames), 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    engine.cropdetect_letterbox_device(d.data_ptr(), len(frames), 16, w, h, crops.data_ptr())
    torch.cuda.synchronize()
    got = crops.cpu().numpy().astype(np.uint32)
    want = np.array([orc.cropdetect_letterbox(c) for c in frames], np.uint32)
    assert np.array_equal(got, want), (got.tolist(), want.tolist())
    assert tuple(want[0]) == (bar_w, bar_w, 0, 0) and tuple(want[1]) == (0, 0, bar_h, bar_h)


@pytest.mark.parametrize("h,w,offset", [(512, 640, 0), (540, 1920, 0), (512, 640, 64), (512, 576, 0), (520, 704, 32)])
def test_clean_side_bars_are_probed_in_aligned_windows(engine, h, w, offset):
    """Round 5: the side walk of frames from 512 rows on probes clean bars in ALIGNED windows of 128, then 64 bytes (columns_narrow: whole
    cache lines / sectors, read once) before it falls back to the counted 32-column batches - the walk is bound by HBM transactions.  Bar
    widths on both sides of every window boundary, different left and right, a bar that is clean only up to a boundary, a gradient bar
    (constant columns of different values), a noisy bar (every probe fails at once), a frame whose rows are all equal (every column constant:
    left and right meet, "no crop"), frame widths that are multiples of 128, of 64 only and of neither, and a buffer whose frames start off
    a line boundary: boxes equal the oracle's."""
    rng = np.random.default_rng(h + w + offset)
    widths = [1, 2, 31, 33, 63, 64, 65, 96, 127, 128, 129, 160, 191, 192, 193, 255, 256, 257]
    cases = []
    for k, bw in enumerate(widths):
        f = rng.integers(60, 200, size=(16, h, w), dtype=np.uint8)
        f[:, :, :bw] = 16
        f[:, :, w - widths[(k * 7 + 3) % len(widths)]:] = 18
        cases.append(f)
    f = rng.integers(60, 200, size=(16, h, w), dtype=np.uint8)  # clean up to column 128, then a bar with one odd pixel per column block
    f[:, :, :200] = 16
    f[:, 7, 130] = 99
    f[:, h - 3, 170] = 99
    cases.append(f)
    f = rng.integers(60, 200, size=(16, h, w), dtype=np.uint8)  # gradient bar
    f[:, :, :150] = (np.arange(150, dtype=np.uint8) + 3)[None, None, :]
    cases.append(f)
    f = rng.integers(60, 200, size=(16, h, w), dtype=np.uint8)  # noisy bars inside +-16
    f[:, :, :140] = rng.integers(10, 24, size=(16, h, 140), dtype=np.uint8)
    f[:, :, w - 70:] = rng.integers(10, 24, size=(16, h, 70), dtype=np.uint8)
    cases.append(f)
    row = rng.integers(60, 200, size=(1, 1, w), dtype=np.uint8)  # all rows equal: every column is constant
    cases.append(np.broadcast_to(row, (16, h, w)).copy())
    frames = np.stack(cases)
    n = len(frames)
    buf = torch.zeros(n * 16 * h * w + 256, dtype=torch.uint8, device="cuda")
    buf[offset:offset + frames.size] = torch.from_numpy(frames).cuda().reshape(-1)
    crops = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    engine.cropdetect_letterbox_device(buf.data_ptr() + offset, n, 16, w, h, crops.data_ptr())
    torch.cuda.synchronize()
    got = crops.cpu().numpy().astype(np.uint32)
    want = np.array([orc.cropdetect_letterbox(c) for c in frames], np.uint32)
    assert np.array_equal(got, want), (got.tolist(), want.tolist())
    assert tuple(want[widths.index(128)][:1]) == (128,) and tuple(want[-1]) == (0, 0, 0, 0)


@pytest.mark.parametrize("h,w", [(1080, 1920), (540, 1024), (300, 1001), (96, 250)])
def test_noisy_bars_are_accepted_by_their_range_and_wide_ranges_are_counted(engine, h, w):
    """Round 5: strips whose pixels all lie inside a window of `tolerance` are letterbox whatever their mode is, so a bar a lossy codec has
    left as 16 + {0..3} is accepted from its loads alone (rows: exact min / max over four rows; columns: every row within tolerance / 2 of row
    0's pixel) instead of through a histogram whose few bins are all LDS conflicts.  The edges of that rule against the oracle: ranges of
    exactly 16 and of 17 (the 17th value rare: still letterbox by the count; common: not), a first row that is the outlier (the anchor of
    the column check), values 8 and 9 away from the anchor on both sides, bars at 0 and at 255 (no wrap in the 16-bit lanes), one noisy
    strip among clean ones, noise in one frame of the clip only."""
    rng = np.random.default_rng(h * 3 + w)
    bw, bh = max(w // 7, 5), max(h // 8, 5)
    cases = []

    def clip(edit):
        f = rng.integers(70, 190, size=(16, h, w), dtype=np.uint8)
        edit(f)
        cases.append(f)

    def side_bars(f, lo, hi, p=None):
        vals = np.arange(lo, hi + 1)
        f[:, :, :bw] = rng.choice(vals, size=(16, h, bw), p=p).astype(np.uint8)
        f[:, :, w - bw:] = rng.choice(vals, size=(16, h, bw), p=p).astype(np.uint8)

    def row_bars(f, lo, hi, p=None):
        vals = np.arange(lo, hi + 1)
        f[:, :bh] = rng.choice(vals, size=(16, bh, w), p=p).astype(np.uint8)
        f[:, h - bh:] = rng.choice(vals, size=(16, bh, w), p=p).astype(np.uint8)

    for lo, hi in ((16, 19), (0, 3), (252, 255), (0, 16), (239, 255), (100, 116)):  # ranges of at most 16: letterbox for certain
        clip(lambda f, lo=lo, hi=hi: side_bars(f, lo, hi))
        clip(lambda f, lo=lo, hi=hi: row_bars(f, lo, hi))
    rare = np.array([0.97 / 17] * 17 + [0.03])   # 18 values, range 17: the last one rare -> > 90 % near any mode
    rare /= rare.sum()
    clip(lambda f: side_bars(f, 20, 37, rare))
    clip(lambda f: row_bars(f, 20, 37, rare))
    ends = np.array([0.45] + [0.1 / 16] * 16 + [0.45])  # range 17, both ends common: the mode's window leaves out 45 %
    ends /= ends.sum()
    clip(lambda f: side_bars(f, 20, 37, ends))
    clip(lambda f: row_bars(f, 20, 37, ends))
    for d in (8, 9, -8, -9):  # a constant bar of 50 whose first row / first column is d away and a few more pixels 8 the other way
        def e(f, d=d):
            f[:, :, :bw] = 50
            f[:, 0, :bw] = 50 + d
            f[:, 1 + h // 2, :bw] = 50 - (8 if d > 0 else -8)
        clip(e)
        def e2(f, d=d):
            f[:, :bh] = 50
            f[:, :bh, 0] = 50 + d
            f[:, :bh, w // 2] = 50 - (8 if d > 0 else -8)
        clip(e2)
    def first_row_outlier(f):  # the anchor row is far off: the window check fails at once, the count accepts (one row in h)
        f[:, :, :bw] = 16
        f[:, 0, :bw] = 200
    clip(first_row_outlier)
    def one_noisy_strip(f):
        f[:, :, :bw] = 16
        f[:, :, bw // 2] = rng.integers(0, 256, size=(16, h), dtype=np.uint8)   # a strip of picture inside the bar: the walk stops there
        f[:, :bh] = 16
        f[:, bh // 2] = rng.integers(0, 256, size=(16, w), dtype=np.uint8)
    clip(one_noisy_strip)
    def one_frame_only(f):  # frame 0 has clean bars, frame 8 noisy wider ones: per-frame boxes differ, the union takes the smaller
        f[0, :, :bw] = 16
        f[8, :, :bw + 9] = rng.integers(14, 20, size=(h, bw + 9), dtype=np.uint8)
        f[0, :bh + 4] = rng.integers(14, 20, size=(bh + 4, w), dtype=np.uint8)
        f[8, :bh] = 16
    clip(one_frame_only)
    frames = np.stack(cases)
    d = torch.from_numpy(frames).cuda()
    crops = torch.zeros((len(frames), 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    engine.cropdetect_letterbox_device(d.data_ptr(), len(frames), 16, w, h, crops.data_ptr())
    torch.cuda.synchronize()
    got = crops.cpu().numpy().astype(np.uint32)
    want = np.array([orc.cropdetect_letterbox(c) for c in frames], np.uint32)
    assert np.array_equal(got, want), [(i, g, x) for i, (g, x) in enumerate(zip(got.tolist(), want.tolist())) if g != x]
    assert tuple(want[0]) == (bw, bw, 0, 0) and tuple(want[1]) == (0, 0, bh, bh)
    assert tuple(want[14]) == (0, 0, 0, 0) and tuple(want[15]) == (0, 0, 0, 0)  # the "both ends common" bars are picture


@pytest.mark.parametrize("h,w", [(64, 64), (48, 80), (96, 96), (128, 128), (90, 160), (100, 240), (128, 256), (33, 17), (64, 200), (120, 250), (17, 256), (128, 16)])
def test_small_frames_crop_boxes_take_one_workgroup_per_clip(h, w):
    """Round 5: crop boxes of small frames (at most 128 rows, 256 columns) go through resize_dct_hash_cropped_small_kernel - one workgroup per clip, a
    wave per four frames - instead of one workgroup per frame.  Random boxes (none, rows only, sides only, all four, one-pixel and one-row boxes,
    boxes that start off every alignment), 300 clips so that the last clips' loads reach the buffer's end (the careful loader), against the
    oracle on the cropped copies and against the kernel before (VDF_NO_SMALLCROP)."""
    import os

    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(h * 11 + w)
    n = 300
    frames = rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
    crops = np.zeros((n, 4), np.uint32)
    for c in range(n):
        kind = c % 6
        if kind == 0:
            continue
        t, b = (int(rng.integers(0, max(1, h // 2))) for _ in range(2))
        l, r = (int(rng.integers(0, max(1, w // 2))) for _ in range(2))
        if kind == 1: l = r = 0
        if kind == 2: t = b = 0
        if kind == 4: l, r, t, b = int(rng.integers(0, w)), 0, int(rng.integers(0, h)), 0; r = w - l - 1; b = h - t - 1  # one pixel
        if kind == 5: t = int(rng.integers(0, h)); b = h - t - 1; l = r = 0                                               # one row
        if t + b >= h: b = 0
        if l + r >= w: r = 0
        crops[c] = (l, r, t, b)
    crops[n - 1] = (0, 0, h - 1, 0) if h > 1 else (0, 0, 0, 0)  # the last clip's box is the buffer's last row
    want = np.stack([orc.hash_clip(np.ascontiguousarray(frames[c][:, crops[c][2]:h - crops[c][3], crops[c][0]:w - crops[c][1]]))[1] for c in range(n)])
    d = torch.from_numpy(frames).cuda()
    for env in ({}, {"VDF_NO_SMALLCROP": "1"}):
        os.environ.update(env)
        try:
            eng = vdf.Engine(0)
        finally:
            for k in env:
                os.environ.pop(k)
        try:
            out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            eng.hash_frames_cropped_device(d.data_ptr(), n, 16, w, h, crops, out.data_ptr())
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint64)
        finally:
            eng.close()
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, (env, bad[:8].tolist(), crops[bad[:8]].tolist())


def test_frames_and_boxes_of_one_chunk_in_large_batches():
    """Round 5: a frame (or crop box) that fits ONE chunk of the chunk-stream kernels ends in the step that began with wave 0 reading the previous
    frame's partial sums, and nothing ordered that read before the other waves' next write - a wave without a block in a short chunk got
    there at once: 1 - 2 clips in 30 000 came out wrong (found by a size sweep with the kernel forced onto 64 x 48 frames; in the product path only
    letterbox boxes of at most 64 rows reach it).  Now an LDS barrier in that case.  Large batches of such boxes through the ROWCROP chunk
    kernel and the cropped stream kernel against the general kernels (and the oracle on the first clips), and the forced form of the find."""
    import os

    import vid_dup_finder_lib_amd as vdf

    def engine_with(env):
        for k, v in env.items():
            os.environ[k] = v
        try:
            return vdf.Engine(0)
        finally:
            for k in env:
                os.environ.pop(k, None)

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    h, w, n = 144, 176, 12000
    frames = torch.randint(0, 256, (n, 16, h, w), dtype=torch.uint8, device=dev, generator=g)
    host8 = frames[:8].cpu().numpy()
    for crop, general in (((0, 0, 50, 50), {"VDF_NO_ROWCROP": "1", "VDF_RESIZE_MODE": "4"}), ((8, 8, 50, 50), {"VDF_RESIZE_MODE": "4"}),
                          ((0, 0, 100, 11), {"VDF_NO_ROWCROP": "1", "VDF_RESIZE_MODE": "4"})):   # boxes of 44, 44 and 33 rows: one chunk, waves without a block
        crops = np.tile(np.array(crop, np.uint32), (n, 1))
        l, r, t, b = crop
        want8 = np.stack([orc.hash_clip(np.ascontiguousarray(host8[c][:, t:h - b, l:w - r]))[1] for c in range(8)])
        ref_eng = engine_with(general)
        ref = torch.zeros((n, 16), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()  # the library's own stream does not wait for torch's fill
        ref_eng.hash_frames_cropped_device(frames.data_ptr(), n, 16, w, h, crops, ref.data_ptr())
        torch.cuda.synchronize()
        ref_eng.close()
        assert np.array_equal(ref[:8].cpu().numpy().view(np.uint64), want8)
        for rep in range(3):
            eng = vdf.Engine(0)
            out = torch.zeros((n, 16), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            eng.hash_frames_cropped_device(frames.data_ptr(), n, 16, w, h, crops, out.data_ptr())
            torch.cuda.synchronize()
            eng.close()
            bad = torch.nonzero((out != ref).any(dim=1)).flatten()
            assert len(bad) == 0, (crop, rep, bad[:10].tolist())
    del frames
    for hh, ww, nn in ((48, 64, 30000), (32, 64, 30000), (40, 128, 20000)):  # the forced form: whole frames of one short chunk
        frames = torch.randint(0, 256, (nn, 16, hh, ww), dtype=torch.uint8, device=dev, generator=g)
        torch.cuda.synchronize()  # the library's own stream does not wait for torch's
        ref_eng = vdf.Engine(0)
        ref = torch.zeros((nn, 16), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ref_eng.hash_frames_device(frames.data_ptr(), nn, 16, ww, hh, ref.data_ptr())
        torch.cuda.synchronize()
        ref_eng.close()
        for rep in range(3):
            eng = engine_with({"VDF_RESIZE_MODE": "5"})
            out = torch.zeros((nn, 16), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            eng.hash_frames_device(frames.data_ptr(), nn, 16, ww, hh, out.data_ptr())
            torch.cuda.synchronize()
            eng.close()
            bad = torch.nonzero((out != ref).any(dim=1)).flatten()
            assert len(bad) == 0, (hh, ww, rep, bad[:10].tolist())


def test_a_large_mixed_batch_of_large_frames():
    """203 clips of 640 x 416 with bars of every kind - top / bottom, sides, a corner, none - and black probe frames (a fade-in: every strip
    of every edge is letterbox until two walkers meet; frame 8 decides), through the two-pass detect (first strips of all four edges; then
    four walkers per frame with bars) and the crop split, on the caller's own non-default stream, twice: boxes and hashes equal the oracle's."""
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(97)
    h, w, n = 416, 640, 203
    frames = rng.integers(40, 220, size=(n, 16, h, w), dtype=np.uint8)
    for c in range(n):
        kind = c % 5 if c < 153 else 0
        if kind == 1:
            frames[c, :, :30 + c % 7] = 16
            frames[c, :, h - 28:] = 17
        elif kind == 2:
            frames[c, :, :, :80 + 4 * (c % 3)] = 16
            frames[c, :, :, w - 80:] = 16
        elif kind == 3:
            frames[c, :, :20] = 15
            frames[c, :, :, :64] = 15
        elif kind == 4 and c % 20 == 4:
            frames[c, 0] = 16
    res = [orc.hash_clip_letterbox(f) for f in frames]
    want = np.stack([r[1] for r in res])
    want_crops = np.array([r[3] for r in res], np.uint32)
    assert len({tuple(c) for c in want_crops}) > 10
    eng = vdf.Engine(0)
    try:
        d = torch.from_numpy(frames).cuda()
        out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        for _ in range(2):
            with torch.cuda.stream(side):
                out.zero_()
                crops = eng.hash_frames_letterbox_device(d.data_ptr(), n, 16, w, h, out.data_ptr(), stream=side.cuda_stream)
            side.synchronize()
            assert np.array_equal(crops, want_crops)
            got = out.cpu().numpy().view(np.uint64)
            assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0].tolist()
    finally:
        eng.close()


def test_contexts_give_their_device_memory_back():
    """Every device buffer a context grows (the crop split's descriptor arrays, the detect work list, tables, staging, hit lists) is released
    with the context: create - hash letterboxed, pillarboxed and mixed batches - search - close, and the library's own count of live
    device bytes (vdf_live_device_bytes) is back where it was (a context once forgot three buffers in its destructor)."""
    import vid_dup_finder_lib_amd as vdf
    from vid_dup_finder_lib_amd import _capi

    lib = _capi.load()
    rng = np.random.default_rng(7)
    h, w, n = 360, 1280, 48
    frames = rng.integers(40, 220, size=(n, 16, h, w), dtype=np.uint8)
    frames[0::3, :, :40] = 16
    frames[0::3, :, h - 40:] = 16
    frames[1::3, :, :, :160] = 16
    frames[1::3, :, :, w - 160:] = 16
    d = torch.from_numpy(frames).cuda()
    out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
    words = rng.integers(0, 2**63, size=(5000, 16), dtype=np.int64).astype(np.uint64)
    before, pinned_before = lib.vdf_live_device_bytes(), lib.vdf_live_pinned_bytes()
    for _ in range(3):
        eng = vdf.Engine(0)
        try:
            eng.hash_frames_letterbox_device(d.data_ptr(), n, 16, w, h, out.data_ptr())
            eng.hash_frames(frames[:2])
            eng.search_self_sorted(words, np.zeros(len(words), np.uint32), 350)
            eng.search_refs_sorted(words, np.zeros(len(words), np.uint32), words[:100], np.zeros(100, np.uint32), 350)
            torch.cuda.synchronize()
            assert lib.vdf_live_device_bytes() > before and lib.vdf_live_pinned_bytes() > pinned_before
            assert lib.vdf_ctx_device(eng.ctx) == 0
        finally:
            eng.close()
        assert lib.vdf_live_device_bytes() == before and lib.vdf_live_pinned_bytes() == pinned_before
    eng = vdf.Engine(0)  # a batching queue: two slots per GPU with private contexts
    try:
        from vid_dup_finder_lib_amd.engine import HashQueue

        q = HashQueue(eng, w, h, max_batch=4, max_wait_us=100, letterbox=True)
        try:
            for c in range(3):
                q.submit(frames[c])
        finally:
            q.close()
    finally:
        eng.close()
    assert lib.vdf_live_device_bytes() == before
    eng = vdf.Engine(devices=[0, 0])  # the multi-device form: per-device sub-contexts
    try:
        eng.hash_frames_letterbox(frames[:8])
        eng.search_self_sorted(words, np.zeros(len(words), np.uint32), 350)
    finally:
        eng.close()
    assert lib.vdf_live_device_bytes() == before
