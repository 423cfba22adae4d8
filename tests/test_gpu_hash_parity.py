"""HIP VideoHash construction (through the C ABI) vs the CPU oracle.

Device and oracle run the SAME f64 operation sequence (rustdct's split-radix butterflies, products and sums rounded
separately, identical twiddle constants), and the resize is exact integer arithmetic, so the comparison is on whole hash
words: no bit is masked.  The don't-care rule of DESIGN.md (|coef| < 1e-6) only describes how far the ORACLE can be
trusted against the real crate; the device reports the per-clip count of such coefficients and it must equal the
oracle's count exactly."""
import numpy as np
import pytest

from oracle import vdf_oracle as orc

pytestmark = pytest.mark.gpu
TINY = 1e-6


def _bits(words):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little").reshape(len(words), 1024)


def _check(engine, frames):
    got, dc = engine.hash_frames(frames, want_dontcare=True)
    want, coefs = orc.hash_clips_with_coefs(frames)
    care = np.abs(coefs) >= TINY
    gb, wb = _bits(got)[:, :1000], _bits(want)[:, :1000]
    bad = gb != wb
    assert not bad.any(), f"{bad.sum()} hash bits differ ({(bad & care).sum()} of them outside the don't-care set)"
    assert (_bits(got)[:, 1000:] == 0).all()  # padding bits stay zero when built from frames
    assert np.array_equal(dc, (~care).sum(axis=1)), "device don't-care count differs from the oracle's"
    return got, want, care


@pytest.mark.parametrize("h,w", [(16, 16), (64, 64), (48, 80), (16, 64), (64, 16), (33, 17), (120, 68), (9, 9)])
def test_hash_bits_match_oracle(engine, h, w):
    rng = np.random.default_rng(100 + h * 131 + w)
    frames = rng.integers(0, 256, size=(24, 16, h, w), dtype=np.uint8)
    _check(engine, frames)


def test_video_like_clips(engine):
    """Smooth (low-pass) content is where f32 would flip ~2% of hashes; f64 must hold."""
    from scipy.ndimage import gaussian_filter

    rng = np.random.default_rng(7)
    noise = rng.standard_normal((32, 16, 64, 64))
    smooth = gaussian_filter(noise, sigma=(0, 2.0, 6.0, 6.0))
    smooth = (smooth - smooth.min()) / (smooth.max() - smooth.min()) * 255.0
    got, want, care = _check(engine, smooth.astype(np.uint8))
    assert care.all() or care.mean() > 0.99


def test_extra_frames_ignored_and_too_few_rejected(engine):
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(8)
    frames = rng.integers(0, 256, size=(3, 20, 32, 32), dtype=np.uint8)
    a = engine.hash_frames(frames)
    b = engine.hash_frames(frames[:, :16])
    assert np.array_equal(a, b)  # dct_3d.rs:25 take(16)
    with pytest.raises(vdf.VdfError) as ei:
        engine.hash_frames(frames[:, :10])
    assert ei.value.code == -1  # VDF_E_NOT_ENOUGH_FRAMES <-> Error::NotEnoughFrames
    with pytest.raises(vdf.NotEnoughFrames):
        vdf.VideoHash.from_frames(list(frames[0, :10]), "x.mp4", 3, engine=engine)
    with pytest.raises(vdf.NotEnoughFrames):
        vdf.VideoHash.from_frames([], "x.mp4", 3, engine=engine)
    vh = vdf.VideoHash.from_frames(list(frames[0]), "x.mp4", 3, engine=engine)
    assert np.array_equal(vh.hash, a[0]) and vh.duration() == 3 and vh.src_path() == "x.mp4"


@pytest.mark.parametrize("h,w", [(16, 16), (64, 64), (90, 160), (270, 480)])
def test_static_constant_and_symmetric_clips_unmasked(engine, h, w):
    """The clips whose coefficients are mathematically zero in bulk (slideshows, black frames, mirrored content): under
    rustdct's butterfly structure those coefficients are exact +-0.0, so the bits are 0 - not rounding noise.
    Unmasked: identical frames -> every kt >= 1 bit is 0; constant frames -> every AC bit is 0, DC bit = (value > 128);
    left-right mirrored frames -> every odd-kx bit is 0; and all words equal the oracle's."""
    rng = np.random.default_rng(h * 77 + w)
    one = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    half = rng.integers(0, 256, size=(16, h, w // 2), dtype=np.uint8)
    frames = np.stack([np.broadcast_to(one, (16, h, w)), np.zeros((16, h, w), np.uint8), np.full((16, h, w), 255, np.uint8),
                       np.full((16, h, w), 128, np.uint8), np.full((16, h, w), 129, np.uint8),
                       np.concatenate([half, half[:, :, ::-1]], axis=2)]).astype(np.uint8)
    got, want, care = _check(engine, frames)
    bits = _bits(got)[:, :1000]
    kt, kx = np.arange(1000) // 100, (np.arange(1000) // 10) % 10
    assert not bits[0][kt >= 1].any() and bits[0][:100].any()
    assert not bits[1].any() and not bits[3].any()            # all-black and mid-grey (pix - 128 <= 0): the empty hash
    assert bits[2][0] == 1 and not bits[2][1:].any()          # all-white: only the DC bit
    assert bits[4][0] == 1 and not bits[4][1:].any()
    small = np.stack([orc.resize_frame(f) for f in frames[5]])
    if np.array_equal(small, small[:, :, ::-1]):              # the resized frames are still mirror-symmetric
        assert not bits[5][kx % 2 == 1].any()


def test_batch_consistency(engine):
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256, size=(300, 16, 64, 64), dtype=np.uint8)
    allh = engine.hash_frames(frames)
    for i in (0, 17, 299):
        assert np.array_equal(engine.hash_frames(frames[i:i + 1])[0], allh[i])


@pytest.mark.parametrize("mode", [1, 3, 4])
@pytest.mark.parametrize("h,w", [(64, 64), (270, 480), (131, 67), (200, 136), (16, 16), (17, 300), (97, 150), (33, 129),
                                 (360, 640)])
def test_every_resize_kernel_matches_oracle(mode, h, w, monkeypatch):
    """Mode 1 = scalar fixed-point kernel, 3 = fused MFMA + DCT kernel,
    4 = MFMA per-frame kernel with whole-line (8 rows x 128 B) loads + DCT kernel (the default for tall frames).
    Odd widths exercise unaligned 16-byte loads and the end-of-buffer guard; 150 and 129 wide = an odd number of
    64-column K tiles (the wide kernel's last window has no odd half); 97 / 33 / 270 rows = partial quads."""
    import vid_dup_finder_lib_amd as vdf

    monkeypatch.setenv("VDF_RESIZE_MODE", str(mode))
    eng = vdf.Engine(0)
    try:
        rng = np.random.default_rng(1000 + mode + h * 7 + w)
        frames = rng.integers(0, 256, size=(5, 16, h, w), dtype=np.uint8)
        _check(eng, frames)
    finally:
        eng.close()


@pytest.mark.parametrize("mode", [0, 5])
@pytest.mark.parametrize("h,w,n", [(270, 480, 3), (129, 272, 45), (240, 320, 40), (288, 352, 3), (480, 720, 3), (540, 960, 2),
                                   (200, 1360, 2), (333, 1440, 2), (130, 496, 3), (1080, 1520, 1), (191, 464, 3),
                                   (360, 600, 3), (480, 854, 2), (240, 426, 40), (300, 500, 3), (256, 333, 3), (768, 1366, 1),
                                   (360, 640, 3), (432, 768, 2), (720, 1280, 1), (1080, 1920, 1), (300, 1984, 1), (300, 2000, 1),
                                   (426, 240, 40), (144, 176, 40), (200, 160, 3), (256, 192, 3), (256, 128, 3), (300, 200, 3), (333, 64, 3), (200, 80, 3),
                                   (900, 1600, 1), (576, 1024, 1), (136, 1440, 40), (150, 1920, 36), (1080, 1904, 1), (140, 1366, 40), (200, 1536, 24), (130, 1792, 24), (300, 1916, 2),
                                   (144, 1300, 36), (768, 1534, 1), (130, 640, 60), (150, 854, 60), (144, 1152, 40), (129, 768, 45), (140, 1024, 40),
                                   (144, 1950, 36), (300, 2340, 2), (200, 2001, 24), (1096, 1950, 1)])
def test_linear_stream_resize_kernel_matches_oracle(mode, h, w, n, monkeypatch):
    """Tightly packed frames whose every frame starts and ends on a 16-byte boundary go through the linear-stream kernels (LDS-DMA of
    whole chunks / blocks, operands read back from LDS).  Up to 512 wide = the chunk form, two workgroups per CU with 64-row chunks
    (480 / 272 / 320 / 352 / 464; 500 / 426 / 333 = rows re-pitched by the DMA to an odd multiple of 16 bytes, the last two with the
    0..3-byte operand shift of row starts that are not dword-aligned; 240 / 176 / 160 / 192 / 128 / 200 / 64 / 80 wide = narrow tall
    frames, one to four K tiles).  From there to 1920 wide = one block stream per wave (resize_mfma_frame_wavestream_kernel), the
    horizontal table in band form, with 8 waves per workgroup (496 ... 960), 6 (1024 ... 1300), 5 (1360 ... 1600) or 4 (... 1920):
    600 / 854 / 1366 / 1534 = re-pitched rows (854, 1366, 1534: shifted), 768 / 1024 / 1280 / 1536 / 1792 = multiples of 256 bytes,
    re-pitched to dodge the 16-way bank conflict; 45 / 40 / 36 clips = more frames than resident workgroups, so the persistent loops
    cross frame boundaries (the parity-doubled partial sums) with 9 or 10 blocks per frame (uneven shares of the waves, waves
    without a block, a partial last block); 270 / 129 / 333 / 191 rows = partial last chunks and blocks.  1916 / 1950 / 2001 / 2340 wide
    (not multiples of 16, pitches up to 2368) = three waves with 37 KB blocks; 1984 / 2000 = the K-split form."""
    import vid_dup_finder_lib_amd as vdf

    monkeypatch.setenv("VDF_RESIZE_MODE", str(mode))
    eng = vdf.Engine(0)
    try:
        rng = np.random.default_rng(5000 + h * 7 + w)
        frames = rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
        _check(eng, frames)
    finally:
        eng.close()


@pytest.mark.parametrize("h,w", [(48, 80), (64, 96), (80, 80), (96, 96), (112, 112), (72, 128), (96, 128), (128, 128), (128, 64), (96, 64), (100, 48),
                                 (63, 112), (128, 32), (100, 16), (17, 128), (65, 64), (64, 80), (65, 80), (127, 127 - 15), (128, 16), (16, 128),
                                 (81, 144), (90, 160), (120, 160), (99, 176), (108, 192), (64, 192), (33, 176), (128, 144), (128, 192), (65, 160),
                                 (117, 208), (126, 224), (64, 256), (96, 256), (128, 256), (100, 240), (17, 256), (48, 208),
                                 (160, 64), (256, 64), (160, 96), (160, 128), (256, 128), (200, 112), (240, 80), (144, 176), (200, 160), (208, 176), (129, 16),
                                 (256, 16), (192, 128), (130, 64), (193, 80), (256, 256), (144, 256), (192, 208),
                                 (44, 60), (56, 100), (68, 120), (84, 150), (112, 200), (120, 250), (200, 100), (60, 60), (62, 62), (50, 50), (17, 33),
                                 (100, 130), (250, 90), (129, 65), (64, 63), (65, 63), (128, 255), (256, 241)])  # widths off a multiple of 16: the last clip apart
def test_tiled_persistent_kernel_matches_oracle_and_the_per_clip_kernel(h, w, monkeypatch):
    """Round 5: frames of up to 256 x 256 (that are not a single 64 x 64 tile of a width that is a multiple of 16) can take
    resize_dct_hash_tiled_kernel - persistent workgroups, units of (at most) eight 16-byte loads per lane in two register buffers, the next
    clip's first unit in flight under the DCT.  All seven shapes (2 x 1, 1 x 2 tiles: a frame per unit; 2 x 2: a row group per unit; 3 and 4
    K tiles x 1 and 2 row groups: half a row group per unit, the block results carried between the halves; and the four-row-group forms of all
    four widths for 129 ... 256 rows, whose fourth group is empty up to 192 rows), partial tiles in both
    directions, more clips than resident workgroups (the persistent loop runs several times), the first clips against the oracle and every
    clip against the one-workgroup-per-clip kernel (VDF_HASH_NO_PERSISTENT) - through the fused family by force (VDF_RESIZE_MODE=3), since
    the default dispatch hands the largest of these sizes to the stream kernels.  Widths off a multiple of 16 (rows whose last load runs into
    the next row) go the same way, their last clip through the per-clip kernel's careful loader; the single-tile sizes among them take
    resize_dct_hash_persistent_kernel."""
    import torch

    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(7000 + h * 7 + w)
    n = 2000 if h * w <= 20000 else 1000
    frames = rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
    d = torch.from_numpy(frames).cuda()
    outs = {}
    for name, env in (("tiled", {"VDF_RESIZE_MODE": "3"}), ("per_clip", {"VDF_RESIZE_MODE": "3", "VDF_HASH_NO_PERSISTENT": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = vdf.Engine(0)
        for k in env:
            monkeypatch.delenv(k)
        try:
            if name == "tiled":
                _check(eng, frames[:48])
            out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            eng.hash_frames_device(d.data_ptr(), n, 16, w, h, out.data_ptr())
            torch.cuda.synchronize()
            outs[name] = out
        finally:
            eng.close()
    assert torch.equal(outs["tiled"], outs["per_clip"])


@pytest.mark.parametrize("mode", [0, 3, 5])
@pytest.mark.parametrize("h,w,n", [(128, 256, 70), (126, 224, 70), (117, 208, 70), (112, 200, 70), (120, 160, 70), (108, 192, 70), (96, 320, 70),
                                   (128, 480, 50), (120, 640, 50), (128, 854, 45), (128, 1280, 40), (128, 1920, 36), (64, 1920, 40), (100, 1366, 40),
                                   (65, 300, 70), (80, 240, 70), (64, 528, 60), (127, 150, 70), (90, 160, 70), (64, 512, 70), (48, 1920, 40), (72, 1000, 40)])
def test_short_wide_frames_stream_and_match_oracle(mode, h, w, n, monkeypatch):
    """Round 5: frames of at most 128 rows used to fuse resize and DCT whatever their width; the wide ones (resize_short_prefers_stream:
    more than 64 rows and 19 000 pixels, or wider than 512) now take the linear-stream kernels where they are eligible - frames of ONE or TWO
    chunks, of 4 ... 8 blocks, more clips than resident workgroups.  Default dispatch, the fused kernel and the stream kernels by force
    against the oracle; sizes on both sides of the rule."""
    import vid_dup_finder_lib_amd as vdf

    monkeypatch.setenv("VDF_RESIZE_MODE", str(mode))
    eng = vdf.Engine(0)
    try:
        rng = np.random.default_rng(6000 + h * 7 + w)
        frames = rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
        _check(eng, frames)
    finally:
        eng.close()


@pytest.mark.parametrize("mode", [0, 6])
@pytest.mark.parametrize("h,w,n", [(576, 1024, 2), (864, 1536, 1), (1152, 2048, 1), (1440, 2560, 1), (2160, 3840, 1), (300, 4096, 1),
                                   (333, 3008, 1), (130, 1040, 40), (720, 1280, 1), (1080, 1920, 1)])
def test_ksplit_stream_resize_kernel_matches_oracle(mode, h, w, n, monkeypatch):
    """Wide frames (1024..4096 columns, multiples of 16) that the stream kernel leaves out take its K-split form: the
    horizontal table in registers, the four waves of a workgroup sharing each 16-row block, partial sums through LDS behind
    an LDS-only barrier.  1024 / 1536 / 2048 wide = 64 / 48 / 32-row chunks, 2560..4096 = 16-row chunks (4 / 8 / 16 K
    tiles per wave), 3008 = a K-tile count that is not a multiple of 4, 40 clips of 130 rows = frame boundaries in the
    persistent loop and a 2-row last block; 1280 / 1920 run it only when forced (mode 6)."""
    import vid_dup_finder_lib_amd as vdf

    monkeypatch.setenv("VDF_RESIZE_MODE", str(mode))
    eng = vdf.Engine(0)
    try:
        rng = np.random.default_rng(6000 + h * 7 + w)
        frames = rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
        _check(eng, frames)
    finally:
        eng.close()


def test_linear_stream_kernel_publishes_its_partial_sums():
    """Regression: the per-frame partial sums of waves 1..3 cross a barrier that sits on the persistent loop's back edge, and
    the compiler emitted that barrier without the LDS wait; wave 0 then read stale sums in about one launch in ten when two
    workgroups shared a CU and the LDS pipe was busy (widths that are not a multiple of 16).  Fresh engines, 640 frames per
    launch, every launch must reproduce the scalar kernel's hashes."""
    import os

    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(422)
    frames = rng.integers(0, 256, size=(40, 16, 240, 422), dtype=np.uint8)

    def run(mode, calls):
        os.environ["VDF_RESIZE_MODE"] = str(mode)
        try:
            eng = vdf.Engine(0)
        finally:
            os.environ.pop("VDF_RESIZE_MODE", None)
        try:
            return [eng.hash_frames(frames) for _ in range(calls)]
        finally:
            eng.close()

    want = run(1, 1)[0]
    for rep in range(12):
        for got in run(0, 2):
            assert np.array_equal(got, want), rep


def test_full_hd_clip_matches_oracle(engine):
    """1080p (the size real decoders hand over): 68 row blocks, 15 windows of 128 bytes, all four waves busy."""
    rng = np.random.default_rng(1080)
    frames = rng.integers(0, 256, size=(2, 16, 1080, 1920), dtype=np.uint8)
    _check(engine, frames)


@pytest.mark.parametrize("h,w", [(64, 64), (48, 80), (120, 136), (270, 480), (301, 203), (300, 854), (200, 2048)])
@pytest.mark.parametrize("base,pad_f,pad_c", [(3, 37, 101), (16, 48, 112)])
def test_strided_and_misaligned_device_buffers(engine, h, w, base, pad_f, pad_c):
    """vdf_hash_frames_u8_device takes any frame_stride >= W*H, any clip_stride, any base alignment and ignores frames
    beyond the 16th (video_hash.rs:53): padded strides, 18 frames per clip and a base pointer 3 bytes off 16-byte alignment
    must hash exactly like the packed copy (covers every resize kernel's addressing and its end-of-buffer guard); with
    16-byte-aligned padding the linear-stream kernels still apply (frames start on 16-byte boundaries, rows are packed)."""
    import torch

    rng = np.random.default_rng(h * 1000 + w)
    n, nf = 5, 18
    frames = rng.integers(0, 256, size=(n, nf, h, w), dtype=np.uint8)
    if (w * h) % 16:  # keep every frame start 16-byte aligned in the aligned variant
        pad_f += 16 - (w * h + pad_f) % 16 if base == 16 else 0
    fs = w * h + pad_f
    cs = nf * fs + pad_c
    buf = np.full(base + (n - 1) * cs + (nf - 1) * fs + w * h, 0xAB, np.uint8)  # ends exactly at the last byte of the last frame
    for c in range(n):
        for f in range(nf):
            o = base + c * cs + f * fs
            buf[o:o + w * h] = frames[c, f].reshape(-1)
    d_buf = torch.from_numpy(buf).cuda()
    d_out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    engine.hash_frames_device(d_buf.data_ptr() + base, n, nf, w, h, d_out.data_ptr(), frame_stride=fs, clip_stride=cs)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(np.uint64)
    want = engine.hash_frames(frames[:, :16].copy())
    assert np.array_equal(got, want)
    _check(engine, frames[:, :16].copy())


def test_uhd_clip_matches_oracle(engine):
    """3840 x 2160: 68 quads, 30 windows; also the largest coefficient windows the tests see (405 vertical taps)."""
    rng = np.random.default_rng(2160)
    frames = rng.integers(0, 256, size=(1, 16, 2160, 3840), dtype=np.uint8)
    _check(engine, frames)
