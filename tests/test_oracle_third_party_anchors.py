"""Independent anchors for the two pieces of the oracle whose arithmetic lives in third-party crates that are not
under /root/reference: fast_image_resize 5.x (resize_gray.rs:34-47) and rustdct 0.7
(raw_dct_ops.rs:114-131).

Neither crate can run here, so these are NOT parity pins (the oracle header keeps saying "parity unpinned" for the hash
bits).  They bound how far the restatement can be from the published algorithms using implementations that ARE here:

* Pillow's `Image.resize(..., LANCZOS)` is the same separable Lanczos3 u8 fixed-point convolution (support 3*max(scale,1),
  weights normalised per output pixel, horizontal pass then vertical pass through a u8 intermediate), only with 22-bit
  coefficients instead of fast_image_resize's <=15-bit i16 ones => the two must agree to +-1 LSB, and nearly everywhere on smooth
  images.
* scipy.fft.dct(type=2, norm=None) is 2x the unnormalised DCT-II rustdct computes; signs (all the hash consumes) match
  wherever the coefficient is not mathematically zero.
"""
import numpy as np
import pytest

from oracle import vdf_oracle as orc

PIL = pytest.importorskip("PIL.Image")
scipy_fft = pytest.importorskip("scipy.fft")


def pillow_resize(frame):
    return np.asarray(PIL.fromarray(frame, mode="L").resize((16, 16), PIL.Resampling.LANCZOS))


@pytest.mark.parametrize("w,h", [(64, 64), (128, 72), (320, 240), (17, 33), (1920, 1080), (16, 48), (8, 8), (12, 20)])
def test_resize_within_one_lsb_of_pillow_on_noise(w, h):
    rng = np.random.default_rng(w * 10007 + h)
    worst = 0
    differing = 0
    total = 0
    for _ in range(4):
        frame = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        a = orc.resize_frame(frame).astype(np.int32)
        b = pillow_resize(frame).astype(np.int32)
        worst = max(worst, int(np.abs(a - b).max()))
        differing += int((a != b).sum())
        total += a.size
    assert worst <= 1, f"{w}x{h}: oracle resize is {worst} LSB away from Pillow's Lanczos3"
    # 15-bit vs 22-bit coefficient rounding only moves results that sit on a rounding boundary
    assert differing / total < 0.05


@pytest.mark.parametrize("w,h", [(64, 64), (200, 120), (640, 360)])
def test_resize_matches_pillow_closely_on_smooth_images(w, h):
    y, x = np.mgrid[0:h, 0:w]
    frame = (127.5 + 100 * np.sin(x / w * 5.0) * np.cos(y / h * 3.0)).astype(np.uint8)
    a = orc.resize_frame(frame).astype(np.int32)
    b = pillow_resize(frame).astype(np.int32)
    assert np.abs(a - b).max() <= 1
    assert (a != b).mean() < 0.02


def test_numpy_twin_equals_c_oracle_resize_on_the_same_sizes():
    rng = np.random.default_rng(5)
    for w, h in [(64, 64), (128, 72), (17, 33), (8, 8)]:
        frame = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        assert np.array_equal(orc.resize_frame(frame), orc.np_resize_frame(frame))


def scipy_dct3(cube_txy):
    """Unnormalised 3-D DCT-II of a [t][x][y] cube (scipy's type-2 is 2x rustdct's per axis)."""
    return scipy_fft.dctn(cube_txy, type=2, norm=None) / 8.0


def test_dct3d_equals_scipy_dctn():
    rng = np.random.default_rng(9)
    for _ in range(8):
        cube = rng.integers(0, 256, size=(16, 16, 16)).astype(np.float64) - 128.0
        mine = orc.dct3d(cube)
        ref = scipy_dct3(cube)
        assert np.abs(mine - ref).max() < 1e-7  # values reach ~1e5; f64 butterfly-order noise only


def test_hash_bits_equal_scipy_signs_outside_dont_care():
    rng = np.random.default_rng(10)
    frames = rng.integers(0, 256, size=(64, 16, 16, 16), dtype=np.uint8)
    for clip in frames:
        rc, words, _ = orc.hash_clip(clip, want_coefs=True)
        assert rc == 0
        # dct_3d.rs:40-44: cube axes are [t][x][y]; frames arrive [t][row=y][col=x]
        cube = clip.transpose(0, 2, 1).astype(np.float64) - 128.0
        ref = scipy_dct3(cube)[:10, :10, :10].reshape(-1)
        care = np.abs(ref) >= 1e-6
        bits = np.unpackbits(np.asarray(words, dtype=np.uint64).view(np.uint8), bitorder="little")
        assert not bits[1000:].any()  # padding bits stay zero when built from frames
        bits = bits[:1000]
        assert np.array_equal(bits[care].astype(bool), (ref > 0.0)[care])
