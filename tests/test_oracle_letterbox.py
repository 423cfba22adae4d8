"""The reference's known-answer tests for letterbox detection, restated against the oracle
(vid_dup_finder_common/src/video_frames_gray.rs:216-459; only the AnyColour arm is on the product path:
cropdetect_letterbox uses AnyColour(16), :201-210).  Crop tuples are (left, right, top, bottom)."""
import numpy as np

from oracle import vdf_oracle as orc


def _img(w, h, pix):
    return np.array(pix, np.uint8).reshape(h, w)


def test_white_and_black_images_find_no_crop():  # :224-275: all strips are letterbox -> edges converge -> zero crop
    assert orc.letterbox_crop(_img(3, 3, [255] * 9), 1) == (0, 0, 0, 0)
    assert orc.letterbox_crop(_img(3, 3, [0] * 9), 1) == (0, 0, 0, 0)


def test_any_colour_gray():  # :277-303
    assert orc.letterbox_crop(_img(3, 3, [127, 127, 127, 127, 0, 127, 127, 127, 127]), 1) == (1, 1, 1, 1)


def test_any_threshold():  # :305-330: mode 120, |130 - 120| = 10
    pix = [120, 130, 120, 130, 0, 130, 120, 130, 120]
    assert orc.letterbox_crop(_img(3, 3, pix), 9) == (0, 0, 0, 0)
    assert orc.letterbox_crop(_img(3, 3, pix), 10) == (1, 1, 1, 1)


def test_onepix():  # :332-358
    assert orc.letterbox_crop(_img(3, 3, [0, 0, 0, 0, 127, 0, 0, 0, 0]), 1) == (1, 1, 1, 1)


def test_topcorner():  # :360-386
    assert orc.letterbox_crop(_img(3, 3, [127, 0, 0, 0, 0, 0, 0, 0, 0]), 1) == (0, 2, 0, 2)


def test_rightedge():  # :388-414
    assert orc.letterbox_crop(_img(3, 3, [0, 0, 200, 0, 0, 120, 0, 0, 100]), 1) == (2, 0, 0, 0)


def test_bottom_right_2pix():  # :416-442
    assert orc.letterbox_crop(_img(3, 3, [0, 0, 0, 0, 127, 0, 0, 0, 127]), 1) == (1, 0, 1, 0)


def test_2pix_bottom():  # :444-459
    pix = [0, 0, 0, 0, 0,
           0, 255, 255, 255, 0,
           0, 255, 255, 255, 0,
           0, 255, 255, 255, 0,
           0, 0, 0, 0, 0,
           0, 0, 0, 0, 0]
    assert orc.letterbox_crop(_img(5, 6, pix), 1) == (1, 1, 1, 2)


def test_clip_level_union_uses_frames_0_and_8():  # :201-210 + crop.rs:53-68 (per-edge minimum)
    rng = np.random.default_rng(0)
    frames = rng.integers(60, 200, size=(16, 40, 56), dtype=np.uint8)
    frames[:, :6, :] = 16   # 6-row bar on top everywhere
    frames[:, -4:, :] = 16  # 4-row bar at the bottom
    frames[0, :, :5] = 16   # frame 0 also has a 5-column left bar; frame 8 does not
    assert orc.letterbox_crop(frames[0]) == (5, 0, 6, 4)
    assert orc.letterbox_crop(frames[8]) == (0, 0, 6, 4)
    assert orc.cropdetect_letterbox(frames) == (0, 0, 6, 4)
    frames[3, :, :] = 0  # frames other than 0 and 8 are never looked at
    assert orc.cropdetect_letterbox(frames) == (0, 0, 6, 4)
    rc, h, _, crop = orc.hash_clip_letterbox(frames)
    assert rc == 0 and crop == (0, 0, 6, 4)
    rc2, h2, _ = orc.hash_clip(np.ascontiguousarray(frames[:, 6:-4, :]))
    assert rc2 == 0 and np.array_equal(h, h2)
