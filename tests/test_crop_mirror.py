"""The host-side `Crop` type (vid_dup_finder_common/src/crop.rs) - the reference's own unit tests of it, restated on the mirror
(crop.rs:204-360: as_view_args, from_topleft_and_dims, enumerate_coords / _excluded), plus union, the constructor's checks and the
C ABI's {left, right, top, bottom} row.  CPU only."""
import numpy as np
import pytest

from vid_dup_finder_lib_amd import Crop


@pytest.mark.parametrize("res, edges, exp", [
    ((100, 100), (0, 0, 0, 0), (0, 0, 100, 100)),       # test_as_view_args_nocrop
    ((100, 100), (1, 0, 0, 0), (1, 0, 99, 100)),        # .._1pix_left
    ((100, 100), (0, 1, 0, 0), (0, 0, 99, 100)),        # .._1pix_right
    ((100, 100), (0, 0, 1, 0), (0, 1, 100, 99)),        # .._1pix_top
    ((100, 100), (0, 0, 0, 1), (0, 0, 100, 99)),        # .._1pix_bot
    ((100, 100), (25, 25, 25, 25), (25, 25, 50, 50)),   # .._four_values
    ((768, 432), (96, 96, 0, 0), (96, 0, 576, 432)),    # .._four_more
])
def test_as_view_args(res, edges, exp):
    assert Crop.from_edge_offsets(res, *edges).as_view_args() == exp


def test_from_offset_and_dims():
    assert Crop.from_topleft_and_dims((100, 100), 11, 12, 13, 14).as_view_args() == (11, 12, 13, 14)


def test_nothing_left_is_refused():
    with pytest.raises(AssertionError):  # crop.rs:21-22 (the reference's commented-out should_panic case)
        Crop.from_edge_offsets((100, 100), 50, 50, 50, 50)
    with pytest.raises(AssertionError):
        Crop.from_edge_offsets((100, 100), 0, 0, 60, 40)
    with pytest.raises(OverflowError):
        Crop.from_topleft_and_dims((10, 10), 5, 0, 6, 1)


def test_enumerate_coords_nocrop():
    crop = Crop.from_edge_offsets((3, 3), 0, 0, 0, 0)
    assert len(list(crop.enumerate_coords())) == 9 and len(list(crop.enumerate_coords_excluded())) == 0


@pytest.mark.parametrize("edges, inside", [
    ((1, 1, 1, 1), (1, 1)),   # test_enumerate_coords_1pixinthemiddle
    ((1, 1, 0, 2), (1, 0)),   # .._1pixinthetop
    ((2, 0, 2, 0), (2, 2)),   # .._1pixintheright
])
def test_enumerate_coords_one_pixel(edges, inside):
    crop = Crop.from_edge_offsets((3, 3), *edges)
    assert list(crop.enumerate_coords()) == [inside]
    everything = {(x, y) for x in range(3) for y in range(3)}
    excluded = list(crop.enumerate_coords_excluded())
    assert sorted(excluded) == sorted(everything - {inside}) and len(excluded) == 8


def test_alternate_constructor_agrees():
    assert Crop.from_edge_offsets((3, 3), 2, 0, 2, 0) == Crop.from_topleft_and_dims((3, 3), 2, 2, 1, 1)


def test_enumeration_orders():
    crop = Crop.from_edge_offsets((4, 5), 1, 1, 1, 2)
    assert list(crop.enumerate_coords()) == [(1, 1), (1, 2), (2, 1), (2, 2)]  # x outermost
    # clockwise from the top left: tl, tm, tr, mr, bl, bm, br, ml
    assert list(crop.enumerate_coords_excluded())[:4] == [(0, 0), (1, 0), (2, 0), (3, 0)]
    assert crop.width() == 2 and crop.height() == 2 and crop.area() == 4 and crop.aspect_ratio() == 1.0


def test_union_is_the_per_edge_minimum():
    a = Crop.from_edge_offsets((100, 80), 10, 0, 5, 7)
    b = Crop.from_edge_offsets((100, 80), 3, 4, 9, 2)
    assert a.union(b) == Crop.from_edge_offsets((100, 80), 3, 0, 5, 2) == b.union(a)
    # folding from the 'enormous' default (crop.rs:183-194) leaves the real crops' minimum
    acc = Crop.default()
    for c in (a, b):
        acc = Crop(c.orig_res, *acc.union(Crop(acc.orig_res, c.left, c.right, c.top, c.bottom)).as_abi())
    assert acc == a.union(b)


def test_eroded_and_uncropped():
    assert Crop.from_edge_offsets((3, 3), 0, 0, 0, 0).is_uncropped()
    assert Crop.from_edge_offsets((3, 3), 0, 0, 0, 0).eroded() == Crop.from_edge_offsets((3, 3), 1, 1, 1, 1)
    assert Crop.from_edge_offsets((3, 3), 1, 1, 1, 1).eroded() is None
    assert Crop.from_edge_offsets((4, 9), 0, 0, 0, 0).eroded().eroded() is None  # two columns left: the next step would leave none


def test_abi_row_round_trip_and_order():
    row = np.array([96, 96, 0, 0], np.uint32)
    c = Crop.from_abi((768, 432), row)
    assert c.as_view_args() == (96, 0, 576, 432) and (c.as_abi() == row).all()
    assert sorted([Crop((2, 2), 1, 0, 0, 0), Crop((2, 2), 0, 1, 0, 0), Crop((1, 9), 5, 5, 5, 5)])[0].orig_res == (1, 9)  # derive(Ord): orig_res first
    assert len({c, Crop.from_abi((768, 432), row)}) == 1
