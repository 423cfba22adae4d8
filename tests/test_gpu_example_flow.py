"""Example-as-test, the analogue of vid_dup_finder_lib/examples/example.rs:26-83 (3 cat + 3 dog videos -> 2 groups
of 3 at the default tolerance) without a decoder: two synthetic "videos" (smooth space-time noise), each in three
variants - original, rescaled to another resolution with noise, and letterboxed - go through the post-decode half of
gen_hash (letterbox crop detection + VideoHash construction) and search()."""
import numpy as np
import pytest
from scipy.ndimage import gaussian_filter, zoom

pytestmark = pytest.mark.gpu


def _video(rng, h, w):
    v = gaussian_filter(rng.standard_normal((16, h, w)), sigma=(2.0, h / 10, w / 10))
    v = (v - v.min()) / (v.max() - v.min())
    return (30 + v * 200).astype(np.uint8)


def _variants(rng, base):
    h, w = base.shape[1:]
    small = zoom(base.astype(np.float64), (1, 0.5, 0.5), order=1)
    small = np.clip(small + rng.normal(0, 3, small.shape), 0, 255).astype(np.uint8)  # re-encode noise
    boxed = np.full((16, h + 2 * (h // 6), w), 16, np.uint8)                          # black bars top and bottom
    boxed[:, h // 6:h // 6 + h, :] = np.clip(base.astype(np.int16) + rng.integers(-2, 3, base.shape), 0, 255)
    return [base, small, boxed]


def test_two_videos_three_variants_each(engine):
    import vid_dup_finder_lib_amd as vdf

    rng = np.random.default_rng(2025)
    hashes = []
    for name in ("cat", "dog"):
        base = _video(rng, 96, 128)
        for k, clip in enumerate(_variants(rng, base)):
            hashes += vdf.gen_hashes(clip[None], [f"{name}.{k + 1}.mp4"], [30], engine=engine)  # Cropdetect::Letterbox
    # the letterboxed variant must have been cropped back to the picture
    crops = engine.hash_frames_letterbox(_variants(rng, _video(rng, 96, 128))[2][None])[1]
    assert tuple(crops[0]) == (0, 0, 16, 16)
    groups = vdf.search(hashes, vdf.DEFAULT_SEARCH_TOLERANCE, engine=engine)
    assert len(groups) == 2 and all(g.len() == 3 for g in groups)  # example.rs:78-82
    names = sorted(sorted(str(p).split(".")[0] for p in g.duplicates()) for g in groups)
    assert names == [["cat"] * 3, ["dog"] * 3]
    # variants of one video are far closer than different videos
    d_same = hashes[0].hamming_distance(hashes[1])
    d_diff = hashes[0].hamming_distance(hashes[3])
    assert d_same <= 350 < d_diff, (d_same, d_diff)  # inside / outside the default tolerance (definitions.rs:5)
    # search_with_references: one reference per video finds its two other variants (lib.rs doc example shape)
    refs = [hashes[0], hashes[3]]
    rest = hashes[1:3] + hashes[4:6]
    rg = vdf.search_with_references(refs, rest, 0.35, engine=engine)
    assert [(g.reference(), g.len()) for g in rg] == [("cat.1.mp4", 2), ("dog.1.mp4", 2)]


def test_hash_then_reference_search_end_to_end(engine):
    """BASELINE configs[4] shape in miniature (world size 1): hash candidate and reference clips on the device, sort,
    search_with_references; every reference that is a noisy copy of a candidate must find exactly that candidate."""
    import torch

    from oracle import vdf_oracle as orc
    from vid_dup_finder_lib_amd import distributed as vd

    rng = np.random.default_rng(9)
    n_cand, n_ref = 300, 60
    cand = np.stack([_video(rng, 32, 48) for _ in range(n_cand)])
    src = rng.choice(n_cand, size=n_ref // 2, replace=False)
    ref = np.concatenate([np.clip(cand[src].astype(np.int16) + rng.integers(-4, 5, cand[src].shape), 0, 255).astype(np.uint8),
                          np.stack([_video(rng, 32, 48) for _ in range(n_ref - n_ref // 2)])])
    cd = rng.integers(100, 120, size=n_cand).astype(np.int32)
    rd = np.concatenate([cd[src], rng.integers(100, 120, size=n_ref - n_ref // 2).astype(np.int32)])
    groups, order = vd.hash_and_search_refs(engine, torch.from_numpy(cand).cuda(), torch.from_numpy(cd).cuda(),
                                            torch.from_numpy(ref).cuda(), torch.from_numpy(rd).cuda(), 350)
    found = {r: [int(order[m]) for m in ms] for r, ms in groups}
    for k, s in enumerate(src):
        assert int(s) in found.get(k, []), (k, s)
    # the same thing through the oracle (hash on CPU, search on CPU)
    ch, rh = orc.hash_clips(cand), orc.hash_clips(ref)
    o = np.argsort(cd, kind="stable")
    want = orc.search_refs_sorted(ch[o], cd[o].astype(np.uint32), rh, rd.astype(np.uint32), 350)
    assert [(r, [int(o[m]) for m in ms]) for r, ms in want] == [(r, found[r]) for r, _ in want] and len(want) == len(groups)
