"""The hand-written stable radix sort behind Search::sort on the device (csrc/sort_order.hip; search_algorithm.rs:55-61) on its own:
tile boundaries, adversarial key distributions and REPETITION - its scatter pass chains the tiles of a launch by decoupled look-back (a tile
spins on the status words of the tiles in front of it), which is exactly the kind of code whose bugs show once in a thousand launches.
Every result against numpy's stable sorts."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(engine, dur, rank):
    n = len(dur)
    t_d = torch.from_numpy(dur.view(np.int32).copy()).cuda()
    perm = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    engine.sort_order_device(t_d.data_ptr(), n, perm.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(perm.cpu().numpy().view(np.uint32), np.argsort(dur, kind="stable").astype(np.uint32)), "durations only"
    if rank is not None:
        t_r = torch.from_numpy(rank.view(np.int32).copy()).cuda()
        perm.fill_(-1)
        torch.cuda.synchronize()
        engine.sort_order_device(t_d.data_ptr(), n, perm.data_ptr(), d_path_rank=t_r.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(perm.cpu().numpy().view(np.uint32), np.lexsort((rank, dur)).astype(np.uint32)), "(duration, rank)"


DISTRIBUTIONS = ["random32", "all_equal", "two_values", "sorted", "reversed", "high_byte_only", "low_byte_only", "max_keys", "seconds"]


def _keys(rng, kind, n):
    if kind == "random32":
        return rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    if kind == "all_equal":
        return np.full(n, 0x01020304, np.uint32)
    if kind == "two_values":
        return np.where(rng.random(n) < 0.5, np.uint32(7), np.uint32(0x80000007)).astype(np.uint32)
    if kind == "sorted":
        return np.sort(rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32))
    if kind == "reversed":
        return np.sort(rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32))[::-1].copy()
    if kind == "high_byte_only":
        return (rng.integers(0, 256, size=n).astype(np.uint32) << np.uint32(24))
    if kind == "low_byte_only":
        return rng.integers(0, 256, size=n).astype(np.uint32)
    if kind == "max_keys":
        return np.where(rng.random(n) < 0.9, np.uint32(0xFFFFFFFF), np.uint32(0)).astype(np.uint32)
    return np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=n))).astype(np.uint32)  # durations as a cache holds them


@pytest.mark.parametrize("kind", DISTRIBUTIONS)
@pytest.mark.parametrize("n", [3, 4095, 4096, 4097, 8192, 12289, 65536, 250_001])
def test_sizes_around_the_tiles_and_hostile_keys(engine, n, kind):
    rng = np.random.default_rng(n * 31 + len(kind))
    dur = _keys(rng, kind, n)
    rank = _keys(rng, DISTRIBUTIONS[(DISTRIBUTIONS.index(kind) + 3) % len(DISTRIBUTIONS)], n)
    _check(engine, dur, rank)


def test_many_launches_of_many_tiles(engine):
    """160 sorts of 100 k ... 1.2 M keys (25 ... 300 tiles chained per pass) back to back, and three of 6 M keys: a look-back that reads a
    status word too early, a ticket handed out twice or a stale word of the previous pass would show as a wrong permutation."""
    rng = np.random.default_rng(2026)
    for rep in range(160):
        n = int(rng.integers(100_000, 1_200_000))
        kind = DISTRIBUTIONS[rep % len(DISTRIBUTIONS)]
        dur = _keys(rng, kind, n)
        _check(engine, dur, _keys(rng, "random32", n) if rep % 4 == 0 else None)
    for _ in range(3):
        dur = _keys(rng, "seconds", 6_000_000)
        _check(engine, dur, rng.permutation(6_000_000).astype(np.uint32))
