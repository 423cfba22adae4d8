"""Synthetic hash fixtures: the build's own restatement of the reference's test utilities
(vid_dup_finder_lib/src/video_hashing/video_hash.rs:240-308 `test_util`, and the scenario builders of
vid_dup_finder_lib/tests/test_find_all.rs:14-132) with numpy's PCG64 instead of rand 0.9's StdRng.
Every assertion upstream is structural (group counts and sizes), so the construction - not the
random stream - is what has to match.
"""
from __future__ import annotations

import numpy as np

HASH_WORDS = 16
HASH_BITS = 1000


def random_hash(rng: np.random.Generator) -> np.ndarray:
    """1000 fair bits, padding bits 1000..1023 zero (video_hash.rs:293-306)."""
    bits = rng.integers(0, 2, size=1024, dtype=np.uint8)
    bits[HASH_BITS:] = 0
    return np.packbits(bits, bitorder="little").view(np.uint64).copy()


def random_hashes(rng: np.random.Generator, n: int) -> np.ndarray:
    words = rng.integers(0, 2**64, size=(n, HASH_WORDS), dtype=np.uint64)
    words[:, 15] &= np.uint64((1 << 40) - 1)  # bits 960..999 live in word 15; 1000..1023 cleared
    return words


def hamming(a: np.ndarray, b: np.ndarray) -> int:
    return int(np.unpackbits((a ^ b).view(np.uint8)).sum())


def hash_with_spatial_distance(h: np.ndarray, target: int, rng: np.random.Generator) -> np.ndarray:
    """A hash at EXACTLY `target` bits from `h`, bits chosen anywhere in the 1024 (padding included).
    Upstream (video_hash.rs:272-291) random-walks single bit flips until the distance first reaches
    `target`; by symmetry the first-hit point is uniform on the sphere of that radius, which is what
    flipping `target` distinct random positions samples directly.  (The walk itself is hopeless in Python
    for radius 600 > the 512-bit equilibrium, which test_find_with_refs needs.)"""
    flips = rng.choice(1024, size=target, replace=False)
    bits = np.unpackbits(h.view(np.uint8), bitorder="little")
    bits[flips] ^= 1
    out = np.packbits(bits, bitorder="little").view(np.uint64).copy()
    assert hamming(h, out) == target
    return out


class HashesWithDistance:
    """test_find_all.rs:14-60"""

    def __init__(self, start_hash, distance, num_hashes, rng):
        self.start_hash = start_hash
        self.members_ = [hash_with_spatial_distance(start_hash, distance, rng) for _ in range(num_hashes)]
        self.distance = distance

    def members(self, rng):
        idx = rng.permutation(len(self.members_))
        return [self.members_[i] for i in idx]


class HashesWithDistanceSet:
    """test_find_all.rs:62-132"""

    def __init__(self, num_groups, hashes_per_group, intergroup_distance, intragroup_distance, rng):
        assert intragroup_distance * 2 < intergroup_distance
        assert (19 * 64) // num_groups > intergroup_distance
        start = random_hash(rng)
        self.groups = []
        cur = 0
        for _ in range(num_groups):
            g_start = hash_with_spatial_distance(start, cur, rng)
            cur += intergroup_distance
            self.groups.append(HashesWithDistance(g_start, intragroup_distance, hashes_per_group, rng))
            hashes_per_group += 10

    def all_members(self, rng):
        allm = [m for g in self.groups for m in g.members(rng)]
        idx = rng.permutation(len(allm))
        return [allm[i] for i in idx]


def planted_set(rng: np.random.Generator, n: int, n_clusters: int, max_copies: int = 4, max_flips: int = 350,
                durations: str = "zero"):
    """Random hashes with planted near-duplicates: some copies land exactly on or over
    the tolerance.  Returns (hashes [n,16] u64, durations [n] u32), unsorted."""
    words = random_hashes(rng, n)
    if durations == "zero":
        dur = np.zeros(n, np.uint32)
    else:
        dur = np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=n))).astype(np.uint32)
    src = rng.choice(n, size=min(n_clusters, n), replace=False)
    free = np.setdiff1d(np.arange(n), src)
    rng.shuffle(free)
    pos = 0
    for s in src:
        for _ in range(int(rng.integers(1, max_copies + 1))):
            if pos >= len(free):
                break
            t = free[pos]
            pos += 1
            k = int(rng.integers(0, max_flips + 30))
            flips = rng.choice(1024, size=k, replace=False)
            bits = np.unpackbits(words[s].view(np.uint8), bitorder="little")
            bits[flips] ^= 1
            words[t] = np.packbits(bits, bitorder="little").view(np.uint64)
            if durations != "zero":
                jitter = rng.uniform(0.9, 1.12)
                dur[t] = np.uint32(max(0, int(dur[s] * jitter)))
    return words, dur


def sort_by_duration(words: np.ndarray, dur: np.ndarray):
    """Search::sort with all paths equal: stable by duration."""
    order = np.argsort(dur, kind="stable")
    return words[order], dur[order], order
