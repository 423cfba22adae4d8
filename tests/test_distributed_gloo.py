"""world_size-2 run of the multi-GPU orchestration on CPU (gloo): database all-gather with uneven shards,
round-robin row tiles, hit gather/merge on rank 0, consumption-bitmap broadcast after a hit-buffer overflow,
contiguous reference split.  The device kernel is replaced by a numpy stand-in that honours the same contract
as Engine.search_self_device / search_refs_device (the real one is covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TILE = 64  # stand-in row tile


class FakeEngine:
    """Contract of Engine.search_*_device on host tensors (data_ptr is ignored: arrays are captured)."""

    def __init__(self, words, dur, ref_words=None, ref_dur=None):
        self.w, self.d, self.rw, self.rd = words, dur, ref_words, ref_dur
        self.bits = np.unpackbits(words.view(np.uint8), axis=1).astype(np.int16)
        self.calls = 0

    def search_self_device(self, d_hashes, d_durations, n, tol_int, shard_index=0, shard_count=1, row_begin=0,
                           row_end=0xFFFFFFFF, d_matched=0, capacity=1 << 22, stream=0):
        self.calls += 1
        matched = self.matched_bits  # set by the harness before each call
        hits = []
        for i in range(n):
            if (i // TILE) % shard_count != shard_index or i < row_begin or i >= row_end:
                continue
            if matched is not None and (matched[i >> 5] >> (i & 31)) & 1:
                continue
            thresh = min(int(float(self.d[i]) * 1.1), 2**32 - 1)
            hi = int(np.searchsorted(self.d, np.uint32(thresh), side="right"))
            if hi <= i + 1:
                continue
            dist_ = (self.bits[i + 1:hi] != self.bits[i]).sum(axis=1)
            for j in np.nonzero(dist_ <= tol_int)[0]:
                c = i + 1 + int(j)
                if matched is not None and (matched[c >> 5] >> (c & 31)) & 1:
                    continue
                hits.append((i, c))
        n_hits = len(hits)
        overflow = 0xFFFFFFFF
        if n_hits > capacity:  # emulate the device: an arbitrary subset survives, smallest lost row is reported
            rng = np.random.default_rng(n_hits)
            keep = set(rng.choice(n_hits, size=capacity, replace=False).tolist())
            overflow = min(h[0] for k, h in enumerate(hits) if k not in keep)
            hits = [h for k, h in enumerate(hits) if k in keep]
        arr = np.array(sorted(hits), np.uint32).reshape(-1, 2)
        return arr, n_hits, overflow

    def search_self_device_replay(self, d_hashes, d_durations, n, tol_int, shard_index=0, shard_count=1, row_begin=0,
                                  row_end=0xFFFFFFFF, d_matched=0, capacity=1 << 22, stream=0, exchange=None):
        """Contract of Engine.search_self_device_replay (vdf_search_self_device_replay): the shards agree that nobody overflowed,
        OR their has-incoming / covered bitmaps through the exchange, and keep only the hits of rows that can become targets.
        The bitmaps live in host memory here; the exchange reaches them through bitmap_or_device like the real engine's."""
        arr, n_hits, overflow = self.search_self_device(d_hashes, d_durations, n, tol_int, shard_index, shard_count, row_begin,
                                                        row_end, d_matched, capacity, stream)
        complete, total = (overflow == 0xFFFFFFFF and n_hits <= capacity), n_hits
        if exchange is not None:
            complete, total = exchange.agree(complete, total)
        elif shard_count != 1:
            complete = False
        if not (complete and total >= 1):
            return arr, n_hits, overflow
        words = (n + 31) // 32
        has_in, covered = np.zeros(words, np.uint32), np.zeros(words, np.uint32)
        bit = lambda bm, i: (int(bm[i >> 5]) >> (i & 31)) & 1  # noqa: E731
        for _, c in arr:
            has_in[c >> 5] |= np.uint32(1 << (c & 31))
        if exchange is not None:
            exchange.or_bitmap(self, has_in.ctypes.data, words, 0)
        for r, c in arr:
            if not bit(has_in, int(r)):
                covered[c >> 5] |= np.uint32(1 << (c & 31))
        if exchange is not None:
            exchange.or_bitmap(self, covered.ctypes.data, words, 0)
        keep = np.array([h for h in arr if not bit(covered, int(h[0]))], np.uint32).reshape(-1, 2)
        self.filtered = getattr(self, "filtered", 0) + len(arr) - len(keep)
        return keep, len(keep), overflow

    def bitmap_or_device(self, d_dst, d_srcs, n_words, n_srcs, stream=0):
        import ctypes as C

        dst = np.ctypeslib.as_array((C.c_uint32 * n_words).from_address(d_dst))
        src = np.ctypeslib.as_array((C.c_uint32 * (n_words * n_srcs)).from_address(d_srcs)).reshape(n_srcs, n_words)
        dst |= np.bitwise_or.reduce(src, axis=0)

    def search_refs_device(self, d_cand_hashes, d_cand_durations, n_cand, d_ref_hashes, d_ref_durations, n_ref,
                           tol_int, ref_index_base=0, capacity=1 << 22, stream=0):
        rbits = np.unpackbits(self.rw.view(np.uint8), axis=1).astype(np.int16)
        hits = []
        for r in range(n_ref):
            lo = int(np.searchsorted(self.d, np.uint32(int(float(self.rd[r]) * 0.95)), side="left"))
            hi = int(np.searchsorted(self.d, np.uint32(min(int(float(self.rd[r]) * 1.05), 2**32 - 1)), side="right"))
            if hi <= lo:
                continue
            dist_ = (self.bits[lo:hi] != rbits[r]).sum(axis=1)
            hits += [(r + ref_index_base, lo + int(j)) for j in np.nonzero(dist_ <= tol_int)[0]]
        return np.array(hits, np.uint32).reshape(-1, 2), len(hits)


def _worker(rank, world, port, capacity, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hashgen as hg
    from vid_dup_finder_lib_amd import distributed as vd

    rng = np.random.default_rng(77)
    words, dur = hg.planted_set(rng, 700, n_clusters=30, max_copies=10, max_flips=200, durations="windowed")
    w, d, _ = hg.sort_by_duration(words, dur)
    # uneven shards in rank order; the all-gather must hand every rank the same full, ordered database
    if world == 2:
        cut = 301
        lo, hi = (0, cut) if rank == 0 else (cut, len(d))
    else:  # eight ranks: uneven cuts, one EMPTY shard (a rank that hashed nothing)
        cuts = [0, 90, 90, 301, 340, 512, 600, 689, len(d)]
        lo, hi = cuts[rank], cuts[rank + 1]
    fw, fd = vd.all_gather_database(torch.from_numpy(w[lo:hi].view(np.int64)), torch.from_numpy(d[lo:hi].view(np.int32)))
    assert np.array_equal(fw.numpy().view(np.uint64), w) and np.array_equal(fd.numpy().view(np.uint32), d)

    eng = FakeEngine(w, d)
    eng.matched_bits = None
    spy = eng
    # monkeypatch Tensor.to so the stand-in sees the matched bitmap that would be uploaded
    import vid_dup_finder_lib_amd.distributed as mod

    orig_to = torch.Tensor.to

    def spy_to(self, *a, **kw):
        out = orig_to(self, *a, **kw)
        if self.dtype == torch.int32 and self.dim() == 1 and self.numel() == (len(d) + 31) // 32:
            eng.matched_bits = self.numpy().view(np.uint32).copy()
        return out

    torch.Tensor.to = spy_to
    try:
        st = {}
        groups = mod.search_self_sharded(spy, fw, fd, 350, capacity=capacity, stats=st)
        np.save(os.path.join(out_dir, f"filter_{rank}.npy"), np.array([getattr(eng, "filtered", 0), st["filtered_launches"], st["hits_downloaded"]]))
    finally:
        torch.Tensor.to = orig_to
    # references: contiguous split, results concatenated in rank order on rank 0
    rrng = np.random.default_rng(5)
    pick = rrng.choice(len(d), size=41, replace=False)
    rw, rd = w[pick].copy(), d[pick].copy()
    a, b = vd.split_range(len(rd), rank, world)
    eng_r = FakeEngine(w, d, rw[a:b], rd[a:b])
    refs = mod.search_refs_sharded(eng_r, fw, fd, torch.from_numpy(rw[a:b].view(np.int64)),
                                   torch.from_numpy(rd[a:b].view(np.int32)), a, 300)
    if rank == 0:
        np.save(os.path.join(out_dir, "calls.npy"), np.array([eng.calls]))
        import pickle

        with open(os.path.join(out_dir, "res.pkl"), "wb") as f:
            pickle.dump({"groups": groups, "refs": refs, "w": w, "d": d, "rw": rw, "rd": rd}, f)
    else:
        assert groups is None and refs is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("capacity", [1 << 20, 150])
def test_two_rank_search_matches_oracle(tmp_path, capacity):
    import pickle

    from oracle import vdf_oracle as orc

    mp.spawn(_worker, args=(2, _free_port(), capacity, str(tmp_path)), nprocs=2, join=True)
    res = pickle.load(open(tmp_path / "res.pkl", "rb"))
    assert res["groups"] == orc.search_self_sorted(res["w"], res["d"], 350)
    assert res["refs"] == orc.search_refs_sorted(res["w"], res["d"], res["rw"], res["rd"], 300)
    calls = int(np.load(tmp_path / "calls.npy")[0])
    assert (calls > 1) == (capacity < 1000)  # the small buffer must have gone through the overflow protocol
    # the replay filter across ranks (DistExchange): with room for every hit both ranks drop rows that cannot become targets - rows whose
    # root lives on the OTHER rank included, which only the OR of the bitmaps over the ranks can tell them
    f0, f1 = np.load(tmp_path / "filter_0.npy"), np.load(tmp_path / "filter_1.npy")
    if capacity >= 1000:
        assert f0[0] > 0 and f1[0] > 0 and f0[1] == 1 and f1[1] == 1


@pytest.mark.timeout(300)
@pytest.mark.parametrize("capacity", [1 << 20, 20])
def test_eight_rank_search_matches_oracle(tmp_path, capacity):
    """G = 8 on the CPU: the orchestration an 8-GPU node runs under torch.distributed (all-gather of eight uneven shards with an empty one,
    row tiles dealt over eight ranks, hit gather / merge on rank 0, the overflow protocol's MIN all-reduce and bitmap broadcast, the replay
    filter's agree / OR-of-bitmaps over eight ranks, references split eight ways)."""
    import pickle

    from oracle import vdf_oracle as orc

    mp.spawn(_worker, args=(8, _free_port(), capacity, str(tmp_path)), nprocs=8, join=True)
    res = pickle.load(open(tmp_path / "res.pkl", "rb"))
    assert res["groups"] == orc.search_self_sorted(res["w"], res["d"], 350)
    assert res["refs"] == orc.search_refs_sorted(res["w"], res["d"], res["rw"], res["rd"], 300)
    calls = int(np.load(tmp_path / "calls.npy")[0])
    assert (calls > 1) == (capacity < 1000)
    if capacity >= 1000:
        f = [np.load(tmp_path / f"filter_{r}.npy") for r in range(8)]
        assert all(x[1] == 1 for x in f) and sum(int(x[0]) for x in f) > 0  # every rank took part in ONE filtered launch; hits were dropped


def test_split_range_is_contiguous_and_complete():
    from vid_dup_finder_lib_amd.distributed import split_range

    for n in (0, 1, 7, 100, 1001):
        for world in (1, 2, 3, 8):
            parts = [split_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _failing_worker(rank, world, port, out_dir):
    """Rank 1's launch fails before it reaches the filter's agree (an out-of-memory on one GPU, say): nobody may hang."""
    import datetime

    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    import hashgen as hg
    from vid_dup_finder_lib_amd import distributed as vd

    rng = np.random.default_rng(78)
    words, dur = hg.planted_set(rng, 400, n_clusters=20, max_copies=6, max_flips=200, durations="windowed")
    w, d, _ = hg.sort_by_duration(words, dur)
    fw, fd = torch.from_numpy(w.view(np.int64)), torch.from_numpy(d.view(np.int32))

    class Failing(FakeEngine):
        hit_filter_enabled = True

        def search_self_device_replay(self, *a, **kw):
            if rank == 1:
                raise MemoryError("rank 1: hipMalloc failed")
            return super().search_self_device_replay(*a, **kw)

    eng = Failing(w, d)
    eng.matched_bits = None
    msg = "no exception"
    try:
        vd.search_self_sharded(eng, fw, fd, 350)
    except Exception as e:  # noqa: BLE001
        msg = f"{type(e).__name__}: {e}"
    with open(os.path.join(out_dir, f"exc_{rank}.txt"), "w") as f:
        f.write(msg)
    dist.barrier()  # both ranks got here: neither is stuck in a collective the other never entered
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_a_rank_that_fails_before_agree_releases_its_peers(tmp_path):
    """DistExchange's failure protocol (distributed.py): the failed rank still makes the launch's agree collective, flagged; its peer
    skips the filter, both meet at the status exchange and BOTH raise - the failed one its own error, the other one naming a peer."""
    mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    e0, e1 = open(tmp_path / "exc_0.txt").read(), open(tmp_path / "exc_1.txt").read()
    assert e1 == "MemoryError: rank 1: hipMalloc failed"
    assert e0.startswith("RuntimeError: search_self_sharded: the launch failed on another rank")
