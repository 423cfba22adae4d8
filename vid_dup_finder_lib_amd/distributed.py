"""Multi-GPU search: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

How the path shards (DESIGN.md "Multi-GPU"):
  * hashing: clips are independent -> every rank hashes its own clips, no communication;
  * search(): every rank holds a shard of the hash database.  ONE all-gather (the only exchange step of the
    path) replicates the database in every GPU's HBM; row tiles of the upper triangle are dealt round-robin
    (tile t -> rank t % world) so the triangular work balances; each rank emits the thresholded pairs of its
    tiles; rank 0 merges the (sparse) hit lists and replays the greedy consumption of
    Search::search_self (search_algorithm.rs:131-170) once -> MatchGroups identical for every world size;
  * search_with_references(): candidates replicated the same way, references split contiguously by rank;
    per-rank results concatenate in rank order (= reference input order).

torch is plumbing here (device buffers, streams, collectives); the arithmetic is in libvdf_hip.so.
The `engine` argument only needs search_self_device / search_refs_device, so the orchestration is testable on
CPU with the gloo backend and a stand-in hit producer (tests/test_distributed_gloo.py).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import engine as _eng
from ._capi import HASH_WORDS

UINT32_MAX = 0xFFFFFFFF


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def all_gather_database(local_words: torch.Tensor, local_dur: torch.Tensor, group=None, force: bool = False):
    """Replicate the sharded database: local_words [n_local, 16] int64, local_dur [n_local] int32 (same device).
    Returns (words [n, 16], dur [n]) in rank order.  Shards may have different sizes (padded for the collective).
    force=True runs the collective even at world size 1 (bench.py under torchrun: exercises the RCCL path)."""
    rank, world = _world(group)
    if world == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return local_words, local_dur
    dev = local_words.device
    if dev.type == "cuda" and dist.get_backend(group) != "nccl":
        # test-only route (e.g. two ranks sharing one GPU over gloo): run the collective on host copies
        w, d = all_gather_database(local_words.cpu(), local_dur.cpu(), group, force)
        return w.to(dev), d.to(dev)
    n_local = torch.tensor([local_words.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad_w = torch.zeros((m, HASH_WORDS), dtype=torch.int64, device=dev)
    pad_d = torch.zeros((m,), dtype=torch.int32, device=dev)
    pad_w[: local_words.shape[0]] = local_words
    pad_d[: local_dur.shape[0]] = local_dur
    out_w = torch.empty((world * m, HASH_WORDS), dtype=torch.int64, device=dev)
    out_d = torch.empty((world * m,), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out_w, pad_w, group=group)  # X1: (n/G) * 16 x int64 per rank
    dist.all_gather_into_tensor(out_d, pad_d, group=group)  # X2: n/G x int32 per rank
    if all(s == m for s in sizes):
        return out_w, out_d
    keep = torch.cat([torch.arange(r * m, r * m + s, device=dev) for r, s in enumerate(sizes)])
    return out_w.index_select(0, keep).contiguous(), out_d.index_select(0, keep).contiguous()


def _gather_hits(hits: np.ndarray, group=None) -> Optional[np.ndarray]:
    """Variable-length gather of [k, 2] uint32 hit lists to rank 0 (merged and sorted by (row, col)): one all-gather of the
    lengths, one all_gather_into_tensor of the lists padded to the longest - plain tensors on the collective's device (HBM
    for RCCL), nothing is pickled."""
    rank, world = _world(group)
    if world == 1:
        return hits
    cdev = _coll_device(group)
    hits = np.ascontiguousarray(hits, dtype=np.uint32).reshape(-1, 2)
    n_local = torch.tensor([hits.shape[0]], dtype=torch.int64, device=cdev)
    sizes = torch.zeros(world, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(sizes, n_local, group=group)
    sizes = [int(x) for x in sizes.tolist()]
    m = max(sizes)
    if m == 0:
        return np.zeros((0, 2), np.uint32) if rank == 0 else None
    pad = torch.zeros((m, 2), dtype=torch.int32, device=cdev)
    if hits.shape[0]:
        pad[: hits.shape[0]] = torch.from_numpy(hits.view(np.int32)).to(cdev)
    out = torch.empty((world * m, 2), dtype=torch.int32, device=cdev)
    dist.all_gather_into_tensor(out, pad, group=group)
    if rank != 0:
        return None
    allh = out.cpu().numpy().view(np.uint32).reshape(world, m, 2)
    allh = np.concatenate([allh[r, : sizes[r]] for r in range(world)])
    return _eng.sort_hits(allh)  # every rank's list is sorted already and the ranks own disjoint rows: C++ radix sort


def _stream_for(t: torch.Tensor, stream: Optional[int]) -> int:
    """The hipStream_t the library must launch on so that it is ordered behind the torch work that produced `t`
    (collectives, index_select, the bitmap upload): torch's CURRENT stream unless the caller names one.  The library's
    own stream (handle 0 / NULL) has no ordering with torch's streams, so it is never the default here."""
    if stream:
        return int(stream)
    if t.device.type == "cuda":
        return int(torch.cuda.current_stream(t.device).cuda_stream)
    return 0


def _before_engine_call(t: torch.Tensor, stream: int) -> None:
    """torch's legacy default stream has handle 0, which the C ABI reads as "the context's own stream": in that case
    (and only then) the torch-side producers are drained first."""
    if not stream and t.device.type == "cuda":
        torch.cuda.synchronize(t.device)


def _coll_device(group=None) -> torch.device:
    backend = dist.get_backend(group) if dist.is_initialized() else "gloo"
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


class DistExchange:
    """How the ranks of one sharded search() launch meet for the replay filter (include/vdf.h: vdf_shard_exchange; the library
    drops hits of rows that can never become targets, which is a property of the COMPLETE hit set): one all-gather of
    (complete, hits) per launch, and - only when the filter runs - two all-gathers of a bitmap of 1 bit per entry (125 KB
    per million entries and rank), OR-ed on the device by the library (RCCL has no bitwise-OR reduction).

    Failure protocol: the agree all-gather carries an error flag.  A rank whose launch failed BEFORE it reached agree still performs that
    one collective (search_self_sharded calls agree(failed=True) for it), so its peers are not left waiting in it; they see the flag, go on
    without the filter (no further exchange collectives in that launch) and raise together with the failed rank at the launch's status
    exchange.  What is NOT covered: a callback that fails on one rank only AFTER agree (or_bitmap raising asymmetrically) - callbacks
    must fail on every rank or on none."""

    def __init__(self, device: torch.device, group=None):
        self.dev, self.group = device, group
        self.rank, self.world = _world(group)
        self.filtered_launches = 0
        self.agree_calls = 0
        self.peer_failed = False  # some rank reported a failed launch in the last agree

    def agree(self, complete: bool, total_hits: int, failed: bool = False):
        cdev = _coll_device(self.group)
        mine = torch.tensor([1 if complete else 0, int(total_hits), 1 if failed else 0], dtype=torch.int64, device=cdev)
        allv = torch.empty(3 * self.world, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(allv, mine, group=self.group)
        self.agree_calls += 1
        v = allv.view(self.world, 3).cpu()
        self.peer_failed = bool(v[:, 2].max().item())
        if self.peer_failed:
            return False, 0  # nobody filters: the launch ends without another exchange collective
        return bool(v[:, 0].min().item()), int(v[:, 1].sum().item())

    def or_bitmap(self, engine, d_bitmap: int, n_words: int, stream: int):
        dev = self.dev
        on_gpu = dev.type == "cuda"
        # the library runs on `stream`; torch's tensors and collectives are ordered on torch's current stream
        foreign = on_gpu and int(stream) != int(torch.cuda.current_stream(dev).cuda_stream)

        def fence():
            if foreign:
                torch.cuda.synchronize(dev)

        local = torch.zeros(n_words, dtype=torch.int32, device=dev)
        fence()
        engine.bitmap_or_device(local.data_ptr(), d_bitmap, n_words, 1, stream)  # a copy: the destination is zero
        fence()
        if on_gpu and dist.get_backend(self.group) == "nccl":
            gathered = torch.empty(self.world * n_words, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(gathered, local, group=self.group)
        else:  # test route (gloo): the collective runs on host copies
            host = torch.empty(self.world * n_words, dtype=torch.int32)
            dist.all_gather_into_tensor(host, local.cpu(), group=self.group)
            gathered = host.to(dev)
        fence()
        engine.bitmap_or_device(d_bitmap, gathered.data_ptr(), n_words, self.world, stream)
        fence()  # `gathered` goes back to torch's allocator when this returns
        self.filtered_launches += 1


def search_self_sharded(engine, d_words: torch.Tensor, d_dur: torch.Tensor, tol_int: int, capacity: int = 1 << 22,
                        group=None, stream: Optional[int] = None, stats: Optional[dict] = None) -> Optional[List[List[int]]]:
    """search() over a replicated, sorted database.  Returns the groups (lists of sorted indices, reference
    order) on rank 0 and None elsewhere.  d_words/d_dur live on this rank's GPU.  stream=None: torch's current stream
    (the one the all-gather and the consumption-bitmap copies are ordered on).  The hits feed the replay only, so every
    rank drops the rows that cannot become targets before its list leaves the device (DistExchange).
    stats (optional dict) receives hits_downloaded (this rank) and filtered_launches."""
    rank, world = _world(group)
    stream = _stream_for(d_words, stream)
    replay_call = getattr(engine, "search_self_device_replay", None)
    xchg = DistExchange(d_words.device, group) if (world > 1 and replay_call is not None) else None
    downloaded = 0
    n = int(d_dur.shape[0])
    if n == 0:
        return [] if rank == 0 else None
    matched = np.zeros(n, np.uint8) if rank == 0 else None
    groups = None
    d_matched = None
    row_begin = 0
    row_end = UINT32_MAX
    while row_begin < n:
        _before_engine_call(d_words, stream)
        kw = dict(shard_index=rank, shard_count=world, row_begin=row_begin, row_end=row_end,
                  d_matched=(d_matched.data_ptr() if d_matched is not None else 0), capacity=capacity, stream=stream)
        err = None
        calls_before = xchg.agree_calls if xchg is not None else 0
        try:
            if replay_call is not None:
                hits, n_hits, overflow = replay_call(d_words.data_ptr(), d_dur.data_ptr(), n, tol_int, exchange=xchg, **kw)
            else:
                hits, n_hits, overflow = engine.search_self_device(d_words.data_ptr(), d_dur.data_ptr(), n, tol_int, **kw)
        except Exception as e:  # noqa: BLE001 - this rank still owes its peers the launch's collectives before it may leave
            if world == 1:
                raise
            err, hits, overflow = e, np.zeros((0, 2), np.uint32), UINT32_MAX
            if xchg is not None and xchg.agree_calls == calls_before and getattr(engine, "hit_filter_enabled", True):
                xchg.agree(False, 0, failed=True)  # the one collective the library would have made for this launch
        downloaded += len(hits)
        if world > 1:
            # the launch's status exchange: the first row that overflowed anywhere, and whether any rank's launch failed
            t = torch.tensor([overflow, -1 if err is not None else 0], dtype=torch.int64, device=_coll_device(group))
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            overflow, any_failed = int(t[0].item()), int(t[1].item()) != 0
            if err is not None:
                raise err
            if any_failed:
                raise RuntimeError("search_self_sharded: the launch failed on another rank (its own exception names the cause)")
        merged = _gather_hits(hits, group)
        complete_end = min(overflow, n if row_end == UINT32_MAX else min(row_end, n))
        if rank == 0:
            groups = _eng.replay_self(n, merged, matched, row_begin, complete_end, groups)
        if overflow == UINT32_MAX:
            if row_end == UINT32_MAX or row_end >= n:
                break
            row_begin, row_end = row_end, UINT32_MAX
        else:
            if complete_end == row_begin:  # not even one row fit: give that row the whole buffer
                row_end = row_begin + 1
                capacity = max(capacity, n)
            else:
                row_end = UINT32_MAX
            row_begin = complete_end
        # feed the consumption bitmap back to every rank
        words = (n + 31) // 32
        cdev = _coll_device(group)
        if rank == 0:
            bits = np.packbits(matched, bitorder="little")
            bits = np.concatenate([bits, np.zeros(words * 4 - len(bits), np.uint8)]).view(np.int32)
            bm = torch.from_numpy(bits.copy()).to(cdev)
        else:
            bm = torch.empty(words, dtype=torch.int32, device=cdev)
        if world > 1:
            dist.broadcast(bm, src=0, group=group)
        d_matched = bm.to(d_words.device)
    if stats is not None:
        stats["hits_downloaded"] = downloaded
        stats["filtered_launches"] = xchg.filtered_launches // 2 if xchg is not None else 0
    if rank != 0:
        return None
    if groups is None:
        groups = _eng.replay_self(n, np.zeros((0, 2), np.uint32), matched, 0, 0, None)
    return _eng.finish_self(groups)


def split_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, order-preserving split of n items."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def search_refs_sharded(engine, d_cand_words: torch.Tensor, d_cand_dur: torch.Tensor, d_ref_words: torch.Tensor,
                        d_ref_dur: torch.Tensor, ref_index_base: int, tol_int: int, capacity: int = 1 << 22,
                        group=None, stream: Optional[int] = None) -> Optional[List[Tuple[int, List[int]]]]:
    """search_with_references() with the candidates replicated and THIS rank's slice of the references
    (global positions ref_index_base ..).  Rank 0 returns [(ref_index, [candidate indices])] in reference order."""
    rank, world = _world(group)
    stream = _stream_for(d_cand_words, stream)
    n_ref = int(d_ref_dur.shape[0])
    n_cand = int(d_cand_dur.shape[0])
    if n_ref and n_cand:
        _before_engine_call(d_cand_words, stream)
        hits, _ = engine.search_refs_device(d_cand_words.data_ptr(), d_cand_dur.data_ptr(), n_cand,
                                            d_ref_words.data_ptr(), d_ref_dur.data_ptr(), n_ref, tol_int,
                                            ref_index_base=ref_index_base, capacity=capacity, stream=stream)
    else:
        hits = np.zeros((0, 2), np.uint32)
    merged = _gather_hits(hits, group)
    if rank != 0:
        return None
    return _eng.groups_from_ref_hits(merged)


def hash_and_search_refs(engine, cand_frames: torch.Tensor, cand_dur: torch.Tensor, ref_frames: torch.Tensor,
                         ref_dur: torch.Tensor, tol_int: int, group=None, stream: Optional[int] = None, as_lists: bool = True,
                         timings: Optional[dict] = None):
    """BASELINE configs[4] end to end: every rank hashes ITS candidate clips and ITS reference clips (uint8 device
    tensors [n, >=16, H, W]; no communication), the candidate hashes are replicated with one all-gather and put into
    Search::sort order ON THE DEVICE (vdf_sort_order_device: stable by duration; paths, if any, stay with the caller),
    references keep rank order, then search_with_references runs sharded.  Nothing but the hit list leaves HBM.
    Rank 0 returns (groups, order): order[k] = global candidate index (rank-major) at sorted position k; groups =
    [(global reference index, [positions in the sorted candidate order])] or, with as_lists=False, the CSR arrays
    (offsets, members, ref_index) of vdf_groups.  timings (optional dict) receives wall ms per phase, each closed by a device
    synchronisation."""
    import time

    rank, world = _world(group)
    dev = cand_frames.device
    stream = _stream_for(cand_frames, stream)
    t_last = [time.perf_counter()]

    def lap(name):
        if timings is not None:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now

    def fence():
        """stream handle 0 = the library's own stream, which torch's streams are not ordered with: drain the device
        whenever work changes hands between torch and the library."""
        if not stream and dev.type == "cuda":
            torch.cuda.synchronize(dev)

    cd32 = cand_dur.to(torch.int32).contiguous()
    rd32 = ref_dur.to(torch.int32).contiguous()

    def _hash(frames):
        n, nf, h, w = frames.shape
        out = torch.empty((n, HASH_WORDS), dtype=torch.int64, device=dev)
        if n:
            fence()
            engine.hash_frames_device(frames.data_ptr(), n, nf, w, h, out.data_ptr(), stream=stream)
        return out

    cw = _hash(cand_frames if cand_frames.is_contiguous() else cand_frames.contiguous())
    rw = _hash(ref_frames if ref_frames.is_contiguous() else ref_frames.contiguous())
    lap("hash_ms")
    if world > 1:
        fence()
    full_w, full_d = all_gather_database(cw, cd32, group)
    lap("all_gather_ms")
    n = int(full_d.shape[0])
    order = torch.empty(n, dtype=torch.int32, device=dev)
    sorted_w = torch.empty_like(full_w)
    sorted_d = torch.empty_like(full_d)
    if n:
        fence()
        engine.sort_order_device(full_d.data_ptr(), n, order.data_ptr(), stream=stream)
        engine.apply_order_device(full_w.data_ptr(), full_d.data_ptr(), order.data_ptr(), n, sorted_w.data_ptr(),
                                  sorted_d.data_ptr(), stream=stream)
    lap("sort_ms")
    # global index of this rank's first reference
    base = 0
    if world > 1:
        cdev = _coll_device(group)
        counts = torch.zeros(world, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(counts, torch.tensor([rw.shape[0]], dtype=torch.int64, device=cdev), group=group)
        base = int(counts[:rank].sum().item())
    n_ref = int(ref_dur.shape[0])
    if n_ref and n:
        hits, _ = engine.search_refs_device(sorted_w.data_ptr(), sorted_d.data_ptr(), n, rw.data_ptr(), rd32.data_ptr(), n_ref, tol_int, ref_index_base=base,
                                            stream=stream)
    else:
        hits = np.zeros((0, 2), np.uint32)
    lap("search_ms")
    merged = _gather_hits(hits, group)
    if rank != 0:
        lap("group_ms")
        return None, None
    groups = _eng.groups_from_ref_hits(merged) if as_lists else _eng.ref_groups_csr(merged)
    fence()
    order_h = order.cpu().numpy().view(np.uint32)
    lap("group_ms")
    return groups, order_h
