"""Multi-GPU on DISTINCT devices - the tests that activate themselves the first time the suite meets a box with two or
more GPUs (they skip cleanly on the one-GPU box; tests/test_gpu_multi_ctx.py and test_gpu_two_ranks.py run the same logic
there with a repeated device 0, where RCCL is replaced by device-to-device copies).

What only distinct devices can show:
  * the library's own RCCL path behind the C ABI (csrc/multi.cpp: dlopen'd librccl, ncclCommInitAll over the device list,
    ONE grouped ncclAllGather per array for equal shards, grouped ncclBroadcasts for uneven / empty shards) - the crate's
    one-call-one-process shape (vid_dup_finder_lib/src/video_hashing/video_dup_finder.rs:7-13,19-46);
  * torch.distributed "nccl" (= RCCL over xGMI) with one rank per GPU (vid_dup_finder_lib_amd/distributed.py);
  * both forms of bench.py --gpus N, including the named BASELINE configs[3] / configs[4] legs.
Every result is compared with the CPU oracle."""
import json
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

import hashgen as hg
from oracle import vdf_oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _n_devices():
    import torch

    return torch.cuda.device_count()  # counting devices does not initialise the GPU


needs_two = pytest.mark.skipif(_n_devices() < 2, reason="needs at least two distinct GPUs")


@pytest.fixture(scope="module")
def multi():
    import vid_dup_finder_lib_amd as vdf

    eng = vdf.Engine(devices=list(range(_n_devices())))
    yield eng
    eng.close()


def _to(dev, arr, view):
    import torch

    return torch.from_numpy(arr.view(view).copy()).to(torch.device("cuda", dev))


def _shard_sizes(n, G, kind):
    if kind == "equal":
        assert n % G == 0
        return [n // G] * G
    if kind == "uneven":
        base = [n // (2 * G)] * G
        base[-1] += n - sum(base)
        return base
    sizes = [0] * G  # "empty": one device holds nothing
    rest = n
    for k in range(1, G):
        sizes[k] = rest // (G - k) if k < G - 1 else rest
        rest -= sizes[k]
    return sizes


@needs_two
def test_replay_filter_over_distinct_devices(multi):
    """A duplicate-dense database over DISTINCT GPUs: the devices OR their has-incoming / covered bitmaps through ONE grouped
    ncclAllGather each (csrc/multi.cpp: LocalExchange -> replicate) and every device drops the rows that cannot become targets
    (search_algorithm.rs:141-161); what comes down is s - 1 pairs per cluster.  tests/test_gpu_dup_heavy.py runs the same on a
    repeated device 0, where the exchange is plain device copies."""
    import bench

    G = multi.n_devices
    w, d, n_clusters, cluster_pairs = bench.make_dup_heavy(200_000)
    want = orc.search_self_sorted(w, d, 350)
    for _ in range(2):
        assert multi.search_self_sorted(w, d, 350) == want
        st, tm = multi.last_stats(), multi.last_timing()
        per = [multi.device_timing(k)["hits_filtered"] for k in range(G)]
        assert st["n_hits"] == cluster_pairs and all(p > 0 for p in per) and sum(per) == tm["hits_filtered"]
        assert st["n_hits"] - tm["hits_filtered"] == 20_000 - n_clusters <= 0.02 * cluster_pairs


@needs_two
@pytest.mark.parametrize("kind", ["equal", "uneven", "empty"])
def test_shards_on_distinct_devices_match_the_oracle(multi, kind):
    """vdf_search_self_shards / vdf_search_refs_shards with shard k resident on GPU k: equal shards take ncclAllGather,
    uneven and empty ones the grouped ncclBroadcast form."""
    import torch

    G = multi.n_devices
    assert multi.devices == list(range(G))
    n = 1200 * G
    rng = np.random.default_rng(100 + G + len(kind))
    words, dur = hg.planted_set(rng, n, n_clusters=n // 40, max_copies=5, durations="windowed")
    w, d, _ = hg.sort_by_duration(words, dur)
    sizes = _shard_sizes(n, G, kind)
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    tw = [_to(k, w[a:b], np.int64) for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
    td = [_to(k, d[a:b], np.int32) for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
    for k in range(G):
        torch.cuda.synchronize(k)
    before = torch.cuda.current_device()
    pw = [t.data_ptr() if t.numel() else 0 for t in tw]
    pd = [t.data_ptr() if t.numel() else 0 for t in td]
    for _ in range(2):  # the second call reuses the communicators
        assert multi.search_self_shards(pw, pd, sizes, 350) == orc.search_self_sorted(w, d, 350)
    assert torch.cuda.current_device() == before  # the call leaves the caller's current device alone
    per = [multi.device_stats(k) for k in range(G)]
    assert sum(p["pairs"] for p in per) == multi.last_stats()["pairs"] and all(p["pairs"] > 0 for p in per)
    pick = rng.choice(n, size=90 * G + 1, replace=False)
    rw, rd = w[pick].copy(), d[pick].copy()
    rsz = _shard_sizes(len(rd) - (len(rd) % G if kind == "equal" else 0), G, kind)
    rw, rd = rw[: sum(rsz)], rd[: sum(rsz)]
    rcut = np.concatenate([[0], np.cumsum(rsz)])
    trw = [_to(k, rw[a:b], np.int64) for k, (a, b) in enumerate(zip(rcut[:-1], rcut[1:]))]
    trd = [_to(k, rd[a:b], np.int32) for k, (a, b) in enumerate(zip(rcut[:-1], rcut[1:]))]
    for k in range(G):
        torch.cuda.synchronize(k)
    got = multi.search_refs_shards(pw, pd, sizes, [t.data_ptr() if t.numel() else 0 for t in trw],
                                   [t.data_ptr() if t.numel() else 0 for t in trd], rsz, 300)
    assert got == orc.search_refs_sorted(w, d, rw, rd, 300)
    assert torch.cuda.current_device() == before


@needs_two
def test_host_array_calls_and_hashing_fan_out_over_distinct_devices(multi):
    import torch

    G = multi.n_devices
    rng = np.random.default_rng(7)
    words, dur = hg.planted_set(rng, 9000, n_clusters=150, max_copies=6, durations="windowed")
    w, d, _ = hg.sort_by_duration(words, dur)
    assert multi.search_self_sorted(w, d, 350) == orc.search_self_sorted(w, d, 350)
    pick = rng.choice(len(d), size=257, replace=False)
    assert multi.search_refs_sorted(w, d, w[pick], d[pick], 300) == orc.search_refs_sorted(w, d, w[pick], d[pick], 300)
    frames = rng.integers(0, 256, size=(40 * G + 3, 16, 72, 96), dtype=np.uint8)
    assert np.array_equal(multi.hash_frames(frames), orc.hash_clips(frames))
    # clips resident per device
    fr = [torch.randint(0, 256, (9 + 2 * k, 16, 64, 64), dtype=torch.uint8, device=torch.device("cuda", k)) for k in range(G)]
    outs = [torch.zeros((f.shape[0], 16), dtype=torch.int64, device=f.device) for f in fr]
    for k in range(G):
        torch.cuda.synchronize(k)
    multi.hash_frames_shards([f.data_ptr() for f in fr], [f.shape[0] for f in fr], 16, 64, 64, [o.data_ptr() for o in outs])
    for f, o in zip(fr, outs):
        assert np.array_equal(o.cpu().numpy().view(np.uint64), orc.hash_clips(f.cpu().numpy()))
    # all-identical hashes against a small hit buffer: the overflow protocol with the bitmap going back to every GPU
    n = 2500
    wi = np.tile(hg.random_hashes(np.random.default_rng(1), 1), (n, 1))
    di = np.zeros(n, np.uint32)
    multi.set_hit_capacity(1500)
    try:
        got = multi.search_self_sorted(wi, di, 0)
        assert len(got) == 1 and len(got[0]) == n and multi.last_stats()["n_launches"] > 1
    finally:
        multi.set_hit_capacity(1 << 24)


def _nccl_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import hashgen as hg2
    import vid_dup_finder_lib_amd as vdf
    from vid_dup_finder_lib_amd import distributed as vd

    eng = vdf.Engine(rank)
    rng = np.random.default_rng(999)
    words, dur = hg2.planted_set(rng, 5001, n_clusters=60, max_copies=20, max_flips=150, durations="windowed")
    words[2000:2300] = words[2000]
    dur[2000:2300] = dur[2000]
    w, d, _ = hg2.sort_by_duration(words, dur)
    lo, hi = vd.split_range(len(d), rank, world)  # uneven shards: 5001 over the ranks
    fw, fd = vd.all_gather_database(torch.from_numpy(w[lo:hi].view(np.int64)).to(dev), torch.from_numpy(d[lo:hi].view(np.int32)).to(dev))
    groups = vd.search_self_sharded(eng, fw, fd, 350, capacity=1 << 20)
    small = vd.search_self_sharded(eng, fw, fd, 350, capacity=700)  # overflow protocol: all-reduce MIN + bitmap broadcast
    pick = np.random.default_rng(5).choice(len(d), size=301, replace=False)
    rw, rd = w[pick].copy(), d[pick].copy()
    a, b = vd.split_range(len(rd), rank, world)
    refs = vd.search_refs_sharded(eng, fw, fd, torch.from_numpy(rw[a:b].view(np.int64)).to(dev),
                                  torch.from_numpy(rd[a:b].view(np.int32)).to(dev), a, 300)
    # configs[4] end to end from frames: every rank hashes its own clips
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    cf = torch.randint(0, 256, (300 + 7 * rank, 16, 64, 64), dtype=torch.uint8, device=dev, generator=g)
    rf = cf[: 40 + rank].clone()
    rf[:, :, :2, :2] ^= 1  # near-copies of this rank's first candidates
    cd = torch.full((cf.shape[0],), 100 + rank, dtype=torch.int32, device=dev)
    rdur = torch.full((rf.shape[0],), 100 + rank, dtype=torch.int32, device=dev)
    e2e, order = vd.hash_and_search_refs(eng, cf, cd, rf, rdur, 350)
    allc = [None] * world
    dist.all_gather_object(allc, (cf.cpu().numpy(), cd.cpu().numpy(), rf.cpu().numpy(), rdur.cpu().numpy()))
    if rank == 0:
        with open(os.path.join(out_dir, "res.pkl"), "wb") as f:
            pickle.dump({"groups": groups, "small": small, "refs": refs, "w": w, "d": d, "rw": rw, "rd": rd, "e2e": e2e,
                         "order": order, "clips": allc}, f)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@needs_two
def test_one_rank_per_gpu_over_rccl_matches_the_oracle(tmp_path):
    import torch.multiprocessing as mp

    world = min(_n_devices(), 8)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_nccl_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = pickle.load(open(tmp_path / "res.pkl", "rb"))
    want = orc.search_self_sorted(res["w"], res["d"], 350)
    assert res["groups"] == want and res["small"] == want
    assert res["refs"] == orc.search_refs_sorted(res["w"], res["d"], res["rw"], res["rd"], 300)
    cf = np.concatenate([c[0] for c in res["clips"]])
    cd = np.concatenate([c[1] for c in res["clips"]]).astype(np.uint32)
    rf = np.concatenate([c[2] for c in res["clips"]])
    rd = np.concatenate([c[3] for c in res["clips"]]).astype(np.uint32)
    ch, rh = orc.hash_clips(cf), orc.hash_clips(rf)
    order = np.argsort(cd, kind="stable")
    assert np.array_equal(res["order"], order)
    assert res["e2e"] == orc.search_refs_sorted(ch[order], cd[order], rh, rd, 350)


def _bench(extra, timeout=1500):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "VDF_DIST_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--hash-clips", "2000",
                          "--hash-hd-clips", "0", "--no-cpu-baseline", "--no-windowed", "--no-valu", "--no-refs", "--cache-entries", "0"] + extra,
                         capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@needs_two
def test_bench_on_distinct_devices_runs_both_forms_and_the_named_legs():
    """`python bench.py --gpus 2`: the torch.distributed ranks over real RCCL, then the C ABI's single-process form in a
    fresh child (its own RCCL inside the library), with the BASELINE configs[3] / configs[4] legs at test sizes."""
    d = _bench(["--gpus", "2", "--n-hashes", "40000", "--c4-hashes", "120000", "--c5-cands", "6000", "--c5-refs", "600"])
    assert d["n_gpus"] == 2 and d["match_groups"] > 0
    c4 = d["c4_10m_sharded"]
    assert c4["n_hashes"] == 120000 and c4["scaling"] == "strong" and c4["match_groups"] >= c4["planted_pairs"] - 1
    assert d["rccl_ranks_seen"] == 2 and c4["rccl_ranks_seen"] == 2  # torch.distributed "nccl": one RCCL rank per GPU took part
    c5 = d["c5_end_to_end"]
    assert c5["n_candidates"] == 6000 and c5["n_references"] == 600 and c5["groups"] == c5["planted_references"]
    sp = d["single_process"]
    assert sp["rccl"] == "ok", sp
    assert sp["match_groups"] == d["match_groups"] and len(sp["per_device_kernel_ms"]) == 2 and sp["value"] > 0
    assert sp["c4_10m_sharded"]["match_groups"] == c4["match_groups"]
    assert sp["rccl_ranks_seen"] == 2 and sp["c4_10m_sharded"]["rccl_ranks_seen"] == 2  # the library's own ncclCommInitAll over both devices
