#!/usr/bin/env python3
"""bench.py -- headline benchmark: all-pairs Hamming search() over random VideoHashes (BASELINE.json configs[1]:
1 M hashes, default tolerance 0.35 -> 350, one MI355X), plus the DCT-hash throughput on 64x64 frame stacks
(configs[2]) reported in the same JSON line under "hash", and the named legs below.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL.  Launched by torch.distributed.run (RANK in the environment) this process IS a
rank; launched plainly it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process
BEFORE anything touches the GPU, relays rank 0's JSON line and exits with the child's status.  Either way, once the ranks
have torn their process group down, rank 0 also runs the SAME workload through the C ABI's multi-GPU context in a fresh
child process (`--single-process`: vdf_ctx_create_multi, one host thread per device inside the library, the library's own
RCCL all-gather - the form a Rust caller of search() gets) and carries its result as "single_process" in the one line;
a failing or hanging leg is reported there, the exit status stays the ranks'.

A step = one full pass of the search hot path over the database resident in HBM: (N > 1: RCCL all-gather of the
per-rank shards,) duration windows + tile list, the distance kernel, hit download, host replay of the
greedy grouping.  value = hash pairs admitted by the reference's duration windows / wall time, whole job.
Scaling of the headline is weak: the database grows as sqrt(N) so the pairs PER GPU stay fixed (~5e11).

Named legs in the same line (sizes are flags, so tests run them small):
  c4_10m_sharded   BASELINE configs[3]: all-pairs search() over 10 M hashes, database sharded over the N ranks, one
                   all-gather, row tiles round-robin (strong scaling; at N = 1 also reported as "ten_million")
  c5_end_to_end    BASELINE configs[4]: 1 M candidate + 100 k reference clips of 16 x 64 x 64 hashed per rank, hashes
                   all-gathered, Search::sort on the device, search_with_references, groups - with a phase breakdown
  dup_heavy        a duplicate-DENSE database (10 % of 1 M hashes in clusters of 2-200): what hits, suspect queue,
                   download and replay cost when the finder finds a lot (N = 1; also sharded over two slots of one GPU)
  cache_ingest     SURVEY 8f N1 at the scale it exists for: a synthetic 10 M-entry app cache (bincode bytes) -> vdf_cache_decode_mt ->
                   vdf_search_cache_entries (paths + hashes up, Search::sort on the device - the PathBuf order included -, search, map), phase by phase (N = 1)
  windowed, valu_backend, refs_c5_shape, hash.* (N = 1), cpu_baseline (the oracle on host cores; N = 1)
The line is kept short (what each leg runs is written down in DESIGN.md section 6, not repeated in every line) and ENDS with
"summary" (at most 1.5 KB): "hash_summary" - BASELINE's metric is "pairs/s + frames/s", and whoever keeps only the tail of the line still
reads the second half - the 10 M leg, c5_end_to_end, cache_ingest, the three letterbox_64x64 times and the GPU clock / power sampled
during the headline steps.
Side legs report the median and the minimum over their steps.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# VALU issue ceiling: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz.  tools/ubench_valu.hip measures 1.76 ns per
# wave64 instruction per SIMD for every VALU opcode tried (profiles/r01_ubench_valu_issue_rates.txt), i.e. 16
# lanes/clk/SIMD -- the rate behind the 78.6 TF f64 / 157 TF packed-f32 spec figures.
VALU_PEAK_LANEOPS = 256 * 4 * 16 * 2.4e9
BYTES_PER_PAIR = 128  # one 128-B candidate streamed per comparison
LANEOPS_PER_PAIR = 64  # 32 dwords x (v_xor_b32 + v_bcnt_u32_b32)
BYTES_PER_FRAME = 4104  # 4096 B read + 8 B written per 64x64 frame
# fp4 MFMA backend: a pair is a 1024-long +-1 dot product = 1024 MACs = 2048 FLOP on the f8f6f4 matrix path.
FLOP_PER_PAIR = 2048
MFMA_FP4_PEAK_TFLOPS = 10000.0  # MI355X_MICROARCH.md: ~10 PF dense FP4/FP6 (spec).  tools/ubench_mfma.hip sustains
#                                  4.4e12 pairs/s = 9.0 PFLOP/s with operands held in registers.
LEG_STEPS_MAX = 3  # the side legs repeat at most this often, whatever --steps the driver picks for the headline


def make_hashes(n, seed, planted_every=1000):
    """random_hash-style database (bits 1000..1023 zero, durations 0) with planted near-duplicates: every
    `planted_every`-th hash gets 1-4 copies with 0..379 flipped bits (some land on/over the 350 threshold)."""
    rng = np.random.default_rng(seed)
    words = rng.integers(0, 2**64, size=(n, 16), dtype=np.uint64)
    words[:, 15] &= np.uint64((1 << 40) - 1)
    src = np.arange(0, n - 8, planted_every)
    for s in src:
        for c in range(int(rng.integers(1, 5))):
            t = s + 1 + c
            k = int(rng.integers(0, 380))
            bits = np.unpackbits(words[s].view(np.uint8), bitorder="little")
            bits[rng.choice(1024, size=k, replace=False)] ^= 1
            words[t] = np.packbits(bits, bitorder="little").view(np.uint64)
    return words


def make_dup_heavy(n, seed=20250619, frac=0.10, max_cluster=200):
    """A duplicate-dense database in Search::sort order: `frac` of the n hashes sit in clusters of 2..max_cluster
    near-copies of a centre (every bit flipped with a per-member probability <= 0.15, so all members of a cluster are
    within tolerance 350 of each other) that share the centre's duration; durations log-uniform 5 s .. 2 h.
    Returns (words [n, 16] u64, durations [n] u32, n_clusters, pairs_in_clusters)."""
    rng = np.random.default_rng(seed)
    words = rng.integers(0, 2**64, size=(n, 16), dtype=np.uint64)
    words[:, 15] &= np.uint64((1 << 40) - 1)
    dur = np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=n))).astype(np.uint32)
    n_dup = int(n * frac)
    sizes = []
    left = n_dup
    while left >= 2:
        s = int(min(rng.integers(2, max_cluster + 1), left))
        if left - s == 1:
            s += 1
        sizes.append(s)
        left -= s
    idx = rng.permutation(n)[: sum(sizes)]
    pos = 0
    pairs = 0
    for s in sizes:
        mem = idx[pos:pos + s]
        pos += s
        pairs += s * (s - 1) // 2
        centre = np.unpackbits(words[mem[0]].view(np.uint8), bitorder="little")
        flips = rng.random((s - 1, 1024), dtype=np.float32) < rng.uniform(0.0, 0.15, size=(s - 1, 1)).astype(np.float32)
        bits = centre[None, :] ^ flips.astype(np.uint8)
        words[mem[1:]] = np.packbits(bits, axis=1, bitorder="little").view(np.uint64)
        dur[mem[1:]] = dur[mem[0]]
    order = np.argsort(dur, kind="stable")
    return np.ascontiguousarray(words[order]), np.ascontiguousarray(dur[order]), len(sizes), pairs


def cpu_baseline(words, tol_int, target_seconds=12.0):
    """The oracle (C port of Search::search_self, single thread like the reference) on a prefix of the same
    database.  Sized to ~10-20 s from a short calibration run.  all_cores: the search_one loop
    (search_algorithm.rs:67-74) row-parallel over every host thread - NOT in the reference (its search is single-threaded),
    stated so that the box's core count sits next to the 1-thread figure (optional row)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import vdf_oracle as orc

    cal = 6000
    d = np.zeros(cal, np.uint32)
    t0 = time.perf_counter()
    orc.search_self_sorted(words[:cal], d, tol_int)
    dt = time.perf_counter() - t0
    rate = cal * (cal - 1) / 2 / dt
    n = int(min(len(words), max(cal, (2 * rate * target_seconds) ** 0.5)))
    d = np.zeros(n, np.uint32)
    t0 = time.perf_counter()
    orc.search_self_sorted(words[:n], d, tol_int)
    dt = time.perf_counter() - t0
    pairs = n * (n - 1) / 2
    out = {"value": pairs / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
           "sample": f"oracle search_self, single thread, first {n} of the same hashes ({pairs:.3g} pairs, {dt:.1f} s)"}
    cores, visible, quota = host_cpus()
    rows_per = 64
    n_cols = int(min(len(words), 100_000))
    cw, cd = words[:n_cols], np.zeros(n_cols, np.uint32)

    def block(a):
        rows = np.arange(a, a + rows_per) % n_cols
        orc.search_refs_sorted(cw, cd, cw[rows], cd[rows], tol_int)

    n_rows, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL; rounds of one 64-row block per thread for ~5 s
        while time.perf_counter() - t0 < 5.0:
            list(ex.map(block, range(n_rows, n_rows + cores * rows_per, rows_per)))
            n_rows += cores * rows_per
    dta = time.perf_counter() - t0
    out["all_cores"] = {"value": n_rows * n_cols / dta, "unit": "pairs/s", "cores": cores, "hardware_threads_visible": visible, "cpu_quota": quota, "in_reference": False,
                        "sample": f"oracle search_one loop, {n_rows} target rows x {n_cols} candidates over a {cores}-thread pool "
                                  f"({dta:.1f} s); the reference's search is single-threaded"}
    return out


def host_cpus():
    """(threads to use, visible hardware threads, cgroup quota in CPUs or None): a container may see every hardware thread of its host and
    still be allowed a fraction of them (cgroup v2 cpu.max / v1 cfs quota) - the GPU boxes of this pool show 256 threads under a 16-CPU
    quota, and 256 threads then only take turns on 16 cores' worth of time.  `cores` of a CPU baseline is what really ran in parallel."""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    use = visible if quota is None else max(1, min(visible, int(quota + 0.5)))
    return use, visible, quota


def cpu_baseline_hash(clips_per_thread=96):
    from concurrent.futures import ThreadPoolExecutor

    from oracle import vdf_oracle as orc

    cores, visible, quota = host_cpus()
    clips_per_thread = max(clips_per_thread, 24576 // max(cores, 1))  # (the same sample - 24 576 clips - whatever the core count)
    n_clips = cores * clips_per_thread
    rng = np.random.default_rng(20250617)
    frames = rng.integers(0, 256, size=(n_clips, 16, 64, 64), dtype=np.uint8)
    orc.hash_clips(frames[:2])  # build/load outside the timed region
    chunks = [frames[i * clips_per_thread:(i + 1) * clips_per_thread] for i in range(cores)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL: the analogue of the app's rayon par_bridge
        list(ex.map(orc.hash_clips, chunks))
    dt = time.perf_counter() - t0
    n1 = min(n_clips, 4 * clips_per_thread)  # the same work on ONE thread (SURVEY 8d asks for T = 1 and T = all)
    t1 = time.perf_counter()
    orc.hash_clips(frames[:n1])
    dt1 = time.perf_counter() - t1
    return {"value": n_clips * 16 / dt, "unit": "frames/s", "cores": cores, "kind": "port", "hardware_threads_visible": visible, "cpu_quota": quota,
            "single_thread_value": n1 * 16 / dt1,
            "sample": f"oracle from_frames over a {cores}-thread pool, {n_clips} clips of 16x64x64 "
                      f"(single thread: {n1} clips)"}


LIB_PATH = os.path.join(ROOT, "vid_dup_finder_lib_amd", "libvdf_hip.so")
TRAFFIC_PATH = os.path.join(ROOT, "profiles", "pmc_traffic.json")
_lib_sha = {}


def lib_sha256(path=None):
    """sha256 of the library file this process loads (what ties a committed rocprofv3 figure to the binary being timed)."""
    import hashlib

    path = path or LIB_PATH
    if path not in _lib_sha:
        try:
            with open(path, "rb") as f:
                _lib_sha[path] = hashlib.sha256(f.read()).hexdigest()
        except OSError:
            _lib_sha[path] = None
    return _lib_sha[path]


def read_traffic(name, scale=1.0):
    """(HBM bytes per launch, source): the committed rocprofv3 --pmc figure of kernel `name` (profiles/pmc_traffic.json, written by
    tools/summarize_profiles.py from a tools/profile_round.sh run) - but only if that profile was taken with THE library being timed:
    the file records the sha256 of the libvdf_hip.so it profiled, and a figure of another binary is reported as null with the reason."""
    try:
        with open(TRAFFIC_PATH) as f:
            t = json.load(f)
    except Exception:
        return None, "profiles/pmc_traffic.json missing"
    sha = lib_sha256()
    if not t.get("lib_sha256") or t.get("lib_sha256") != sha:
        return None, "profiles/pmc_traffic.json was measured on another libvdf_hip.so (sha256 %s, loaded %s): run tools/profile_round.sh" % (
            str(t.get("lib_sha256"))[:12], str(sha)[:12])
    v = t.get(name, {}).get("hbm_bytes_per_launch")
    if v is None:
        return None, f"no entry for {name} in profiles/pmc_traffic.json"
    return v * scale, "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this library, sha256 %s)" % sha[:12]


def med_min(xs):
    """(median, min) of a leg's per-step figures."""
    a = np.asarray(list(xs), dtype=np.float64)
    return float(np.median(a)), float(a.min())


def executed_pairs(st):
    """Pair comparisons' worth of arithmetic a launch executed: blocks that take the exact early exit stop after
    early_exit_bits of the 1024 bit positions."""
    ee = st["early_exit_bits"]
    return st["pairs_computed"] - st["pairs_early_exit"] * ((1.0 - ee / 1024.0) if ee else 0.0)


def search_roofline(backend, kernel_ms):
    """roofline (+ companions) of the dominant search kernel from the library's per-step statistics:
    kernel_ms = [(kernel_ms, n_launches, pairs, pairs_computed, n_hits, pairs_early_exit, early_exit_bits)] per step,
    of ONE device (HIP-event time recorded by the library on the kernel's own stream).  What the figures mean: DESIGN.md section 6."""
    k_ms = float(np.mean([k[0] / max(k[1], 1) for k in kernel_ms]))
    k_pairs = float(np.mean([k[2] for k in kernel_ms]))  # pairs admitted on THIS rank per launch
    k_comp = float(np.mean([k[3] for k in kernel_ms]))   # pairs the tiles evaluated (>= admitted)
    k_early = float(np.mean([k[5] for k in kernel_ms]))  # pair comparisons that took the exact early exit
    ee_bits = int(kernel_ms[-1][6])
    executed_p = k_comp - k_early * ((1.0 - ee_bits / 1024.0) if ee_bits else 0.0)
    stream_gbs = k_pairs * BYTES_PER_PAIR / (k_ms * 1e-3) / 1e9
    # SURVEY 8d operand-stream MODEL of the reference loop (128 B per pair against 8 TB/s; north_star target x >= 0.5): not a bandwidth
    hbm_model = {"bound": "hbm", "achieved": stream_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "x_of_hbm_model": stream_gbs / HBM_PEAK_GBS}
    if backend == "valu":
        kname = "hamming_tile_kernel"
        traffic, src = read_traffic(kname)
        roofline = {"bound": "hbm", "kernel": kname, "achieved": stream_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": stream_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                    "kernel_ms": k_ms, "pairs_per_launch": k_pairs}
        valu = {"achieved": executed_p * LANEOPS_PER_PAIR / (k_ms * 1e-3), "peak": VALU_PEAK_LANEOPS, "unit": "lane-ops/s"}
        valu["frac"] = valu["achieved"] / valu["peak"]
        extra = {"valu": valu}
        dtype = "u32 (xor + popcount over 32 dwords per hash)"
    else:
        kname = "hamming_mfma2_kernel"
        alg_tflops = k_comp * FLOP_PER_PAIR / (k_ms * 1e-3) / 1e12
        executed = executed_p * FLOP_PER_PAIR / (k_ms * 1e-3) / 1e12
        traffic, src = read_traffic(kname)
        # frac prices the MFMA work the kernel EXECUTED (blocks that take the exact early exit stop after ee_bits of the 1024 bit
        # positions); the algorithmic 2048 FLOP per pair are reported beside it, never as the headline fraction
        roofline = {"bound": "mfma", "kernel": kname, "achieved": executed, "peak": MFMA_FP4_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": executed / MFMA_FP4_PEAK_TFLOPS,
                    "algorithmic_achieved": alg_tflops, "algorithmic_frac": alg_tflops / MFMA_FP4_PEAK_TFLOPS,
                    "traffic": traffic, "traffic_source": src, "kernel_ms": k_ms, "pairs_per_launch": k_pairs,
                    "early_exit": {"after_bits": ee_bits, "pairs_fraction": k_early / max(k_comp, 1.0)}}
        extra = {"hbm_operand_stream_model": hbm_model}
        dtype = "fp4 e2m1 ({0,1}) x fp4 -> f32 accumulate (exact half-integers < 2^11)"
    return roofline, extra, dtype


METRIC = "hash-pairs/sec all-pairs Hamming (search(), tolerance 0.35) [+ frames/sec DCT-hash in 'summary.hash_summary']"


def headline(value, steps, warmup, ms_per_step, n_gpus, dtype, n_hashes, shard, pairs, tol_int, parallelism, roofline):
    """The keys of the driver's contract, in its order."""
    return {"metric": METRIC, "value": value, "unit": "pairs/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: random 1000-bit VideoHashes, durations 0, all-pairs search() at tolerance 350, "
                                   "near-duplicates planted every 1000th hash",
                       "n_hashes": n_hashes, "hashes_per_gpu_shard": shard, "pairs": pairs, "tolerance_int": tol_int,
                       "parallelism": parallelism},
            "roofline": roofline}


class ClockSampler:
    """Samples the GPU's shader clock and socket power from sysfs on a host thread while a timed region runs, so that a box-to-box spread
    of an unchanged kernel can be attributed (0.764 vs 0.787 of the MFMA peak with no clock figure beside it).
    sclk: the level pp_dpm_sclk marks with '*' (or hwmon's freq1_input); power: hwmon's power1_average / power1_input (microwatts).
    Reads nothing but /sys; every failure (no such file, no permission) leaves the field null - the bench never depends on it."""

    def __init__(self, device_index=0, period_s=0.004):
        import glob
        import threading

        self.period = period_s
        self.sclk, self.power = [], []
        self._stop = threading.Event()
        self._thread = None
        self.card = None
        cards = sorted(c for c in glob.glob("/sys/class/drm/card[0-9]*") if os.path.exists(os.path.join(c, "device", "pp_dpm_sclk")))
        want = None
        try:  # the card whose PCI address is the torch device's
            import torch

            pr = torch.cuda.get_device_properties(device_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        except Exception:
            pass
        for c in cards:  # (the device link's last component is the GPU's own address; the ones before it are bridges)
            if want and os.path.basename(os.path.realpath(os.path.join(c, "device"))).startswith(want):
                self.card = c
        if self.card is None and cards:
            self.card = cards[min(device_index, len(cards) - 1)]
        self.hwmon = None
        if self.card:
            hw = sorted(glob.glob(os.path.join(self.card, "device", "hwmon", "hwmon*")))
            self.hwmon = hw[0] if hw else None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read()
        except OSError:
            return None

    def sample(self):
        if not self.card:
            return
        t = self._read(os.path.join(self.card, "device", "pp_dpm_sclk"))
        mhz = None
        if t:
            for line in t.splitlines():
                if "*" in line:
                    try:
                        mhz = float(line.split(":")[1].strip().split("M")[0])
                    except (IndexError, ValueError):
                        pass
        if mhz is None and self.hwmon:
            t = self._read(os.path.join(self.hwmon, "freq1_input"))
            if t and t.strip().isdigit():
                mhz = int(t) / 1e6
        if mhz is not None:
            self.sclk.append(mhz)
        if self.hwmon:
            for name in ("power1_average", "power1_input"):
                t = self._read(os.path.join(self.hwmon, name))
                if t and t.strip().isdigit():
                    self.power.append(int(t) / 1e6)
                    break

    def __enter__(self):
        import threading

        def run():
            while not self._stop.is_set():
                self.sample()
                self._stop.wait(self.period)

        if self.card:
            self._thread = threading.Thread(target=run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread:
            self._thread.join()
        return False

    def result(self):
        med = lambda v: float(np.median(v)) if v else None  # noqa: E731
        return {"sclk_mhz_median": med(self.sclk), "power_w_median": med(self.power), "samples": len(self.sclk)}


SUMMARY_MAX_BYTES = 1536


def summary_object(out):
    """The line's LAST object, at most SUMMARY_MAX_BYTES of JSON: what a reader of the line's tail must find (the driver keeps 2 - 8 KB of a
    line that is 11 KB): the frames/s half of BASELINE's metric, the north_star's own 10 M configuration, and one figure per widened leg."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else None

    s = {}
    if "hash" in out and "roofline" in out["hash"]:
        s["hash_summary"] = hash_summary(out["hash"])
    c4 = out.get("c4_10m_sharded")
    if isinstance(c4, dict):  # BASELINE configs[3]; at one GPU the north_star's own target size
        s["ten_million"] = {"n_hashes": c4.get("n_hashes"), "n_gpus": c4.get("n_gpus"), "ms": c4.get("ms_per_step"), "pairs_per_s": c4.get("pairs_per_s"),
                            "roofline_frac": (c4.get("roofline") or {}).get("frac"), "planted_found": c4.get("match_groups"),
                            "planted": c4.get("planted_pairs")}
        if "skipped" in c4:
            s["ten_million"] = {"skipped": c4["skipped"]}
    if isinstance(out.get("c5_end_to_end"), dict):
        s["c5_end_to_end"] = pick(out["c5_end_to_end"], ("ms_per_step", "ms_min", "skipped"))
    ci = out.get("cache_ingest")
    if isinstance(ci, dict):
        s["cache_ingest"] = {"entries": ci.get("entries"), "host_ms": ci.get("host_ms"), "search_ms": (ci.get("search_cache_entries") or {}).get("search_ms")}
        if "skipped" in ci:
            s["cache_ingest"] = {"skipped": ci["skipped"]}
    lb = (out.get("hash") or {}).get("letterbox_64x64")
    if isinstance(lb, dict):
        s["letterbox_64x64_ms"] = {k: round(v["ms_per_step"], 4) for k, v in lb.items() if isinstance(v, dict) and "ms_per_step" in v}
    hq = (out.get("hash") or {}).get("host_queue_1080p")
    if isinstance(hq, dict):
        s["host_queue_1080p_link_GB_per_s"] = {k: v.get("link_GB_per_s", v.get("error")) for k, v in hq.items() if isinstance(v, dict)}
    if isinstance((out.get("roofline") or {}).get("clock"), dict):
        s["clock"] = out["roofline"]["clock"]
    return s


def host_queue_leg(seconds=1.5):
    """SURVEY 8f N2 as the app would drive it: compiled caller threads, each with one decoded 1080p clip in pageable memory, submitting to
    vdf_hash_queue in a loop (tools/bench_hash_queue.cpp, built here if missing) - run as a CHILD process (bench.py's own threads would be
    bound by the GIL, and this process never execs).  frames cross the PCIe link: reported in GB/s of that link, never part of `value`."""
    import re
    import subprocess

    root = os.path.dirname(os.path.abspath(__file__))
    exe, src, lib = os.path.join(root, "tools", "bench_hash_queue"), os.path.join(root, "tools", "bench_hash_queue.cpp"), os.path.join(root, "vid_dup_finder_lib_amd")
    out = {}
    try:
        if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, src, "-L" + lib, "-lvdf_hip", "-Wl,-rpath," + lib,
                                   "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        for name, lb in (("letterbox", 1), ("plain", 0)):
            r = subprocess.run([exe, "1920", "1080", "32", "16", "2000", str(lb), str(seconds)], capture_output=True, text=True, timeout=90)
            m = re.search(r"queue (\d+) clips/s = ([\d.]+) GB/s .*?(\d+) wrong\) \| batch call of \d+ clips: (\d+) clips/s = ([\d.]+) GB/s", r.stdout)
            if r.returncode != 0 or not m:
                out[name] = {"error": (r.stderr or r.stdout)[-200:]}
                continue
            out[name] = {"clips_per_s": int(m.group(1)), "link_GB_per_s": float(m.group(2)), "wrong": int(m.group(3)),
                         "batch_call_link_GB_per_s": float(m.group(5))}
        out.update({"w": 1920, "h": 1080, "caller_threads": 32, "max_batch": 16, "max_wait_us": 2000})
    except Exception as e:  # no compiler, no time: the leg is informational
        out["error"] = repr(e)[:200]
    return out


def hash_summary(hash_leg):
    """BASELINE's second half (frames/sec DCT-hash, configs[2]) in a form short enough to survive at the END of the line."""
    r = hash_leg["roofline"]
    out = {"metric": "frames/sec DCT-hash, BASELINE configs[2]: clips of 16 x 64 x 64 u8 -> VideoHash", "value": hash_leg["value"],
           "unit": "frames/s", "clips_per_gpu": hash_leg["clips_per_gpu"], "n_gpus": hash_leg["n_gpus"],
           "ms_per_step": hash_leg["ms_per_step"], "dtype": "u8 -> i8 MFMA fixed point (exact) -> f64 DCT",
           "roofline": {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "clock")}}
    cb = hash_leg.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
    return out


def finish_line(out):
    """Order of the one JSON line: the contract's keys first, the legs, and `summary` LAST (a reader of the line's tail gets it):
    hash_summary (BASELINE's frames/s half), the 10 M leg, and one figure per widened leg, in at most SUMMARY_MAX_BYTES."""
    out.pop("hash_summary", None)
    out.pop("summary", None)
    summ = summary_object(out)
    while len(json.dumps(summ)) > SUMMARY_MAX_BYTES and len(summ) > 1:  # never happens with the fields above; the bound is the contract
        summ.pop(list(summ)[-1])
    out["summary"] = summ
    return json.dumps(out)


def free_port():
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def clean_child_env():
    """Environment for a child process that must not inherit this process's rank identity."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE",
              "GROUP_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
              "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING",
              "TORCHELASTIC_ERROR_FILE", "VDF_DIST_BACKEND"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks as a CHILD process tree (never exec: this may run under a
    profiler that has already initialised the GPU) and relay rank 0's line.  Nothing here imports torch."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout.splitlines():
        if line.startswith("{"):
            print(line, flush=True)
    raise SystemExit(proc.returncode)


def single_process_leg(args, timeout_s=600):
    """The C ABI's own multi-GPU form in a fresh CHILD process (never exec), after the ranks released the GPUs.
    Returns the dict carried as "single_process"; failures are reported, never raised."""
    import signal

    cmd = [sys.executable, os.path.abspath(__file__), "--single-process", "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--n-hashes", str(args.n_hashes), "--tolerance", str(args.tolerance),
           "--hash-clips", "0", "--c4-hashes", str(args.c4_hashes)]
    t0 = time.perf_counter()
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=clean_child_env(),
                                start_new_session=True)
        try:
            so, se = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)  # exactly the process group started above
            proc.communicate()
            return {"rccl": f"error: no result within {timeout_s} s (killed)", "wall_s": time.perf_counter() - t0}
    except Exception as e:  # noqa: BLE001
        return {"rccl": f"error: could not start the leg: {e}"}
    lines = [l for l in so.splitlines() if l.startswith("{")]
    if proc.returncode != 0 or not lines:
        tail = (se or so or "").strip().splitlines()[-3:]
        return {"rccl": "error: rc %d: %s" % (proc.returncode, " | ".join(tail)[-600:]), "wall_s": time.perf_counter() - t0}
    d = json.loads(lines[-1])
    out = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "per_device_kernel_ms", "per_device_pairs", "match_groups",
                                 "devices", "replication", "rccl_ranks_seen")}
    out["rccl"] = "ok"
    out["roofline_frac"] = d.get("roofline", {}).get("frac")
    out["c4_10m_sharded"] = d.get("c4_10m_sharded")
    out["form"] = d["config"]["parallelism"]
    out["wall_s"] = time.perf_counter() - t0
    return out


C4_N1_REFERENCE_MS = 10375.4  # ten_million at one GPU, driver run of round 4 (BENCH_r04.json): what N ranks are compared with


def c4_summary(n10, n_gpus, ms):
    """BASELINE configs[3] (all-pairs search() over 10 M hashes sharded over the GPUs; strong scaling): the figures to read first on an
    8-GPU run.  speedup_vs_n1_model = (one GPU's measured time / N) / this run's time: 1.0 = perfect strong scaling."""
    p10 = n10 * (n10 - 1) // 2
    out = {"workload": "BASELINE configs[3]", "n_hashes": n10, "pairs": p10, "scaling": "strong", "n_gpus": n_gpus, "steps": 1,
           "ms_per_step": ms, "pairs_per_s": p10 / (ms * 1e-3), "hbm_operand_stream_x": p10 / (ms * 1e-3) * BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBS}
    if n10 == 10_000_000:
        out["n1_reference_ms"] = C4_N1_REFERENCE_MS
        out["speedup_vs_n1_model"] = C4_N1_REFERENCE_MS / n_gpus / ms
    return out


def gen_device_hashes(torch, dev, n, seed, plant_every=100_000, plant_flips=341):
    """n random VideoHashes generated ON the device (bits 1000..1023 zero); every plant_every-th gets a copy with 0..340
    flipped bits right behind it.  Returns (int64 tensor [n, 16], number of planted pairs)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lo32 = torch.randint(0, 2**32, (n, 16), dtype=torch.int64, device=dev, generator=g)
    hi32 = torch.randint(0, 2**32, (n, 16), dtype=torch.int64, device=dev, generator=g)
    w = lo32 | (hi32 << 32)
    del lo32, hi32
    w[:, 15] &= (1 << 40) - 1
    src = torch.arange(0, n - 1, plant_every, device=dev)
    if len(src):
        rng = np.random.default_rng(seed)
        rows = w[src].cpu().numpy().view(np.uint64)
        for r in range(len(rows)):
            bits = np.unpackbits(rows[r].view(np.uint8), bitorder="little")
            bits[rng.choice(1024, size=int(rng.integers(0, plant_flips)), replace=False)] ^= 1
            rows[r] = np.packbits(bits, bitorder="little").view(np.uint64)
        w[src + 1] = torch.from_numpy(rows.view(np.int64)).to(dev)
    return w, int(len(src))


def run_single_process(args):
    """--single-process: all N GPUs from ONE process through the C ABI's multi-GPU context (vdf_ctx_create_multi: one
    host thread + stream per device inside the library; database shards resident per GPU, replicated by the library's
    own RCCL all-gather).  Same step, same metric, same JSON line as the multi-process form; torch only allocates the
    buffers.  With fewer physical GPUs than --gpus the device list wraps (testing on one GPU)."""
    import gc

    import torch

    import vid_dup_finder_lib_amd as vdf
    from vid_dup_finder_lib_amd import distributed as vd
    from vid_dup_finder_lib_amd import engine as ve

    G = args.gpus
    n_phys = max(torch.cuda.device_count(), 1)
    devices = [k % n_phys for k in range(G)]
    eng = vdf.Engine(devices=devices)
    tol_int = ve.tolerance_int(args.tolerance)
    n_total = int(round(args.n_hashes * G ** 0.5))
    words = make_hashes(n_total, 20250613)
    pairs = n_total * (n_total - 1) // 2
    sw, sd, sizes = [], [], []
    for k in range(G):
        lo, hi = vd.split_range(n_total, k, G)
        dev = torch.device("cuda", devices[k])
        sw.append(torch.from_numpy(words[lo:hi].view(np.int64)).to(dev))
        sd.append(torch.zeros(hi - lo, dtype=torch.int32, device=dev))
        sizes.append(hi - lo)
    pw, pd = [t.data_ptr() for t in sw], [t.data_ptr() for t in sd]

    def sync_all():
        for k in set(devices):
            torch.cuda.synchronize(k)

    per_dev, n_groups = [], None

    def step():
        nonlocal n_groups
        groups = eng.search_self_shards(pw, pd, sizes, tol_int)
        n_groups = len(groups)
        per_dev.append([eng.device_stats(k) for k in range(G)])

    for _ in range(args.warmup):
        step()
    per_dev.clear()
    gc.collect()
    gc.disable()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    gc.enable()
    backend = os.environ.get("VDF_SEARCH_BACKEND", "mfma")
    # the roofline describes ONE device's kernel: the slowest slot of each step
    slow = [max(st, key=lambda q: q["kernel_ms"]) for st in per_dev]
    kernel_ms = [(q["kernel_ms"], q["n_launches"], q["pairs"], q["pairs_computed"], q["n_hits"], q["pairs_early_exit"],
                  q["early_exit_bits"]) for q in slow]
    roofline, extra, dtype = search_roofline(backend, kernel_ms)
    distinct = len(set(devices)) == len(devices)
    out = headline(pairs * args.steps / dt, args.steps, args.warmup, dt / args.steps * 1e3, G, dtype, n_total, sizes[0], pairs, tol_int,
                   f"ONE process, vdf_ctx_create_multi over devices {devices}: shards resident per GPU, RCCL all-gather inside the "
                   "library, row tiles round-robin, one host replay", roofline)
    out.update({"match_groups": n_groups, "search_backend": backend, "devices": devices,
                "replication": ("ncclAllGather over xGMI (librccl loaded by the library)" if distinct and G > 1
                                else "device-to-device copies (repeated or single device: RCCL takes one rank per device)"),
                "rccl_ranks_seen": eng.rccl_ranks(),
                "per_device_kernel_ms": [float(np.mean([st[k]["kernel_ms"] for st in per_dev])) for k in range(G)],
                "per_device_pairs": [int(per_dev[-1][k]["pairs"]) for k in range(G)]})
    out.update(extra)
    del sw, sd
    # ---- BASELINE configs[3] in this form: 10 M hashes, shard k generated on GPU k, one vdf_search_self_shards call
    if args.c4_hashes > 0:
        n10 = args.c4_hashes
        sh, sdur, ssz, planted = [], [], [], 0
        for k in range(G):
            lo, hi = vd.split_range(n10, k, G)
            dev = torch.device("cuda", devices[k])
            w, npl = gen_device_hashes(torch, dev, hi - lo, 20250614 + k)
            sh.append(w)
            sdur.append(torch.zeros(hi - lo, dtype=torch.int32, device=dev))
            ssz.append(hi - lo)
            planted += npl
        sync_all()
        tenth = [max(s // 10, 1) for s in ssz]
        eng.search_self_shards([t.data_ptr() for t in sh], [t.data_ptr() for t in sdur], tenth, tol_int)  # allocations
        sync_all()
        t1 = time.perf_counter()
        g10 = eng.search_self_shards([t.data_ptr() for t in sh], [t.data_ptr() for t in sdur], ssz, tol_int)
        sync_all()
        dt10 = time.perf_counter() - t1
        p10 = n10 * (n10 - 1) // 2
        out["c4_10m_sharded"] = dict(c4_summary(n10, G, dt10 * 1e3), match_groups=len(g10), planted_pairs=planted,
                                     per_device_kernel_ms=[eng.device_stats(k)["kernel_ms"] for k in range(G)],
                                     rccl_ranks_seen=eng.rccl_ranks(), timing=eng.last_timing())
        del sh, sdur
    if args.hash_clips > 0:
        nc = args.hash_clips
        frames, outs = [], []
        for k in range(G):
            dev = torch.device("cuda", devices[k])
            g = torch.Generator(device=dev)
            g.manual_seed(20250617 + k)
            frames.append(torch.randint(0, 256, (nc, 16, 64, 64), dtype=torch.uint8, device=dev, generator=g))
            outs.append(torch.zeros((nc, 16), dtype=torch.int64, device=dev))
        fp, op, ns = [t.data_ptr() for t in frames], [t.data_ptr() for t in outs], [nc] * G
        sync_all()
        for _ in range(max(args.warmup, 1)):
            eng.hash_frames_shards(fp, ns, 16, 64, 64, op)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eng.hash_frames_shards(fp, ns, 16, 64, 64, op)  # returns when every device has finished
        ms = (time.perf_counter() - t1) / args.steps * 1e3
        h_gbs = nc * 16 / (ms * 1e-3) * BYTES_PER_FRAME / 1e9
        out["hash"] = {"value": G * nc * 16 / (ms * 1e-3), "unit": "frames/s", "clips_per_gpu": nc, "n_gpus": G, "ms_per_step": ms,
                       "roofline": {"bound": "hbm", "kernel": "resize_dct_hash_persistent_kernel", "achieved": h_gbs,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": h_gbs / HBM_PEAK_GBS, "traffic": None,
                                    "timed": "host wall per call incl. launch + join of the per-device threads"}}
    print(finish_line(out))
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-hashes", type=int, default=1_000_000, help="database size at 1 GPU (grows as sqrt(gpus))")
    ap.add_argument("--hash-clips", type=int, default=100_000, help="clips for the DCT-hash leg (0 = skip)")
    ap.add_argument("--hash-hd-clips", type=int, default=1000, help="1080p clips for the large-frame hash leg (0 = skip)")
    ap.add_argument("--tolerance", type=float, default=0.35)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-windowed", dest="windowed", action="store_false", help="skip the windowed-durations leg")
    ap.add_argument("--c4-hashes", "--ten-million", dest="c4_hashes", type=int, default=10_000_000,
                    help="BASELINE configs[3] leg: all-pairs over this many hashes sharded over the ranks (the north_star target "
                         "at --gpus 1; 0 = skip)")
    ap.add_argument("--c5-cands", type=int, default=1_000_000, help="BASELINE configs[4] leg: candidate clips, all ranks together (0 = skip)")
    ap.add_argument("--c5-refs", type=int, default=100_000, help="BASELINE configs[4] leg: reference clips")
    ap.add_argument("--dup-heavy", type=int, default=1_000_000, help="size of the duplicate-dense leg (0 = skip; --gpus 1 only)")
    ap.add_argument("--cache-entries", type=int, default=10_000_000,
                    help="entries of the synthetic app cache of the cache_ingest leg (0 = skip; --gpus 1 only)")
    ap.add_argument("--no-valu", dest="valu_leg", action="store_false", help="skip the XOR+popcount backend leg")
    ap.add_argument("--no-host-queue", dest="host_queue", action="store_false", help="skip the batching-queue leg (a child process with compiled callers)")
    ap.add_argument("--no-refs", dest="refs_leg", action="store_false", help="skip the search_with_references leg")
    ap.add_argument("--no-single-process-leg", dest="sp_leg", action="store_false",
                    help="N > 1: do not run the C ABI's single-process form after the ranks")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 inside ONE process through vdf_ctx_create_multi (no torch.distributed)")
    args = ap.parse_args()
    if args.single_process:
        return run_single_process(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist

    import vid_dup_finder_lib_amd as vdf
    from vid_dup_finder_lib_amd import distributed as vd
    from vid_dup_finder_lib_amd import engine as ve

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also at world size 1)
    # VDF_DIST_BACKEND=gloo (testing only): ranks may outnumber GPUs; collectives run on host copies
    dist_backend = os.environ.get("VDF_DIST_BACKEND", "nccl")
    if dist_backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(dist_backend)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", local_rank)
    eng = vdf.Engine(local_rank)
    tol_int = ve.tolerance_int(args.tolerance)
    leg_steps = max(1, min(args.steps, LEG_STEPS_MAX))
    cdev = dev if dist_backend == "nccl" else torch.device("cpu")

    # ---- database: n grows as sqrt(world) so that pairs per GPU stay fixed (weak scaling) ----------------
    n_total = int(round(args.n_hashes * world ** 0.5))
    words = make_hashes(n_total, 20250613)
    lo, hi = vd.split_range(n_total, rank, world)
    shard_w = torch.from_numpy(words[lo:hi].view(np.int64)).to(dev)
    shard_d = torch.zeros(hi - lo, dtype=torch.int32, device=dev)
    pairs = n_total * (n_total - 1) // 2  # all durations equal: one window, the full triangle
    # a real (non-null) stream: the library launches on the stream it is handed, and HIP events must be recorded
    # on that same stream to see its kernels
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    kernel_ms = []
    n_groups = None
    launches, suspects = [], []

    def step():
        nonlocal n_groups
        full_w, full_d = vd.all_gather_database(shard_w, shard_d, force=use_dist)
        groups = vd.search_self_sharded(eng, full_w, full_d, tol_int, stream=stream)
        st = eng.last_stats()
        kernel_ms.append((st["kernel_ms"], st["n_launches"], st["pairs"], st["pairs_computed"], st["n_hits"],
                          st["pairs_early_exit"], st["early_exit_bits"]))
        launches.append(st["n_launches"])
        suspects.append(eng.last_timing()["suspects"])
        if rank == 0:
            n_groups = len(groups)

    import gc

    for _ in range(args.warmup):
        step()
    kernel_ms.clear()
    launches.clear()
    suspects.clear()
    gc.collect()
    gc.disable()  # a generation-2 collection (tens of ms after importing torch) must not land inside a 150 ms step
    barrier()
    clk = ClockSampler(local_rank)
    with clk:  # sysfs reads on a host thread: the GPU's clock and power WHILE the timed steps run (rank 0's figures are reported)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
    gc.enable()
    dt = max_over_ranks(dt)

    backend = os.environ.get("VDF_SEARCH_BACKEND", "mfma")
    roofline, extra, dtype = search_roofline(backend, kernel_ms)
    roofline["clock"] = clk.result()
    out = headline(pairs * args.steps / dt, args.steps, args.warmup, dt / args.steps * 1e3, world, dtype, n_total, hi - lo, pairs, tol_int,
                   f"row tiles round-robin over {world} GPU(s), one RCCL all-gather" if world > 1 else "single GPU", roofline)
    out.update({"match_groups": n_groups, "search_backend": backend, "n_launches": int(round(float(np.mean(launches)))),
                "suspects": int(round(float(np.mean(suspects))))})
    if world > 1:  # ranks that took part in the collectives of the timed steps
        out["rccl_ranks_seen"] = dist.get_world_size() if dist_backend == "nccl" else 0
        out["dist_backend"] = dist_backend
    out.update(extra)
    legs = {}  # named legs, inserted behind the headline in the order chosen at the end

    # ---- windowed-durations variant: same hashes, durations = floor(exp(U(ln 5, ln 7200))) sorted, so the
    # one-sided x1.1 window (search_algorithm.rs:99) admits ~1 % of the triangle; shows what the window/tile culling costs.  rank 0 only.
    if rank == 0 and args.windowed:
        rng = np.random.default_rng(20250613)
        dur = np.sort(np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=args.n_hashes))).astype(np.uint32))
        wd = torch.from_numpy(dur.view(np.int32)).to(dev)
        ww = torch.from_numpy(words[: args.n_hashes].view(np.int64)).to(dev)
        torch.cuda.synchronize()
        for _ in range(max(min(args.warmup, 2), 1)):
            eng.search_self_device(ww.data_ptr(), wd.data_ptr(), args.n_hashes, tol_int, stream=stream)
        w_k, w_ms = [], []
        for _ in range(leg_steps):
            t1 = time.perf_counter()
            hits_w, _, _ = eng.search_self_device(ww.data_ptr(), wd.data_ptr(), args.n_hashes, tol_int, stream=stream)
            w_ms.append((time.perf_counter() - t1) * 1e3)
            w_k.append(eng.last_stats()["kernel_ms"])
        st = eng.last_stats()
        ms_med, ms_min = med_min(w_ms)
        legs["windowed"] = {"pairs": st["pairs"], "pairs_computed": st["pairs_computed"],
                            "waste_ratio": st["pairs_computed"] / max(st["pairs"], 1), "kernel_ms": med_min(w_k)[0],
                            "ms": ms_med, "ms_min": ms_min, "steps": leg_steps, "pairs_per_s": st["pairs"] / (ms_med * 1e-3), "hits": len(hits_w)}
        del ww, wd

    # ---- the north_star's literal formulation next to the default one: XOR + popcount on the VALU (hamming_tile_kernel),
    # same database, same step.  1 GPU only, rank 0.
    if rank == 0 and world == 1 and args.valu_leg and backend != "valu":
        old_env = os.environ.get("VDF_SEARCH_BACKEND")
        os.environ["VDF_SEARCH_BACKEND"] = "valu"
        try:
            eng_v = vdf.Engine(local_rank)
        finally:
            if old_env is None:
                os.environ.pop("VDF_SEARCH_BACKEND", None)
            else:
                os.environ["VDF_SEARCH_BACKEND"] = old_env
        full_w, full_d = shard_w, shard_d
        vd.search_self_sharded(eng_v, full_w, full_d, tol_int, stream=stream)  # warm-up
        torch.cuda.synchronize()
        tv = time.perf_counter()
        gv = vd.search_self_sharded(eng_v, full_w, full_d, tol_int, stream=stream)
        torch.cuda.synchronize()
        dtv = time.perf_counter() - tv
        sv = eng_v.last_stats()
        # lane-ops the kernel executed for rows < n_rows: the last row tile is padded to whole tiles (its lanes run, on
        # zeros - not counted), waves that take the early exit stop after early_exit_bits
        pad_rows = (-n_total) % 512
        lane_ops = executed_pairs(sv) * (1.0 - pad_rows / (n_total + pad_rows)) * LANEOPS_PER_PAIR
        legs["valu_backend"] = {"kernel": "hamming_tile_kernel", "pairs_per_s": pairs / dtv, "ms_per_step": dtv * 1e3,
                                "kernel_ms": sv["kernel_ms"], "match_groups": len(gv), "steps": 1,
                                "valu": {"achieved": lane_ops / (sv["kernel_ms"] * 1e-3), "peak": VALU_PEAK_LANEOPS,
                                         "unit": "lane-ops/s", "frac": lane_ops / (sv["kernel_ms"] * 1e-3) / VALU_PEAK_LANEOPS}}
        assert len(gv) == n_groups, "VALU and MFMA backends disagree"
        eng_v.close()
    del shard_w, shard_d

    # ---- BASELINE configs[3]: all-pairs search() over 10 M VideoHashes, the database sharded over the ranks (shard r
    # generated on GPU r), ONE all-gather, row tiles dealt round-robin, one replay on rank 0.  STRONG scaling: the work is
    # fixed, expect ~10.4 s / N + the all-gather.  At one GPU this is the north_star's own target ("ten_million").
    if args.c4_hashes > 0:
        n10 = args.c4_hashes
        lo10, hi10 = vd.split_range(n10, rank, world)
        w10, planted = gen_device_hashes(torch, dev, hi10 - lo10, 20250614 + rank)
        d10 = torch.zeros(hi10 - lo10, dtype=torch.int32, device=dev)
        planted = int(round(sum_over_ranks(planted)))
        tenth = max((hi10 - lo10) // 10, 1)
        fw, fd = vd.all_gather_database(w10[:tenth].contiguous(), d10[:tenth].contiguous(), force=use_dist)
        vd.search_self_sharded(eng, fw, fd, tol_int, stream=stream)  # allocations
        del fw, fd
        barrier()
        t10 = time.perf_counter()
        fw, fd = vd.all_gather_database(w10, d10, force=use_dist)
        torch.cuda.synchronize()
        t_gather = time.perf_counter() - t10
        g10r = vd.search_self_sharded(eng, fw, fd, tol_int, stream=stream)
        barrier()
        dt10 = max_over_ranks(time.perf_counter() - t10)
        s10 = eng.last_stats()
        k10 = max_over_ranks(s10["kernel_ms"]) * 1e-3
        ex_all = sum_over_ranks(executed_pairs(s10))
        comp_all = sum_over_ranks(s10["pairs_computed"])
        if rank == 0:
            c4 = dict(c4_summary(n10, world, dt10 * 1e3), all_gather_ms=t_gather * 1e3, kernel_ms=k10 * 1e3, n_launches=s10["n_launches"],
                      match_groups=len(g10r), planted_pairs=planted)
            if world > 1:
                c4["rccl_ranks_seen"] = out.get("rccl_ranks_seen")
            if backend != "valu":  # per GPU: executed MFMA FLOP of all ranks / N / the slowest rank's kernel time
                c4["roofline"] = {"bound": "mfma", "kernel": roofline["kernel"], "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "achieved": ex_all * FLOP_PER_PAIR / k10 / 1e12 / world,
                                  "frac": ex_all * FLOP_PER_PAIR / k10 / 1e12 / world / MFMA_FP4_PEAK_TFLOPS,
                                  "algorithmic_frac": comp_all * FLOP_PER_PAIR / k10 / 1e12 / world / MFMA_FP4_PEAK_TFLOPS,
                                  "traffic": None}
            legs["c4_10m_sharded"] = c4
            if world == 1:  # the north_star's own target, under the name it has had since round 1
                legs["ten_million"] = {"same_as": "c4_10m_sharded", "ms_per_step": c4["ms_per_step"], "pairs_per_s": c4["pairs_per_s"],
                                       "match_groups": c4["match_groups"], "planted_pairs": planted,
                                       "roofline_frac": c4.get("roofline", {}).get("frac")}
        del w10, d10, fw, fd

    # ---- search_with_references at the BASELINE configs[4] shape (hash-less half): 1 M candidates x 100 k references,
    # log-uniform durations, +-5 % windows; half of the references are near-copies of candidates.  1 GPU, rank 0.
    if rank == 0 and world == 1 and args.refs_leg:
        n_c, n_r = 1_000_000, 100_000
        rr = np.random.default_rng(20250615)
        cw = make_hashes(n_c, 20250615, planted_every=10**9)
        cdur = np.sort(np.floor(np.exp(rr.uniform(np.log(5), np.log(7200), size=n_c))).astype(np.uint32))
        src_i = rr.choice(n_c, size=n_r // 2, replace=False)
        rw = np.concatenate([cw[src_i], make_hashes(n_r - n_r // 2, 20250616, planted_every=10**9)])
        rdur = np.concatenate([cdur[src_i], np.floor(np.exp(rr.uniform(np.log(5), np.log(7200), size=n_r - n_r // 2))).astype(np.uint32)])
        pr = rr.permutation(n_r)
        rw, rdur = rw[pr], rdur[pr]
        tt = [torch.from_numpy(a).to(dev) for a in (cw.view(np.int64), cdur.view(np.int32), rw.view(np.int64), rdur.view(np.int32))]
        torch.cuda.synchronize()

        def refs_runs(reps):
            kms, wall, tms = [], [], []
            for i in range(reps + 2):
                t1 = time.perf_counter()
                hr, nh = eng.search_refs_device(tt[0].data_ptr(), tt[1].data_ptr(), n_c, tt[2].data_ptr(), tt[3].data_ptr(), n_r, tol_int,
                                                stream=stream)
                if i >= 2:
                    wall.append((time.perf_counter() - t1) * 1e3)
                    kms.append(eng.last_stats()["kernel_ms"])
                    tms.append(eng.last_timing())
            return med_min(kms)[0], med_min(wall), {k: float(np.median([t[k] for t in tms])) for k in ("prep_ms", "stream_ms", "resolve_ms", "download_ms")}, int(nh)

        k_un, w_un, t_un, nh = refs_runs(leg_steps)
        eng.pin_database(tt[0].data_ptr(), n_c)  # the app's situation: ONE cache database, searched with reference set after reference set
        k_pin, w_pin, t_pin, nh = refs_runs(leg_steps)
        eng.pin_database(0, 0)
        sr = eng.last_stats()
        legs["refs_c5_shape"] = {"pairs": sr["pairs"], "pairs_computed": sr["pairs_computed"],
                                 "waste_ratio": sr["pairs_computed"] / max(sr["pairs"], 1), "workgroups": sr["n_tiles"], "kernel_ms": k_pin,
                                 "ms": w_pin[0], "ms_min": w_pin[1], "hits": nh, "pairs_per_s": sr["pairs"] / (w_pin[0] * 1e-3), "timing": t_pin,
                                 "unpinned": {"ms": w_un[0], "ms_min": w_un[1], "kernel_ms": k_un, "timing": t_un}}
        del tt

    # ---- BASELINE configs[4] END TO END: candidate and reference clips (16 x 64 x 64 u8) resident per rank -> hashes ->
    # all-gather -> Search::sort on the device -> search_with_references -> groups.  Half of the references are copies of
    # this rank's candidate clips (same frames, same duration): each must find exactly its source.
    if args.c5_cands > 0 and args.c5_refs > 0:
        out_c5 = leg_c5(args, torch, dist, vd, eng, dev, rank, world, use_dist, tol_int, stream, barrier, max_over_ranks,
                        sum_over_ranks, leg_steps)
        if rank == 0:
            legs["c5_end_to_end"] = out_c5

    # ---- a duplicate-DENSE database: the product's own case.  Whole call through the C ABI on device-resident shards
    # (vdf_search_self_shards on a one-device context: replication = one device copy), with the library's phase timing.
    if rank == 0 and world == 1 and args.dup_heavy > 0:
        legs["dup_heavy"] = leg_dup_heavy(args, torch, vdf, dev, local_rank, tol_int, words, leg_steps)

    # ---- SURVEY 8f N1 at its own scale: app cache bytes -> decode -> vdf_search_cache_entries, phase by phase.  Host + 1 GPU, rank 0.
    if rank == 0 and world == 1 and args.cache_entries > 0:
        legs["cache_ingest"] = leg_cache_ingest(args, vdf, eng, tol_int)

    # ---- DCT-hash leg (configs[2]): frame stacks resident in HBM; clips are independent, so every rank hashes its
    # own args.hash_clips clips with no communication (weak scaling) and the job rate is the sum -----------------
    hash_leg = None
    if args.hash_clips > 0:
        nc = args.hash_clips
        g = torch.Generator(device=dev)
        g.manual_seed(20250617 + rank)
        frames = torch.randint(0, 256, (nc, 16, 64, 64), dtype=torch.uint8, device=dev, generator=g)
        out_h = torch.zeros((nc, 16), dtype=torch.int64, device=dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(max(args.warmup, 1)):
            eng.hash_frames_device(frames.data_ptr(), nc, 16, 64, 64, out_h.data_ptr(), stream=stream)
        barrier()
        # (the leg is milliseconds long: repeat it untimed under the sampler until ~0.25 s of this kernel have run, then time the steps)
        hclk = ClockSampler(local_rank, period_s=0.002)
        with hclk:
            t_clk = time.perf_counter()
            while time.perf_counter() - t_clk < 0.25:
                for _ in range(20):
                    eng.hash_frames_device(frames.data_ptr(), nc, 16, 64, 64, out_h.data_ptr(), stream=stream)
                torch.cuda.synchronize()
            ev0.record()
            for _ in range(args.steps):
                eng.hash_frames_device(frames.data_ptr(), nc, 16, 64, 64, out_h.data_ptr(), stream=stream)
            ev1.record()
            torch.cuda.synchronize()
        ms = max_over_ranks(ev0.elapsed_time(ev1) / args.steps)
        fps = world * nc * 16 / (ms * 1e-3)
        h_gbs = nc * 16 / (ms * 1e-3) * BYTES_PER_FRAME / 1e9  # per GPU: the kernel's own roofline
        h_traffic, h_src = read_traffic("resize_dct_hash_persistent_kernel", nc / 100_000)  # profiled at 100 k clips per launch
        hash_leg = {"value": fps, "unit": "frames/s", "clips_per_gpu": nc, "n_gpus": world, "ms_per_step": ms,
                    "roofline": {"bound": "hbm", "kernel": "resize_dct_hash_persistent_kernel", "achieved": h_gbs,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": h_gbs / HBM_PEAK_GBS,
                                 "traffic": h_traffic, "traffic_source": h_src, "clock": hclk.result()}}
        del frames, out_h
        # the same path at the size decoders really hand over (informational; the headline stays the 64 x 64 config)
        if args.hash_hd_clips > 0 and world == 1:
            def timed_steps(fn):
                """per-step HIP-event times (ms) of leg_steps calls on the bench's stream"""
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(leg_steps + 1)]
                barrier()
                evs[0].record()
                res = None
                for i in range(leg_steps):
                    res = fn()
                    evs[i + 1].record()
                torch.cuda.synchronize()
                return [evs[i].elapsed_time(evs[i + 1]) for i in range(leg_steps)], res

            def big_leg(name, n, w, h, kernel, key, profiled):
                buf = torch.empty((n, 16, h, w), dtype=torch.uint8, device=dev)
                chunk = max(1, (1 << 31) // (16 * h * w))
                for c0 in range(0, n, chunk):
                    buf[c0:c0 + chunk] = torch.randint(0, 256, (min(chunk, n - c0), 16, h, w), dtype=torch.uint8, device=dev, generator=g)
                oh = torch.zeros((n, 16), dtype=torch.int64, device=dev)
                eng.hash_frames_device(buf.data_ptr(), n, 16, w, h, oh.data_ptr(), stream=stream)
                ts, _ = timed_steps(lambda: eng.hash_frames_device(buf.data_ptr(), n, 16, w, h, oh.data_ptr(), stream=stream))
                ms_l, ms_lo = med_min(ts)
                gbs = n * 16 * (w * h + 8) / (ms_l * 1e-3) / 1e9
                hash_leg[name] = {"clips": n, "w": w, "h": h, "ms_per_step": ms_l, "ms_min": ms_lo,
                                  "frames_per_s_per_gpu": n * 16 / (ms_l * 1e-3),
                                  "roofline": {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS,
                                               "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                               "traffic": read_traffic(key, n / profiled)[0]}}
                del buf, oh

            big_leg("full_hd", args.hash_hd_clips, 1920, 1080, "resize_mfma_frame_wavestream_kernel",
                    "resize_mfma_frame_wavestream_kernel@1920x1080", 1000)
            # a pitch that is not a multiple of the 128-byte line: the linear-stream kernel (LDS-DMA of whole chunks)
            big_leg("pitch_480x270", 4000, 480, 270, "resize_mfma_frame_stream_kernel", "resize_mfma_frame_stream_kernel@480x270", 4000)
            # 4K: the K-split form of the stream kernel (horizontal table in registers)
            big_leg("uhd_3840x2160", 250, 3840, 2160, "resize_mfma_frame_ksplit_kernel", "resize_mfma_frame_ksplit_kernel@3840x2160", 250)

            # SURVEY 8f N3, the builder's default (Cropdetect::Letterbox before from_frames) at the decoder's native size: detect + crop +
            # hash of 1080p clips with bars, one vdf_hash_frames_u8_letterbox_device call per step (it downloads the boxes in between).
            # box_GB_per_s counts the bytes of the crop boxes (what the resize has to read); the probe reads of the detect pass are on top.
            def letterbox_leg(n, w, h, key="letterbox_full_hd", names=("no_bars", "top_bottom_bars", "side_bars", "one_black_probe_frame_in_1000")):
                res = {"clips": n, "w": w, "h": h}
                base = torch.empty((n, 16, h, w), dtype=torch.uint8, device=dev)
                chunk = max(1, (1 << 31) // (16 * h * w))
                for c0 in range(0, n, chunk):
                    base[c0:c0 + chunk] = torch.randint(0, 256, (min(chunk, n - c0), 16, h, w), dtype=torch.uint8, device=dev, generator=g)
                oh = torch.zeros((n, 16), dtype=torch.int64, device=dev)
                dcr = torch.zeros((n, 4), dtype=torch.int32, device=dev)  # the boxes stay on the device (vdf_hash_frames_u8_letterbox_device_async)
                bar_t, bar_s = int(h * 0.12), int(w * 0.125)
                for name in names:
                    fr = base
                    if name != "no_bars":
                        fr = base.clone()
                    if name == "top_bottom_bars":  # 2.39 : 1 in 16 : 9
                        fr[:, :, :bar_t, :] = 16
                        fr[:, :, h - bar_t:, :] = 16
                    elif name == "side_bars":  # 4 : 3 in 16 : 9
                        fr[:, :, :, :bar_s] = 16
                        fr[:, :, :, w - bar_s:] = 16
                    elif name == "one_black_probe_frame_in_1000":  # a fade-in: every strip is letterbox, the edges converge, "no crop"
                        fr[::1000, 0] = 16
                    eng.hash_frames_letterbox_device(fr.data_ptr(), n, 16, w, h, oh.data_ptr(), stream=stream, d_crops=dcr.data_ptr())
                    ts, _ = timed_steps(lambda: eng.hash_frames_letterbox_device(fr.data_ptr(), n, 16, w, h, oh.data_ptr(), stream=stream,
                                                                                 d_crops=dcr.data_ptr()))
                    ms_l, ms_lo = med_min(ts)
                    c0 = [int(x) for x in dcr[0].cpu()]
                    kept = (w - c0[0] - c0[1]) * (h - c0[2] - c0[3])
                    res[name] = {"ms_per_step": ms_l, "ms_min": ms_lo, "frames_per_s_per_gpu": n * 16 / (ms_l * 1e-3), "crop_of_clip_0": c0,
                                 "box_GB_per_s": n * 16 * kept / (ms_l * 1e-3) / 1e9}
                    if fr is not base:
                        del fr
                hash_leg[key] = res
                del base, oh, dcr

            letterbox_leg(args.hash_hd_clips, 1920, 1080)
            # the headline's own frame shape with bars: boxes of small frames take one workgroup per clip with the DCT fused (round 5)
            letterbox_leg(min(args.hash_clips, 20000), 64, 64, key="letterbox_64x64", names=("no_bars", "top_bottom_bars", "side_bars"))
            if args.host_queue:
                torch.cuda.synchronize()
                hash_leg["host_queue_1080p"] = host_queue_leg()

    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(words, tol_int)
        if hash_leg is not None:
            hash_leg["cpu_baseline"] = cpu_baseline_hash()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if rank == 0:
        # order of the line: contract keys (+ roofline, cpu_baseline), then BASELINE configs[3] - strong scaling, the leg to read first
        # on a multi-GPU run - then the other legs, the long hash leg, and summary last (finish_line)
        for k in ("c4_10m_sharded", "ten_million", "c5_end_to_end", "dup_heavy", "cache_ingest", "refs_c5_shape", "windowed", "valu_backend"):
            if k in legs:
                out[k] = legs[k]
        if world > 1 and args.sp_leg:
            # the ranks are gone (or going: their memory is not needed - 288 GB per GPU); this process never execs
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            out["single_process"] = single_process_leg(args)
        if hash_leg is not None:
            out["hash"] = hash_leg
        print(finish_line(out), flush=True)


def leg_c5(args, torch, dist, vd, eng, dev, rank, world, use_dist, tol_int, stream, barrier, max_over_ranks, sum_over_ranks,
           leg_steps):
    n_c, n_r = args.c5_cands, args.c5_refs
    clo, chi = vd.split_range(n_c, rank, world)
    rlo, rhi = vd.split_range(n_r, rank, world)
    nc_l, nr_l = chi - clo, rhi - rlo
    need = (nc_l + nr_l) * 65536 + (n_c + n_r) * 700 + (4 << 30)
    free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    short = sum_over_ranks(1.0 if free < need else 0.0)
    if short:
        return {"skipped": f"needs {need / 2**30:.0f} GiB of free HBM per GPU"}
    torch.cuda.empty_cache()
    g = torch.Generator(device=dev)
    g.manual_seed(20250615 + rank)
    cand = torch.empty((nc_l, 16, 64, 64), dtype=torch.uint8, device=dev)
    for c0 in range(0, nc_l, 32768):  # bounded temporaries
        cand[c0:c0 + 32768] = torch.randint(0, 256, (min(32768, nc_l - c0), 16, 64, 64), dtype=torch.uint8, device=dev, generator=g)
    rng = np.random.default_rng(20250616 + rank)
    cd = np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=nc_l))).astype(np.int32)
    n_pl = min(nr_l // 2, nc_l)
    src = rng.choice(nc_l, size=n_pl, replace=False) if n_pl else np.zeros(0, np.int64)
    origin = np.concatenate([src, np.full(nr_l - n_pl, -1)]).astype(np.int64)
    origin = origin[rng.permutation(nr_l)]
    ref = torch.randint(0, 256, (nr_l, 16, 64, 64), dtype=torch.uint8, device=dev, generator=g)
    planted = np.nonzero(origin >= 0)[0]
    if len(planted):
        ref[torch.from_numpy(planted).to(dev)] = cand[torch.from_numpy(origin[planted]).to(dev)]
    rd = np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=nr_l))).astype(np.int32)
    rd[planted] = cd[origin[planted]]
    d_cd, d_rd = torch.from_numpy(cd).to(dev), torch.from_numpy(rd).to(dev)
    n_planted = int(round(sum_over_ranks(len(planted))))
    res = vd.hash_and_search_refs(eng, cand, d_cd, ref, d_rd, tol_int, stream=stream, as_lists=False)  # allocations, tables
    step_ms = []
    for _ in range(leg_steps):
        barrier()
        t0 = time.perf_counter()
        res = vd.hash_and_search_refs(eng, cand, d_cd, ref, d_rd, tol_int, stream=stream, as_lists=False)
        barrier()
        step_ms.append(max_over_ranks((time.perf_counter() - t0) * 1e3))
    dt_med, dt_min = med_min(step_ms)
    tm = {}
    vd.hash_and_search_refs(eng, cand, d_cd, ref, d_rd, tol_int, stream=stream, as_lists=False, timings=tm)  # phases, each behind a sync
    sr = eng.last_stats()
    stl = eng.last_timing()
    tm = {k: max_over_ranks(v) for k, v in sorted(tm.items())} if use_dist else tm
    pairs = sum_over_ranks(sr["pairs"])
    del cand, ref
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    offsets, members, ref_index = res[0]
    clips = n_c + n_r
    # phases_ms: one extra call with a device synchronisation after every phase (max over ranks); group_ms = hit gather + CSR groups +
    # order download on rank 0.  ms_per_step: median over the steps (each between two barriers)
    return {"workload": "BASELINE configs[4] end to end", "n_candidates": n_c, "n_references": n_r, "n_gpus": world, "scaling": "strong",
            "steps": leg_steps, "ms_per_step": dt_med, "ms_min": dt_min, "clips_per_s": clips / (dt_med * 1e-3),
            "frames_per_s": clips * 16 / (dt_med * 1e-3), "resident_GB": clips * 65536 / 1e9, "phases_ms": tm,
            "search_pairs": pairs, "search_kernel_ms": sr["kernel_ms"], "search_call_timing": stl,
            "groups": int(len(ref_index)), "members": int(len(members)), "planted_references": n_planted}


def leg_dup_heavy(args, torch, vdf, dev, local_rank, tol_int, sparse_words, leg_steps):
    """search() over a duplicate-dense database (make_dup_heavy: 10 % of the hashes in clusters of 2..200 near-copies sharing their centre's
    duration, log-uniform durations): the whole vdf_search_self_shards call on the device-resident database, on a one-device context and
    - "two_slots" - sharded over two slots of the same GPU, where the slots meet to OR the replay filter's bitmaps (csrc/multi.cpp:
    LocalExchange) and what comes down must stay s - 1 pairs per cluster.  sparse_same_windows: the headline's random hashes under the
    same sorted durations."""
    n = args.dup_heavy
    t_gen = time.perf_counter()
    words, dur, n_clusters, cluster_pairs = make_dup_heavy(n)
    t_gen = time.perf_counter() - t_gen

    def run(eng1, w, d, reps, slots=1):
        cut = [len(d) * k // slots for k in range(slots + 1)]
        tw = [torch.from_numpy(w[a:b].view(np.int64)).to(dev) for a, b in zip(cut[:-1], cut[1:])]
        td = [torch.from_numpy(d[a:b].view(np.int32)).to(dev) for a, b in zip(cut[:-1], cut[1:])]
        pw, pd, sz = [t.data_ptr() for t in tw], [t.data_ptr() for t in td], [len(t) for t in td]
        torch.cuda.synchronize()
        eng1.search_self_shards(pw, pd, sz, tol_int)  # allocations
        wall, tms, sts, ng, members = [], [], [], 0, 0
        for _ in range(reps):
            t0 = time.perf_counter()
            offsets, mem = eng1.search_self_shards(pw, pd, sz, tol_int, as_arrays=True)
            wall.append((time.perf_counter() - t0) * 1e3)
            tms.append(eng1.last_timing())
            sts.append(eng1.last_stats())
            ng = len(offsets) - 1
            members = len(mem)
        tm = {k: float(np.median([t[k] for t in tms])) for k in tms[0]}
        return med_min(wall), tm, sts[-1], ng, members

    out = {}
    eng1 = vdf.Engine(devices=[local_rank])
    try:
        ns = min(n, len(sparse_words))
        sp_dur = np.ascontiguousarray(dur[np.linspace(0, n - 1, ns).astype(np.int64)])  # the same (sorted) duration profile
        sp_ms, sp_tm, sp_st, sp_groups, _ = run(eng1, np.ascontiguousarray(sparse_words[:ns]), sp_dur, leg_steps)
        dn_ms, dn_tm, dn_st, dn_groups, dn_members = run(eng1, words, dur, leg_steps)
        out = {"n_hashes": n, "clusters": n_clusters, "pairs_inside_clusters": cluster_pairs, "steps": leg_steps,
               "ms_per_call": dn_ms[0], "ms_min": dn_ms[1], "pairs": dn_st["pairs"], "pairs_per_s": dn_st["pairs"] / (dn_ms[0] * 1e-3),
               "n_hits": dn_st["n_hits"], "n_launches": dn_st["n_launches"], "kernel_ms": dn_st["kernel_ms"],
               "timing": dn_tm, "suspect_queue_fill": dn_tm["suspects"] / max(dn_tm["suspect_capacity"], 1),
               "match_groups": dn_groups, "grouped_hashes": dn_members,
               "sparse_same_windows": {"ms_per_call": sp_ms[0], "ms_min": sp_ms[1], "n_hits": sp_st["n_hits"], "match_groups": sp_groups, "timing": sp_tm},
               "dense_over_sparse": dn_ms[0] / sp_ms[0], "generation_s": t_gen}
    finally:
        eng1.close()
    eng2 = vdf.Engine(devices=[local_rank, local_rank])
    try:
        t2_ms, t2_tm, t2_st, t2_groups, t2_members = run(eng2, words, dur, leg_steps, slots=2)
        out["two_slots"] = {"ms_per_call": t2_ms[0], "ms_min": t2_ms[1], "n_hits": t2_st["n_hits"], "hits_filtered": t2_tm["hits_filtered"],
                            "hits_downloaded": t2_st["n_hits"] - t2_tm["hits_filtered"],
                            "downloaded_fraction": (t2_st["n_hits"] - t2_tm["hits_filtered"]) / max(t2_st["n_hits"], 1),
                            "per_slot_hits_filtered": [eng2.device_timing(k)["hits_filtered"] for k in range(2)],
                            "match_groups": t2_groups, "grouped_hashes": t2_members, "timing": t2_tm}
        assert t2_groups == out["match_groups"] and t2_members == out["grouped_hashes"], "sharded and one-device searches disagree"
    finally:
        eng2.close()
    return out


def synth_paths(n, seed=20250620):
    """(blob u8, offsets u64[n + 1]): n paths like /srv/media/lib_07/show_0412/season_03/clip_00001234.mkv in arbitrary (HashMap) order over
    ~n / 40 directories - shared directory prefixes, as a real library has - built with numpy column arithmetic, no per-entry Python strings."""
    rng = np.random.default_rng(seed)
    tmpl = np.frombuffer(b"/srv/media/lib_00/show_0000/season_00/clip_00000000.mkv", np.uint8)
    blob = np.tile(tmpl, (n, 1))
    ids = rng.permutation(n).astype(np.int64)
    show = ids // 40
    for at, width, val in ((15, 2, show // 2000 % 100), (23, 4, show % 2000 * 5 % 10000), (35, 2, ids // 8 % 5), (43, 8, ids)):
        v = val.copy()
        for k in range(width - 1, -1, -1):
            blob[:, at + k] = 48 + v % 10
            v //= 10
    return blob.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(len(tmpl))


def synth_cache(n, seed=20250620, plant_every=1000):
    """The bytes of an app cache file with n Ok entries (vdf_cache_encode: bincode-2 standard(), key = src_path) plus the arrays they were
    made from: random hashes, log-uniform durations, every plant_every-th entry a near-copy (<= 300 flipped bits, same duration) of
    its predecessor - so the search has something to find.  Returns (bytes u8, hashes, durations, path blob, path offsets, planted)."""
    import ctypes as C

    from vid_dup_finder_lib_amd import _capi

    lib = _capi.load()
    rng = np.random.default_rng(seed)
    hashes = rng.integers(0, 2**64, size=(n, 16), dtype=np.uint64)
    hashes[:, 15] &= np.uint64((1 << 40) - 1)
    dur = np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=n))).astype(np.uint32)
    src = np.arange(0, n - 1, plant_every)
    if len(src):
        flips = rng.random((len(src), 1024), dtype=np.float32) < rng.uniform(0.0, 0.29, size=(len(src), 1)).astype(np.float32)
        hashes[src + 1] = hashes[src] ^ np.packbits(flips, axis=1, bitorder="little").view(np.uint64)
        hashes[src + 1, 15] &= np.uint64((1 << 40) - 1)
        dur[src + 1] = dur[src]
    blob, offs = synth_paths(n, seed)
    secs = rng.integers(1_600_000_000, 1_760_000_000, size=n, dtype=np.uint64)
    nanos = rng.integers(0, 10**9, size=n, dtype=np.uint32)
    out, out_len = C.c_void_p(), C.c_size_t()
    rc = lib.vdf_cache_encode(n, hashes.ctypes.data, dur.ctypes.data, offs.ctypes.data, blob.ctypes.data, secs.ctypes.data,
                              nanos.ctypes.data, C.byref(out), C.byref(out_len))
    if rc:
        raise RuntimeError(f"vdf_cache_encode failed: {rc}")
    data = np.ctypeslib.as_array((C.c_uint8 * out_len.value).from_address(out.value)).copy()
    lib.vdf_buffer_free(out)
    return data, hashes, dur, blob, offs, len(src)


def leg_cache_ingest(args, vdf, eng, tol_int):
    """SURVEY 8f N1 at the scale the row exists for (the load it replaces: base_fs_cache.rs:167-223 -> app_fns.rs:428-482): the bytes of an
    app cache with --cache-entries entries -> vdf_cache_decode_mt (all host threads by the library's own choice, and ONE thread beside it)
    -> vdf_search_cache_entries (path blob, durations and hashes up, Search::sort on the device from durations AND paths, search(), map back to entries).
    host_ms = everything but the search kernel's call: what a user waits for on top of the search itself."""
    import ctypes as C

    from vid_dup_finder_lib_amd import _capi
    from vid_dup_finder_lib_amd import cache as vc

    n = args.cache_entries
    t0 = time.perf_counter()
    data, hashes, dur, blob, offs, planted = synth_cache(n)
    t_gen = time.perf_counter() - t0
    del hashes, dur, blob, offs
    lib = _capi.load()

    def decode_ms(nt, reps):
        ts = []
        for _ in range(reps):
            soa = _capi.VdfCacheSoa()
            t1 = time.perf_counter()
            rc = lib.vdf_cache_decode_mt(data.ctypes.data, data.size, nt, C.byref(soa))
            ts.append((time.perf_counter() - t1) * 1e3)
            assert rc == 0 and soa.n_ok == n
            lib.vdf_cache_free(C.byref(soa))
        return med_min(ts)

    dec_auto = decode_ms(0, 3)
    dec_one = decode_ms(1, 1)
    t1 = time.perf_counter()
    cache = vc.decode_cache(data)
    t_dec = (time.perf_counter() - t1) * 1e3
    runs = []
    for _ in range(2):  # the first call allocates the device buffers
        t1 = time.perf_counter()
        offsets, members, refs, tm = vc.search_cache_arrays(cache, args.tolerance, engine=eng)
        tm["wall_ms"] = (time.perf_counter() - t1) * 1e3
        runs.append(tm)
    tm = runs[-1]
    st = eng.last_stats()
    host_ms = t_dec + tm["rank_ms"] + tm["upload_ms"] + tm["sort_ms"] + tm["map_ms"]
    return {"entries": n, "cache_MB": data.size / 1e6, "host_threads": os.cpu_count(), "cpu_quota": host_cpus()[2], "generation_s": t_gen,
            "decode_ms": dec_auto[0], "decode_ms_min": dec_auto[1], "decode_GB_per_s": data.size / (dec_auto[1] * 1e-3) / 1e9,
            "decode_one_thread_ms": dec_one[0], "decode_in_this_call_ms": t_dec,
            "search_cache_entries": {k: tm[k] for k in ("rank_ms", "upload_ms", "sort_ms", "search_ms", "map_ms", "total_ms", "wall_ms")},
            "first_call_total_ms": runs[0]["total_ms"], "host_ms": host_ms, "search_pairs": st["pairs"], "search_kernel_ms": st["kernel_ms"],
            "match_groups": int(len(offsets) - 1), "grouped_entries": int(len(members)), "planted_pairs": planted,
            "tolerance_int": tol_int}


if __name__ == "__main__":
    main()
