#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the files committed under profiles/:
   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (vdf kernels only)
   <tag>_pmc_summary.json   per-kernel counter sums per launch
   pmc_traffic.json         HBM bytes per launch for the dominant kernels (read by bench.py: roofline.traffic)
Usage: python tools/summarize_profiles.py gpurun_out/<dir> <tag>"""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

stats = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
if stats:
    rows = list(csv.reader(open(stats[0])))
    keep = [rows[0]] + [r for r in rows[1:] if "vdf::" in r[0]]
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f).writerows(keep)
stats_h = glob.glob(os.path.join(src, "kt_headline", "*", "*_kernel_stats.csv"))
if stats_h:  # the headline's kernels alone: one launch shape per kernel, so AverageNs is the per-launch time
    rows = list(csv.reader(open(stats_h[0])))
    keep = [rows[0]] + [r for r in rows[1:] if "vdf::" in r[0]]
    with open(os.path.join(out, f"{tag}_kernel_stats_headline.csv"), "w", newline="") as f:
        csv.writer(f).writerows(keep)
bh = os.path.join(src, "bench_headline_under_profiler.json")
if os.path.exists(bh) and os.path.getsize(bh):
    open(os.path.join(out, f"{tag}_bench_headline_under_profiler.json"), "w").write(open(bh).read())
bj = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(bj) and os.path.getsize(bj):
    open(os.path.join(out, f"{tag}_bench_under_profiler.json"), "w").write(open(bj).read())

summary = {}
for d in sorted(glob.glob(os.path.join(src, "*"))):
    fs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "vdf::" not in k:
            continue
        k = k.split("(")[0].replace("void ", "").replace("vdf::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    summary[os.path.basename(d)] = {k: {"launches": len(launches[k]), **{c: v / len(launches[k]) for c, v in cs.items()}}
                                    for k, cs in agg.items()}
json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)


def traffic(fetch_dir, write_dir, kernel, fetch_factor, note):
    f = summary.get(fetch_dir, {}).get(kernel, {}).get("FETCH_SIZE")
    w = summary.get(write_dir, {}).get(kernel, {}).get("WRITE_SIZE")
    if f is None or w is None:
        return None
    # rocprofv3 reports KB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 128-B requests at 64 B, i.e. half the
    # bytes of a wide (16 B/lane) coalesced read -> factor 2 where the kernel streams with dwordx4 loads.
    return {"hbm_bytes_per_launch": f * 1024 * fetch_factor + w * 1024, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
            "fetch_factor": fetch_factor, "note": note}


t = {}
for k in summary.get("fetch_search", {}):
    if k.startswith("hamming_tile_kernel"):
        t["hamming_tile_kernel"] = traffic("fetch_search", "write_search", k, 1,
                                           "candidates arrive through scalar-cache line fills (64-B requests): no x2 correction applied")
    if k.startswith("hamming_mfma2_kernel"):
        t["hamming_mfma2_kernel"] = traffic("fetch_search", "write_search", k, 2,
                                            "candidate tiles stream through buffer_load ... lds (16 B/lane): FETCH_SIZE doubled per the gfx950 correction")
for k in summary.get("fetch_hash", {}):
    if k.startswith("resize_dct_hash_persistent_kernel") or k.startswith("resize_dct_hash_fused_kernel"):  # keyed by the kernel's real name
        t[k.split("<")[0].split("(")[0]] = traffic("fetch_hash", "write_hash", k, 2,
                                                    "16 B/lane streaming reads: FETCH_SIZE doubled per the gfx950 correction; "
                                                    "launch = tools/bench_hash.py, 100 000 clips of 16 x 64 x 64")
for tag_, fd, wd in (("resize_mfma_frame_wavestream_kernel@1920x1080", "fetch_hd", "write_hd"),
                     ("resize_mfma_frame_stream_kernel@480x270", "fetch_sd", "write_sd"),
                     ("resize_mfma_frame_ksplit_kernel@3840x2160", "fetch_uhd", "write_uhd")):
    for k in summary.get(fd, {}):
        if k.startswith(tag_.split("@")[0]):
            t[tag_] = traffic(fd, wd, k, 2, "frames stream through buffer_load ... lds (16 B/lane): FETCH_SIZE doubled per the gfx950 correction; "
                                            "launch = tools/bench_hash.py at the bench leg's shape")
for k in summary.get("fetch_lbs", {}):
    if k.startswith("letterbox_resize_dct_hash_small_kernel"):
        t["letterbox_resize_dct_hash_small_kernel"] = traffic("fetch_lbs", "write_lbs", k, 2,
                                                              "16 B/lane reads (pixels non-temporal, probes and tables plain): FETCH_SIZE doubled per the gfx950 "
                                                              "correction; launch = tools/bench_letterbox_small.py, 20 000 clips of 16 x 64 x 64 (average over its six bar patterns)")
stats_l = glob.glob(os.path.join(src, "kt_lbs", "*", "*_kernel_stats.csv"))
if stats_l:
    rows = list(csv.reader(open(stats_l[0])))
    keep = [rows[0]] + [r for r in rows[1:] if "vdf::" in r[0]]
    with open(os.path.join(out, f"{tag}_kernel_stats_letterbox_small.csv"), "w", newline="") as f:
        csv.writer(f).writerows(keep)
# The file is rebuilt from THIS profile run only (entries of kernels that no longer exist must not linger) and records which binary
# it belongs to: bench.py reports a traffic figure only when the library it loads has this sha256 (tools/profile_round.sh writes the
# sha of the .so it profiled next to the counters; the in-tree file is the fallback - the same file, gpurun ships it).
import hashlib

sha = None
sha_file = os.path.join(src, "lib_sha256.txt")
if os.path.exists(sha_file):
    sha = open(sha_file).read().split()[0]
else:
    sha = hashlib.sha256(open(os.path.join(root, "vid_dup_finder_lib_amd", "libvdf_hip.so"), "rb").read()).hexdigest()
t = {k: v for k, v in t.items() if v}
t["lib_sha256"] = sha
t["profiled_with"] = f"tools/profile_round.sh -> {os.path.basename(os.path.normpath(src))} -> tools/summarize_profiles.py {tag}"
json.dump(t, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(t, indent=1))
