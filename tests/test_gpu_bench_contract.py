"""bench.py's one-line JSON contract (what the driver parses), exercised with tiny sizes on the GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("backend", ["mfma", "valu"])
def test_bench_emits_the_contract_line(backend):
    env = dict(os.environ, VDF_SEARCH_BACKEND=backend)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                          "--n-hashes", "30000", "--hash-clips", "3000", "--hash-hd-clips", "20", "--ten-million", "200000",
                          "--c5-cands", "4000", "--c5-refs", "400", "--dup-heavy", "40000", "--cache-entries", "30000"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # exactly ONE JSON line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and "traffic" in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["sample"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["pairs"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6
    h = d["hash"]
    assert h["unit"] == "frames/s" and h["roofline"]["bound"] == "hbm" and h["cpu_baseline"]["kind"] == "port"
    lb = h["letterbox_full_hd"]  # detect + crop + hash in one call
    h_, w_ = 1080, 1920
    assert lb["no_bars"]["crop_of_clip_0"] == [0, 0, 0, 0] and lb["top_bottom_bars"]["crop_of_clip_0"] == [0, 0, int(h_ * 0.12), int(h_ * 0.12)]
    assert lb["side_bars"]["crop_of_clip_0"] == [int(w_ * 0.125), int(w_ * 0.125), 0, 0]
    assert lb["one_black_probe_frame_in_1000"]["crop_of_clip_0"] == [0, 0, 0, 0] and all(lb[k]["ms_per_step"] > 0 for k in lb if isinstance(lb[k], dict))
    ls = h["letterbox_64x64"]  # the headline's own frame shape with bars (small-frame crop kernel)
    assert ls["no_bars"]["crop_of_clip_0"] == [0, 0, 0, 0] and ls["top_bottom_bars"]["crop_of_clip_0"] == [0, 0, 7, 7] and ls["side_bars"]["crop_of_clip_0"] == [8, 8, 0, 0]
    assert d["match_groups"] > 0 and d["windowed"]["pairs"] > 0 and d["windowed"]["steps"] == 1
    assert d["windowed"]["waste_ratio"] >= 1.0
    rf = d["refs_c5_shape"]
    assert rf["hits"] == 50000 and 1.0 <= rf["waste_ratio"] < 1.5 and rf["kernel_ms"] > 0
    t = d["c4_10m_sharded"]  # BASELINE configs[3]; at one GPU the north_star's target leg (here at a test size)
    assert t["n_hashes"] == 200000 and t["pairs"] == 200000 * 199999 // 2 and t["match_groups"] >= t["planted_pairs"] - 1
    assert d["ten_million"]["ms_per_step"] == t["ms_per_step"] and t["scaling"] == "strong" and d["ten_million"]["same_as"] == "c4_10m_sharded"
    keys = list(d)
    assert keys[-1] == "summary" and keys.index("c4_10m_sharded") < keys.index("c5_end_to_end") < keys.index("hash")
    sm = d["summary"]  # the line's last <= 1.5 KB: the second half of BASELINE's metric, the 10 M leg, one figure per widened leg
    assert len(json.dumps(sm)) <= 1536 and lines[0].endswith(json.dumps(sm) + "}") and len(lines[0]) < 13000
    hs = sm["hash_summary"]
    assert hs["value"] == d["hash"]["value"] and hs["roofline"]["frac"] == d["hash"]["roofline"]["frac"] and hs["cpu_baseline"]["cores"] >= 1
    assert sm["ten_million"]["ms"] == t["ms_per_step"] and sm["ten_million"]["planted_found"] == t["match_groups"]
    assert sm["c5_end_to_end"]["ms_per_step"] == d["c5_end_to_end"]["ms_per_step"] and sm["cache_ingest"]["host_ms"] == d["cache_ingest"]["host_ms"]
    assert set(sm["letterbox_64x64_ms"]) == {"no_bars", "top_bottom_bars", "side_bars"}
    hq = d["hash"]["host_queue_1080p"]  # compiled caller threads through vdf_hash_queue (a child process), GB/s of the PCIe link
    assert hq["letterbox"]["wrong"] == 0 and hq["plain"]["wrong"] == 0 and hq["plain"]["link_GB_per_s"] > 10 and hq["caller_threads"] == 32
    assert set(sm["host_queue_1080p_link_GB_per_s"]) == {"letterbox", "plain"}
    # the GPU's clock and power while the timed steps ran (sysfs; null where the box does not show them)
    for ck in (d["roofline"]["clock"], d["hash"]["roofline"]["clock"]):
        assert set(ck) == {"sclk_mhz_median", "power_w_median", "samples"}
        assert ck["sclk_mhz_median"] is None or 100 < ck["sclk_mhz_median"] < 3500
    ci = d["cache_ingest"]  # cache bytes -> groups, phase by phase
    assert ci["entries"] == 30000 and ci["match_groups"] >= ci["planted_pairs"] - 1 and ci["decode_ms"] > 0 and ci["host_ms"] > 0
    assert set(ci["search_cache_entries"]) >= {"rank_ms", "upload_ms", "sort_ms", "search_ms", "map_ms", "total_ms"}
    c5 = d["c5_end_to_end"]  # BASELINE configs[4] end to end (here at a test size)
    assert c5["groups"] == c5["planted_references"] == 200 and c5["members"] == 200 and c5["clips_per_s"] > 0
    assert set(c5["phases_ms"]) == {"hash_ms", "all_gather_ms", "sort_ms", "search_ms", "group_ms"}
    dh = d["dup_heavy"]  # the duplicate-dense leg
    assert dh["n_hits"] >= dh["pairs_inside_clusters"] > 0 and dh["match_groups"] == dh["clusters"]
    assert dh["grouped_hashes"] == 4000 and dh["n_launches"] >= 1 and dh["timing"]["total_ms"] > 0
    assert dh["sparse_same_windows"]["n_hits"] < dh["n_hits"] and 0 <= dh["suspect_queue_fill"] <= 1.0
    two = dh["two_slots"]  # the same database sharded over two slots: same groups; the slots' filter (when the list is long enough) agrees
    assert two["match_groups"] == dh["match_groups"] and two["n_hits"] == dh["n_hits"] and 0 < two["downloaded_fraction"] <= 1.0
    assert c5["ms_min"] <= c5["ms_per_step"] and d["windowed"]["ms_min"] <= d["windowed"]["ms"]
    assert d["n_launches"] == 1 and d["suspects"] >= 0
    assert d["cpu_baseline"]["all_cores"]["in_reference"] is False and d["cpu_baseline"]["all_cores"]["cores"] >= 1
    if backend == "mfma":
        # frac prices the MFMA work actually executed; the algorithmic figure is reported beside it and is never smaller
        assert r["algorithmic_frac"] >= r["frac"] and "traffic_source" in r
        v = d["valu_backend"]
        assert v["match_groups"] == d["match_groups"] and v["pairs_per_s"] > 0 and 0 < v["valu"]["frac"] < 1.2
        assert "x_of_hbm_model" in d["hbm_operand_stream_model"] and "frac" not in d["hbm_operand_stream_model"]
    else:
        assert "valu_backend" not in d
