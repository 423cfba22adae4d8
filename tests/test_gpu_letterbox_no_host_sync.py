"""Small letterboxed frames go from detect to hash without the host in between (vid_dup_finder_lib_amd/csrc/api.cpp:
letterbox_hash_device_locked; the reference detects, crops and hashes one clip in one pass, video_hash_builder.rs:188-204).

A child process runs with tests/cpp/hip_api_trace.c preloaded - a shim that records, in call order, every kernel launch, every HIP call
that makes the host wait for the device and every copy towards the host that the process makes - and reports what one
vdf_hash_frames_u8_letterbox_device[_async] call did, after a warm-up call (the first call of a frame size builds its tables).
Results are checked against the oracle in the same child."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_SRC = os.path.join(ROOT, "tests", "cpp", "hip_api_trace.c")
SHIM = os.path.join(ROOT, "tests", "cpp", "_build", "libhip_api_trace.so")

LAUNCH, STREAM_SYNC, EVENT_SYNC, DEVICE_SYNC, MEMCPY_SYNC, MEMCPY_D2H, MEMCPY_OTHER, MEMSET, QUERY = range(1, 10)
WAITS = {STREAM_SYNC, EVENT_SYNC, DEVICE_SYNC, MEMCPY_SYNC, QUERY}

CHILD = r"""
import ctypes, json, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
import vid_dup_finder_lib_amd as vdf
from oracle import vdf_oracle as orc
shim = ctypes.CDLL({shim!r})
w, h, n = {w}, {h}, {n}
rng = np.random.default_rng(7)
fr = rng.integers(40, 220, size=(n, 16, h, w), dtype=np.uint8)
fr[::3, :, :max(1, h // 8)] = 16
fr[::3, :, h - max(1, h // 9):] = 17
fr[1::3, :, :, :max(1, w // 8)] = 15
fr[1::3, :, :, w - max(1, w // 7):] = 15
res = [orc.hash_clip_letterbox(c) for c in fr]
want_h = np.stack([r[1] for r in res]); want_c = np.array([r[3] for r in res], np.uint32)
dev = torch.device("cuda", 0)
d = torch.from_numpy(fr).to(dev)
out = torch.zeros((n, 16), dtype=torch.int64, device=dev)
dcr = torch.zeros((n, 4), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
eng = vdf.Engine(0)
report = {{}}
for mode in ("async", "host_boxes"):
    def call():
        if mode == "async":
            eng.hash_frames_letterbox_device(d.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream, d_crops=dcr.data_ptr())
            return None
        return eng.hash_frames_letterbox_device(d.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream)
    call(); torch.cuda.synchronize()          # warm-up: tables of this frame size, buffers
    out.zero_(); dcr.zero_(); torch.cuda.synchronize()
    shim.vdf_trace_reset()
    crops = call()
    ev = [shim.vdf_trace_get(i) for i in range(shim.vdf_trace_count())]
    torch.cuda.synchronize()
    got_c = dcr.cpu().numpy().astype(np.uint32) if mode == "async" else crops
    report[mode] = {{"events": ev, "hashes_ok": bool(np.array_equal(out.cpu().numpy().view(np.uint64), want_h)),
                    "crops_ok": bool(np.array_equal(got_c, want_c)), "boxes": len({{tuple(c) for c in want_c}})}}
eng.close()
print("REPORT " + json.dumps(report))
"""


def _shim():
    if not os.path.exists(SHIM) or os.path.getmtime(SHIM) < os.path.getmtime(SHIM_SRC):
        os.makedirs(os.path.dirname(SHIM), exist_ok=True)
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", SHIM, SHIM_SRC, "-ldl"])
    return SHIM


def _run(w, h, n, env_extra=None):
    env = {k: v for k, v in os.environ.items() if not k.startswith("VDF_")}
    env["LD_PRELOAD"] = _shim()
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, shim=SHIM, w=w, h=h, n=n)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("REPORT ")][-1]
    return json.loads(line[len("REPORT "):])


# 64 x 64 and 48 x 36: the fused kernel (one launch for all but the last clip) + the careful route for the last clip;
# 160 x 90 and 256 x 128: detect kernels, then the cropped kernel reads the boxes on the device
@pytest.mark.parametrize("w,h", [(64, 64), (48, 36), (160, 90), (256, 128)])
def test_no_host_wait_between_detect_and_hash(w, h):
    rep = _run(w, h, 48)
    a = rep["async"]
    assert a["hashes_ok"] and a["crops_ok"] and a["boxes"] >= 3, a
    ev = a["events"]
    assert ev.count(LAUNCH) >= 1, ev
    # the whole call only QUEUES work: no wait of any kind, nothing copied towards the host
    assert not [e for e in ev if e in WAITS or e == MEMCPY_D2H], ev
    s = rep["host_boxes"]
    assert s["hashes_ok"] and s["crops_ok"], s
    ev = s["events"]
    last_launch = max(i for i, e in enumerate(ev) if e == LAUNCH)
    first_launch = min(i for i, e in enumerate(ev) if e == LAUNCH)
    # boxes wanted on the host: ONE copy and ONE wait, both behind the last launch - never between detect and hash
    assert not [e for e in ev[first_launch:last_launch] if e in WAITS or e == MEMCPY_D2H], ev
    tail = ev[last_launch + 1:]
    assert tail.count(MEMCPY_D2H) == 1 and sum(1 for e in tail if e in WAITS) == 1, ev


def test_the_shim_sees_the_round_5_route():
    """The same measurement on the route before round 6 (VDF_LB_HOST_PLAN): the boxes come down and the host waits BETWEEN the launches -
    so a green test above is the shim seeing nothing, not the shim seeing nothing of anything."""
    rep = _run(64, 64, 48, {"VDF_LB_HOST_PLAN": "1"})
    for mode in ("async", "host_boxes"):
        r = rep[mode]
        assert r["hashes_ok"] and r["crops_ok"], r
        ev = r["events"]
        first_launch = min(i for i, e in enumerate(ev) if e == LAUNCH)
        last_launch = max(i for i, e in enumerate(ev) if e == LAUNCH)
        between = ev[first_launch:last_launch]
        assert MEMCPY_D2H in between and any(e in WAITS for e in between), ev


def test_device_route_without_the_fused_kernel():
    """VDF_NO_LB_FUSED: frames of at most 64 x 64 take the detect kernels + the cropped kernel reading boxes on the device - same results,
    still no wait."""
    rep = _run(64, 64, 48, {"VDF_NO_LB_FUSED": "1"})
    a = rep["async"]
    assert a["hashes_ok"] and a["crops_ok"], a
    assert a["events"].count(LAUNCH) >= 3 and not [e for e in a["events"] if e in WAITS or e == MEMCPY_D2H], a["events"]
