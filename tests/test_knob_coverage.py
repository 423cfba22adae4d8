"""CPU check behind tests/test_gpu_knobs.py: every environment switch the library reads (getenv("VDF_...") anywhere in csrc/) is exercised by
a test, and the library reads its switches in ONE place per object (create_single / vdf_ctx_create_multi; vdf_hash_queue_create) - never per call: getenv is not
safe against a concurrent setenv, and a launcher must see the value its caller decided by."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vid_dup_finder_lib_amd", "csrc")


def _sources():
    return [p for pat in ("*.cpp", "*.hip", "*.h") for p in glob.glob(os.path.join(CSRC, pat))]


def test_every_env_knob_is_covered_by_a_test():
    knobs = set()
    for p in _sources():
        knobs |= set(re.findall(r'getenv\("(VDF_[A-Z0-9_]+)"\)', open(p).read()))
    assert len(knobs) >= 20, knobs
    tests = "".join(open(p).read() for p in glob.glob(os.path.join(ROOT, "tests", "*.py")) if not p.endswith("test_knob_coverage.py"))
    missing = sorted(k for k in knobs if k not in tests)
    assert not missing, f"switches without a test: {missing}"


def test_switches_are_read_once_per_context():
    where = {}
    for p in _sources():
        n = len(re.findall(r"getenv\(", open(p).read()))
        if n:
            where[os.path.basename(p)] = n
    # create_single and vdf_ctx_create_multi; the batching queue reads its one switch (VDF_QUEUE_SLOTS) when a QUEUE is made
    assert set(where) == {"api.cpp", "multi.cpp", "hash_queue.cpp"} and where["hash_queue.cpp"] == 1, where
    hq = open(os.path.join(CSRC, "hash_queue.cpp")).read()
    assert "getenv(" in hq[hq.index("int vdf_hash_queue_create("):hq.index("int vdf_hash_queue_submit(")]  # ... and nowhere near a submission
    api = open(os.path.join(CSRC, "api.cpp")).read()
    body = api[api.index("int create_single("):api.index("// search() over a sorted database that is already resident")]
    assert len(re.findall(r"getenv\(", body)) == where["api.cpp"]  # all of them inside create_single
