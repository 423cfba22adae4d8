#!/bin/bash
# One GPU call of a round (run via gpurun from the repo root): the whole GPU suite, the default bench, smoke, the search-call sweep.
# Usage: bash tools/gpu_round.sh <out_dir under gpurun_out> [profile]   (profile: also tools/profile_round.sh into <out_dir>/prof)
O=gpurun_out/${1:-round}; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -120 > $O/pytest.log; tail -6 $O/pytest.log   # no -x here: one failure must not hide the rest of the suite (the driver runs -x)
SECONDS=0; timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $? wall $SECONDS s" >> $O/bench.err; tail -2 $O/bench.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 300 python tools/sweep_search_calls.py > $O/search_calls.txt 2>&1; tail -25 $O/search_calls.txt
if [ "${2:-}" = "profile" ]; then timeout 1500 bash tools/profile_round.sh ${1:-round}/prof > /dev/null 2>&1; ls $O/prof | wc -l; fi
