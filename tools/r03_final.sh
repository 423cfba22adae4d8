#!/bin/bash
# round 3 GPU call: whole GPU suite, the default bench, smoke, the round's profiles (profiles/r03_*)
O=gpurun_out/r03r; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log; tail -3 $O/pytest.log
SECONDS=0; python bench.py > $O/bench.json 2> $O/bench.err; echo "bench wall $SECONDS s" >> $O/bench.err; tail -2 $O/bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r03r/prof > /dev/null 2>&1; ls $O/prof | wc -l
