#!/bin/bash
# round 4, lab C: GPU suite (real asserts, numbers dumped on the side) + the default bench line
mkdir -p gpurun_out/r4c; O=gpurun_out/r4c
rm -f $O/grad_stats.jsonl
EGS_GRAD_STATS=$O/grad_stats.jsonl timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -80 > $O/tests.log
python bench.py > $O/bench.json 2> $O/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -6 $O/tests.log; tail -2 $O/smoke.log; cut -c1-300 $O/bench.json
