#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/labD; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -12 $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops"
EGS_DRAWB_RED=0 $B > $O/red0.json 2> $O/red0.err
EGS_DRAWB_RED=1 $B > $O/red1.json 2> $O/red1.err
EGS_DRAWB_RED=0 $B > $O/red0b.json 2> /dev/null
EGS_DRAWB_RED=1 $B > $O/red1b.json 2> /dev/null
python tools/lab_summ.py $O/red0.json $O/red1.json $O/red0b.json $O/red1b.json
