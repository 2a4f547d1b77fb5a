#!/bin/bash
# generic lab: GPU test suite + a few bench runs given as "NAME:ENV ASSIGNMENTS" arguments
#   bash tools/lab_run.sh OUTDIR "base:" "red0:EGS_DRAWB_RED=0" ...
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
if [ -z "$SKIP_TESTS" ]; then
  ( timeout 1800 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -9 $O/pytest.log
fi
B="python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops"
FILES=""
for spec in "$@"; do
  name="${spec%%:*}"; envs="${spec#*:}"
  env $envs $B $BENCH_ARGS > $O/$name.json 2> $O/$name.err
  FILES="$FILES $O/$name.json"
done
python tools/lab_summ.py $FILES
