#!/bin/bash
# round 2, lab C: new tests (1 M fused/raw, directional derivatives, two-rank HIP exchange), crossbar reduction A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/labC; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -16 $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops"
EGS_DRAWB_RED=0 $B > $O/red0.json 2> $O/red0.err
EGS_DRAWB_RED=1 $B > $O/red1.json 2> $O/red1.err
EGS_DRAWB_RED=0 $B > $O/red0b.json 2> /dev/null
EGS_DRAWB_RED=1 $B > $O/red1b.json 2> /dev/null
EGS_FORCE_EXCHANGE=1 $B > $O/exch.json 2> $O/exch.err
python tools/lab_summ.py $O/red0.json $O/red1.json $O/red0b.json $O/red1b.json $O/exch.json
tail -2 $O/exch.err
python - <<PY
import json
d=json.loads(open("$O/exch.json").read().strip().splitlines()[-1]); print("exchange:", d.get("exchange"))
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_calib.hip -o $O/ubench_calib 2> $O/ubench_build.err
$O/ubench_calib > $O/ubench_calib.txt 2>&1; cat $O/ubench_calib.txt
