#!/bin/bash
cd /tmp && export TMPDIR=/tmp; export PYTHONDONTWRITEBYTECODE=1
EGS_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -- python $GRAFT_REPO_ROOT/tools/profile_step.py --steps 160 > /tmp/trs.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/split
python $GRAFT_REPO_ROOT/tools/trace_timeline.py /tmp/trs --by-bwd > $GRAFT_REPO_ROOT/gpurun_out/split/timeline.txt 2>&1; tail -24 $GRAFT_REPO_ROOT/gpurun_out/split/timeline.txt
