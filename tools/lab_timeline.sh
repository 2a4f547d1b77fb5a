#!/bin/bash
# kernel-trace timeline of one step (no event brackets): true durations and the gaps between launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $R/tools/profile_step.py --steps 160 > /tmp/tr.log 2>&1
python $R/tools/trace_timeline.py /tmp/tr > $O/timeline.txt; cat $O/timeline.txt
