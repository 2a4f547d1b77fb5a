#!/bin/bash
# one step as a kernel timeline after 160 steps (steady clocks), fused and seven-op
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-tl}; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $R/tools/profile_step.py --steps 160 > /tmp/tr.log 2>&1
python $R/tools/trace_timeline.py /tmp/tr > $O/step_timeline.txt; cat $O/step_timeline.txt
if [ -n "$OPS" ]; then
rocprofv3 --kernel-trace --output-format csv -d /tmp/tro -- python $R/tools/profile_step.py --mode ops --steps 60 > /tmp/tro.log 2>&1
python $R/tools/trace_timeline.py /tmp/tro > $O/ops_step_timeline.txt 2>&1; tail -45 $O/ops_step_timeline.txt
fi
