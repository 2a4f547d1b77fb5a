#!/bin/bash
# round 3, first GPU visit: the whole GPU suite (incl. the new 8-ring-view test), the default bench line, the
# 8-views-per-rank line, and a rocprofv3 kernel summary of the SEVEN-OP surface before it is reworked
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3a; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -14 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400; tail -2 $O/bench.err
timeout 300 python bench.py --views-per-rank 8 --steps 10 --warmup 3 --ramp-steps 20 --cpu-sample 0 --no-ops > $O/bench_v8.json 2> $O/bench_v8.err; tail -1 $O/bench_v8.json | cut -c1-400; tail -2 $O/bench_v8.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ko -- python $GRAFT_REPO_ROOT/tools/profile_step.py --mode ops --steps 40 > /tmp/ko.log 2>&1
find /tmp/ko -name "*kernel_stats.csv" -exec cp {} $O/ops_kernel_stats.csv \;
head -30 $O/ops_kernel_stats.csv | cut -c1-160
