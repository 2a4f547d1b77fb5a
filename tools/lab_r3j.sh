#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tro -- python $R/tools/profile_step.py --mode ops --steps 60 > /tmp/tro.log 2>&1
python $R/tools/trace_timeline.py /tmp/tro > $O/ops_step_timeline.txt 2>&1; tail -48 $O/ops_step_timeline.txt | cut -c1-90
cd $R
EGS_FORCE_EXCHANGE=1 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops > $O/bench_exchange.json 2> $O/bench_exchange.err; tail -1 $O/bench_exchange.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('exchange'))"
tail -3 $O/bench_exchange.err
python tools/reach_stats.py 2>&1 | tail -2
