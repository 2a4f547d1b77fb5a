#!/bin/bash
# end-of-round check: the GPU suite, smoke(), the driver's bench command, the default one, the forced exchange
cd $GRAFT_REPO_ROOT; O=gpurun_out/final; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
( timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -14 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; tail -1 $O/bench_20_5.json | cut -c1-330
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-330
EGS_FORCE_EXCHANGE=1 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops > $O/bench_exchange.json 2> $O/bench_exchange.err; tail -1 $O/bench_exchange.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('exchange'))"
tail -3 $O/bench_exchange.err
