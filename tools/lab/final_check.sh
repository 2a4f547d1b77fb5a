cd $GRAFT_REPO_ROOT
O=gpurun_out/r6z; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400
EGS_FORCE_EXCHANGE=1 timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops --no-ring8 --no-skewed --no-train --no-uhd > $O/bench_forced_exchange_flat.json 2>> $O/bench.err
EGS_FORCE_EXCHANGE=1 timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops --no-ring8 --no-skewed --no-train --no-uhd --factored-sh on > $O/bench_forced_exchange_factored.json 2>> $O/bench.err
EGS_BENCH_REHEARSAL=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --cpu-sample 0 > $O/bench_rehearsal_2ranks_gloo.json 2>> $O/bench.err
for f in bench_forced_exchange_flat bench_forced_exchange_factored bench_rehearsal_2ranks_gloo; do tail -1 $O/$f.json | cut -c1-200; done
tail -5 $O/bench.err
