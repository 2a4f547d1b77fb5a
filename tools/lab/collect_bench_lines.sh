#!/bin/bash
# The bench lines of collect_profiles.sh alone (after profiles/pmc_traffic.json has been stamped for the final
# sources, so that roofline.traffic is filled); results land in gpurun_out/prof/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
python $R/bench.py > $O/bench.json 2> $O/bench.err
python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_20_5.json 2>> $O/bench.err
python $R/bench.py --views-per-rank 8 --steps 10 --warmup 3 --ramp-steps 20 --cpu-sample 0 --no-ops > $O/bench_v8.json 2>> $O/bench.err
python $R/bench.py --views-per-rank 8 --view-streams 1 --steps 10 --warmup 3 --ramp-steps 20 --cpu-sample 0 --no-ops > $O/bench_v8_one_stream.json 2>> $O/bench.err
python $R/bench.py --views-per-rank 8 --factored-sh off --steps 10 --warmup 3 --ramp-steps 20 --cpu-sample 0 --no-ops > $O/bench_v8_rows_per_view.json 2>> $O/bench.err
EGS_FORCE_EXCHANGE=1 python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops --no-ring8 > $O/bench_forced_exchange.json 2>> $O/bench.err
EGS_FORCE_EXCHANGE=1 python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops --no-ring8 --factored-sh on > $O/bench_forced_exchange_factored.json 2>> $O/bench.err
EGS_FORCE_EXCHANGE=1 python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops --no-ring8 --overlap-exchange > $O/bench_forced_exchange_overlap.json 2>> $O/bench.err
for f in bench bench_20_5 bench_v8 bench_v8_one_stream bench_v8_rows_per_view bench_forced_exchange bench_forced_exchange_factored bench_forced_exchange_overlap; do tail -1 $O/$f.json | cut -c1-160; done
