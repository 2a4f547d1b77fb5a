#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run through gpurun); results land in gpurun_out/prof/.
#   bash tools/collect_profiles.sh            then copy what should be judged into profiles/ (r4_* names)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
# the default bench line (with the CPU baseline and the seven-op figure), the driver's flags, eight views per rank
python $R/bench.py > $O/bench.json 2> $O/bench.err
python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_20_5.json 2>> $O/bench.err
python $R/bench.py --views-per-rank 8 --steps 10 --warmup 3 --ramp-steps 20 --cpu-sample 0 --no-ops > $O/bench_v8.json 2>> $O/bench.err
python $R/bench.py --views-per-rank 8 --view-streams 1 --steps 10 --warmup 3 --ramp-steps 20 --cpu-sample 0 --no-ops > $O/bench_v8_one_stream.json 2>> $O/bench.err
python $R/bench.py --views-per-rank 8 --factored-sh off --steps 10 --warmup 3 --ramp-steps 20 --cpu-sample 0 --no-ops > $O/bench_v8_rows_per_view.json 2>> $O/bench.err
EGS_FORCE_EXCHANGE=1 python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops --no-ring8 > $O/bench_forced_exchange.json 2>> $O/bench.err
EGS_FORCE_EXCHANGE=1 python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops --no-ring8 --factored-sh on > $O/bench_forced_exchange_factored.json 2>> $O/bench.err
EGS_FORCE_EXCHANGE=1 python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops --no-ring8 --overlap-exchange > $O/bench_forced_exchange_overlap.json 2>> $O/bench.err
# the same command under rocprofv3 kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --cpu-sample 0 --no-ops --no-ring8 > $O/bench_under_rocprof.json 2>/tmp/ks.err
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
# eight views per step on this GPU (configs[3]'s workload; SH gradient factored, four streams): kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k8 -- python $R/bench.py --views-per-rank 8 --steps 10 --warmup 3 --ramp-steps 20 --cpu-sample 0 --no-ops --no-prof > $O/bench_v8_under_rocprof.json 2>/tmp/k8.err
find /tmp/k8 -name "*kernel_stats.csv" -exec cp {} $O/v8_kernel_stats.csv \;
# the seven-op surface: kernel stats of 40 steps of GSFunction(mode="ops")
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ko -- python $R/tools/profile_step.py --mode ops --steps 40 > /tmp/ko.log 2>&1
find /tmp/ko -name "*kernel_stats.csv" -exec cp {} $O/ops_kernel_stats.csv \;
# one step as a timeline (no event brackets), after 150+ steps: steady-state clocks
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $R/tools/profile_step.py --steps 160 > /tmp/tr.log 2>&1
python $R/tools/trace_timeline.py /tmp/tr > $O/step_timeline.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/tro -- python $R/tools/profile_step.py --mode ops --steps 60 > /tmp/tro.log 2>&1
python $R/tools/trace_timeline.py /tmp/tro > $O/ops_step_timeline.txt 2>&1
# one optimizer step (GSRawFunction + HIP loss + FusedAdam) as a timeline: as Trainer.step runs it (SH gradient factored and
# consumed by the optimizer, loss kernels hand over dL/dimage) and with the rows + the loss as an autograd node
rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -- python $R/tools/profile_step.py --train --factored --steps 120 > /tmp/tt.log 2>&1
python $R/tools/trace_timeline.py /tmp/tt > $O/train_step_timeline.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tu -- python $R/tools/profile_step.py --train --steps 120 > /tmp/tu.log 2>&1
python $R/tools/trace_timeline.py /tmp/tu > $O/train_step_timeline_rows.txt 2>&1
# HBM traffic counters (separate passes, no tracing besides kernel-trace), fused step and seven-op step
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p1 -- python $R/tools/profile_step.py --steps 3 > /tmp/p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p2 -- python $R/tools/profile_step.py --steps 3 > /tmp/p2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/p1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/p2 -name "*counter_collection.csv" | head -1)) --all > $O/pmc_fetch_write.txt
cp /tmp/pmc_summary.json $O/pmc_fetch_write.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/q1 -- python $R/tools/profile_step.py --mode ops --steps 3 > /tmp/q1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/q2 -- python $R/tools/profile_step.py --mode ops --steps 3 > /tmp/q2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/q1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/q2 -name "*counter_collection.csv" | head -1)) --all > $O/ops_pmc_fetch_write.txt
cp /tmp/pmc_summary.json $O/ops_pmc_fetch_write.json
# ... and of one optimizer step as Trainer.step runs it (factored SH gradient consumed by FusedAdam)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/r1 -- python $R/tools/profile_step.py --train --factored --steps 3 > /tmp/r1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/r2 -- python $R/tools/profile_step.py --train --factored --steps 3 > /tmp/r2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/r1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/r2 -name "*counter_collection.csv" | head -1)) --all > $O/train_pmc_fetch_write.txt
# SQ counters of the step
C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
C2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d /tmp/s1 -- python $R/tools/profile_step.py --steps 3 > /tmp/s1.log 2>&1
rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d /tmp/s2 -- python $R/tools/profile_step.py --steps 3 > /tmp/s2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/s1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/s2 -name "*counter_collection.csv" | head -1)) --all > $O/sq_counters.txt
cp /tmp/pmc_summary.json $O/sq_counters.json
cd $R && python tools/make_pmc_traffic.py $O/pmc_fetch_write.json $O/sq_counters.json - $O/pmc_traffic.json
python tools/make_pmc_traffic.py $O/ops_pmc_fetch_write.json - - $O/ops_pmc_traffic.json
tail -1 $O/bench.json | cut -c1-400
tail -1 $O/bench_20_5.json | cut -c1-200
tail -1 $O/bench_v8.json | cut -c1-200
tail -1 $O/bench_v8_one_stream.json | cut -c1-200
tail -1 $O/bench_forced_exchange.json | cut -c1-200
head -14 $O/kernel_stats.csv | cut -c1-200
tail -3 $O/step_timeline.txt
tail -3 $O/ops_step_timeline.txt
