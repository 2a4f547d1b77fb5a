#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run through gpurun); results land in gpurun_out/prof/.
#   gpurun --timeout 1000 -- 'timeout 950 bash tools/collect_profiles.sh'   then copy what should be judged into profiles/
# Every step runs under its own `timeout`: a counter pass that hangs (TA_* / TCC_* passes did in round 5: 15 GPU-minutes)
# must not eat the round's GPU budget.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
T="timeout 240"
# ONLY_COUNTERS=1: just the counter passes at the end (pmc_traffic.json), e.g. after tools/profile_step.py changed
if [ -z "$ONLY_COUNTERS" ]; then
# the default bench line (CPU baseline, seven-op figure, eight ring views, the two skewed scenes) and the driver's flags
$T python $R/bench.py > $O/bench.json 2> $O/bench.err
$T python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_20_5.json 2>> $O/bench.err
# the headline command under rocprofv3 kernel trace + stats; the same on the heavy-tailed scene after reset_alpha
$T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --cpu-sample 0 --no-ops --no-ring8 --no-skewed > $O/bench_under_rocprof.json 2>/tmp/ks.err
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
$T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kr -- python $R/bench.py --scene skewed_reset --steps 20 --cpu-sample 0 --no-ops --no-ring8 > $O/bench_skewed_reset_under_rocprof.json 2>/tmp/kr.err
find /tmp/kr -name "*kernel_stats.csv" -exec cp {} $O/skewed_reset_kernel_stats.csv \;
$T python $R/bench.py --scene skewed --steps 20 --cpu-sample 0 --no-ops --no-ring8 > $O/bench_skewed.json 2>> $O/bench.err
EGS_SEGMENTS=0 $T python $R/bench.py --scene skewed_reset --steps 10 --cpu-sample 0 --no-ops --no-ring8 > $O/bench_skewed_reset_unsplit.json 2>> $O/bench.err
# one step as a timeline (no event brackets), after 150+ steps: steady-state clocks
$T rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $R/tools/profile_step.py --steps 160 > /tmp/tr.log 2>&1
python $R/tools/trace_timeline.py /tmp/tr > $O/step_timeline.txt
fi
# HBM traffic counters (separate passes, no tracing besides kernel-trace) and SQ counters of the fused step
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p1 -- python $R/tools/profile_step.py --steps 3 > /tmp/p1.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p2 -- python $R/tools/profile_step.py --steps 3 > /tmp/p2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/p1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/p2 -name "*counter_collection.csv" | head -1)) --all > $O/pmc_fetch_write.txt
cp /tmp/pmc_summary.json $O/pmc_fetch_write.json
C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
C2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
$T rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d /tmp/s1 -- python $R/tools/profile_step.py --steps 3 > /tmp/s1.log 2>&1
$T rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d /tmp/s2 -- python $R/tools/profile_step.py --steps 3 > /tmp/s2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/s1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/s2 -name "*counter_collection.csv" | head -1)) --all > $O/sq_counters.txt
cp /tmp/pmc_summary.json $O/sq_counters.json
cd $R && python tools/make_pmc_traffic.py $O/pmc_fetch_write.json $O/sq_counters.json - $O/pmc_traffic.json profiles/r6_valu_mix.json
tail -1 $O/bench.json | cut -c1-300
tail -1 $O/bench_20_5.json | cut -c1-200
head -14 $O/kernel_stats.csv | cut -c1-160
head -12 $O/skewed_reset_kernel_stats.csv | cut -c1-160
tail -3 $O/step_timeline.txt
