#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run through gpurun); results land in gpurun_out/prof/.
#   bash tools/collect_profiles.sh            then copy what should be judged into profiles/ (r2_* names)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
# the default bench line (with the bounded CPU baseline and the seven-op figure), and the driver's flags
python $R/bench.py > $O/bench.json 2> $O/bench.err
python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_20_5.json 2>> $O/bench.err
# the same command under rocprofv3 kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --cpu-sample 0 --no-ops > $O/bench_under_rocprof.json 2>/tmp/ks.err
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
# one step as a timeline (no event brackets), after 150+ steps: steady-state clocks
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $R/tools/profile_step.py --steps 160 > /tmp/tr.log 2>&1
python $R/tools/trace_timeline.py /tmp/tr > $O/step_timeline.txt
# HBM traffic counters (separate passes, no tracing besides kernel-trace)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p1 -- python $R/tools/profile_step.py --steps 3 > /tmp/p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p2 -- python $R/tools/profile_step.py --steps 3 > /tmp/p2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/p1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/p2 -name "*counter_collection.csv" | head -1)) --all > $O/pmc_fetch_write.txt
cp /tmp/pmc_summary.json $O/pmc_fetch_write.json
# SQ counters of the step and of the calibration kernels
C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
C2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d /tmp/s1 -- python $R/tools/profile_step.py --steps 3 > /tmp/s1.log 2>&1
rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d /tmp/s2 -- python $R/tools/profile_step.py --steps 3 > /tmp/s2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/s1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/s2 -name "*counter_collection.csv" | head -1)) --all > $O/sq_counters.txt
cp /tmp/pmc_summary.json $O/sq_counters.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/ubench_calib.hip -o $O/ubench_calib 2> $O/ubench_build.err
$O/ubench_calib > $O/ubench_calib.txt 2>&1
rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d /tmp/ub -- $O/ubench_calib > /tmp/ub.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/ub -name "*counter_collection.csv" | head -1)) --all > $O/ubench_counters.txt
cp /tmp/pmc_summary.json $O/ubench_counters.json
cd $R && python tools/make_pmc_traffic.py $O/pmc_fetch_write.json $O/sq_counters.json $O/ubench_counters.json $O/pmc_traffic.json
rm -f $O/ubench_calib
tail -1 $O/bench.json | cut -c1-400
tail -1 $O/bench_20_5.json | cut -c1-200
head -14 $O/kernel_stats.csv
tail -3 $O/step_timeline.txt
