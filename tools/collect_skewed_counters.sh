#!/bin/bash
# Counter passes of the fused step on scene.skewed_scene right after reset_alpha (the segment path): HBM traffic and the
# VALU issue share of k_draw_seg / k_draw_bwd<SEG>, as tools/collect_profiles.sh does for the bench scene.
#   gpurun --timeout 900 -- 'timeout 850 bash tools/collect_skewed_counters.sh'   -> gpurun_out/prof_skewed/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_skewed; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
T="timeout 200"
P="python $R/tools/profile_step.py --scene skewed_reset --steps 6"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/q1 -- $P > /tmp/q1.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/q2 -- $P > /tmp/q2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/q1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/q2 -name "*counter_collection.csv" | head -1)) --all > $O/pmc_fetch_write.txt
cp /tmp/pmc_summary.json $O/pmc_fetch_write.json
C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
C2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
$T rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d /tmp/q3 -- $P > /tmp/q3.log 2>&1
$T rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d /tmp/q4 -- $P > /tmp/q4.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/q3 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/q4 -name "*counter_collection.csv" | head -1)) --all > $O/sq_counters.txt
cp /tmp/pmc_summary.json $O/sq_counters.json
cd $R && python tools/make_pmc_traffic.py $O/pmc_fetch_write.json $O/sq_counters.json - $O/pmc_traffic.json profiles/r6_valu_mix.json skewed_reset
python - <<PY
import json
d=json.load(open("$O/pmc_traffic.json"))["kernels"]
for k in ("k_draw_seg","k_draw_seg_fix","k_draw_seg_compose","k_draw_bwd_seg","k_seg_plan","k_bin_emit","k_preprocess_fwd"):
    if k in d: print(k, {a:d[k].get(a) for a in ("hbm_bytes_per_launch","valu_insts_per_launch","valu_issue_util","valu_frac","duration_us_in_counter_pass")})
print(sorted(d.keys()))
PY
tail -n 3 /tmp/q1.log; tail -n 3 /tmp/q3.log
