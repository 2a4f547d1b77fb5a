#!/bin/bash
# round 2, lab B: shared tile order + faster k_tile_order; occupancy sensitivity of k_draw_bwd; SQ counter calibration
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/labB; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -3 $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops"
$B > $O/base.json 2> $O/base.err
EGS_TILE_ORDER_F=0 EGS_TILE_ORDER_B=0 $B > $O/noorder.json 2>/dev/null
EGS_DRAWB_LDS_PAD=6144 $B > $O/bwd4w.json 2>/dev/null
EGS_DRAWB_LDS_PAD=9216 $B > $O/bwd3w.json 2>/dev/null
EGS_TILE_ORDER_F=4 EGS_TILE_ORDER_B=4 $B > $O/o44.json 2>/dev/null
python tools/lab_summ.py $O/base.json $O/noorder.json $O/bwd4w.json $O/bwd3w.json $O/o44.json
# ---- calibration of the SQ counters on single-instruction kernels
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_calib.hip -o $O/ubench_calib 2> $O/ubench_build.err
$O/ubench_calib > $O/ubench_calib.txt 2>&1; cat $O/ubench_calib.txt
cd /tmp && export TMPDIR=/tmp
C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d /tmp/ub1 -- $O/ubench_calib > /tmp/ub1.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/ub1 -name "*counter_collection.csv" | head -1)) --all > $O/ubench_counters.txt; cp /tmp/pmc_summary.json $O/ubench_counters.json
rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d /tmp/st1 -- python $R/tools/profile_step.py --steps 3 > /tmp/st1.log 2>&1
C2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d /tmp/st2 -- python $R/tools/profile_step.py --steps 3 > /tmp/st2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/st1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/st2 -name "*counter_collection.csv" | head -1)) > $O/step_counters.txt; cp /tmp/pmc_summary.json $O/step_counters.json
python - <<PY
import json
u=json.load(open("$O/ubench_counters.json")); s=json.load(open("$O/step_counters.json"))
def busy(c): return c["SQ_ACTIVE_INST_VALU"]*4/1024/(c["GRBM_GUI_ACTIVE"]/8)
for k,v in u.items():
    print("%-40s busy %.3f  active/inst %.3f  wave_cycles/inst*8 %.3f" % (k[-40:], busy(v), v["SQ_ACTIVE_INST_VALU"]/v["SQ_INSTS_VALU"], v["SQ_WAVE_CYCLES"]/v["SQ_INSTS_VALU"]))
for k,v in s.items():
    if "draw" in k or "preprocess" in k:
        print("%-40s busy %.3f  active/inst %.3f  insts %.1fM salu %.1fM wait_any %.2f wait_inst %.2f active_any %.2f of wave_cycles" % (k[5:45], busy(v), v["SQ_ACTIVE_INST_VALU"]/v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU"]/1e6, v["SQ_INSTS_SALU"]/1e6, v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_ANY"]/v["SQ_WAVE_CYCLES"]))
PY
