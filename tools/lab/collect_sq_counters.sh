#!/bin/bash
# SQ instruction / busy counters of one forward+backward step (separate rocprofv3 --pmc passes; run via gpurun).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
i=0
DIRS=""
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/sq$i -- python $R/tools/profile_step.py --steps 3 > /tmp/sq$i.log 2>&1
  D=$(dirname $(find /tmp/sq$i -name "*counter_collection.csv" | head -1))
  DIRS="$DIRS $D"
done
python $R/tools/pmc_summary.py $DIRS > $R/gpurun_out/prof/sq_counters.txt
cp /tmp/pmc_summary.json $R/gpurun_out/prof/sq_counters.json
cat $R/gpurun_out/prof/sq_counters.txt
