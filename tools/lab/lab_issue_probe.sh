#!/bin/bash
# Which issue port limits k_draw?  Library variants with N dummy scalar / vector instructions per (tile, entry)
# (-DEGS_DRAW_DUMMY_SALU=N / -DEGS_DRAW_DUMMY_VALU=N) or with two LDS broadcast reads per entry instead of three
# (-DEGS_DRAW_PROBE_NOK: wrong colours, timing only), built into tools/variants/libegs_<name>.so, are swapped in
# on the box:   hipcc <Makefile FLAGS> -D... -c csrc/egs_draw.hip (after git apply tools/lab/variants/draw_issue_probes.patch) -o /tmp/r.o; hipcc -shared -o tools/variants/libegs_X.so /tmp/r.o <other .o>
# Measured (1 M / 1080p, k_draw 171 us): +4 SALU 178, +8 SALU 193; +4 VALU 173, +8 VALU 185; one LDS read less 162.
cd $GRAFT_REPO_ROOT; O=gpurun_out/probe; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
cp easygaussiansplatting_amd/libegs_hip.so /tmp/libegs_base.so
B="python bench.py --steps 100 --cpu-sample 0 --no-ops"
for v in ${VARIANTS:-base salu4 valu4 salu8 valu8 base}; do
  if [ $v = base ]; then cp /tmp/libegs_base.so easygaussiansplatting_amd/libegs_hip.so; else cp tools/variants/libegs_$v.so easygaussiansplatting_amd/libegs_hip.so; fi
  $B > $O/$v.json 2> $O/$v.err
  python - $O/$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("%-6s ms %.4f  k_draw %.1f us  k_draw_bwd %.1f us" % (sys.argv[2], d["ms_per_step"], k["k_draw"]["avg_us"], k["k_draw_bwd"]["avg_us"]))
PY
done
cp /tmp/libegs_base.so easygaussiansplatting_amd/libegs_hip.so
