#!/bin/bash
# other problem sizes of the same synthetic scene (robustness + DESIGN 5 "other sizes")
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/sizes; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --ramp-steps 60 --cpu-sample 0 --no-ops --no-ring8"
$B --gaussians 4000000 > $O/4m_1080.json 2>$O/4m.err
$B --width 3840 --height 2160 > $O/1m_4k.json 2>$O/4k.err
$B --gaussians 8000000 --width 3840 --height 2160 > $O/8m_4k.json 2>$O/8m.err
$B --sh-dim 3 > $O/1m_sh0.json 2>/dev/null
$B --gaussians 10000 --width 256 --height 256 > $O/10k_256.json 2>/dev/null
python tools/lab/lab_summ.py $O/4m_1080.json $O/1m_4k.json $O/8m_4k.json $O/1m_sh0.json $O/10k_256.json
tail -2 $O/8m.err
