"""``bench.py --gpus 2`` end to end on the one-GPU box: EGS_BENCH_REHEARSAL=1 puts both ranks on cuda:0 and runs the
collectives over gloo (RCCL refuses two ranks on one device).  No number of such a line is a measurement; what is
pinned is that the N > 1 path of the bench -- the self-relaunch under torch.distributed.run, one ring view per rank,
the factored SH exchange chosen by ``factored_exchange_pays``, the per-rank step anatomy, ONE JSON line from rank 0 --
runs, in both exchange forms."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("form", ["auto", "off"])
def test_two_rank_bench_line_on_one_gpu(form):
    env = dict(os.environ, EGS_BENCH_REHEARSAL="1", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gaussians", "30000", "--width", "320",
           "--height", "192", "--steps", "3", "--warmup", "1", "--ramp-steps", "2", "--cpu-sample", "0",
           "--factored-sh", form]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]              # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and "rehearsal" in d
    assert d["config"]["views_per_step"] == 2 and d["scaling"] == "weak"
    ex = d["exchange"]
    assert len(ex["t_render_ms"]) == 2 and len(ex["t_exchange_ms"]) == 2
    assert ex["form"].startswith("SH gradient factored" if form == "auto" else "flat all-reduce")
    n = d["config"]["gaussians"]
    want = (12 * n + 16 + 44 * n) if form == "auto" else 236 * n       # bytes a rank receives over its one link
    assert abs(ex["link_bytes_per_rank"] - want) <= 64, (ex, want)
    assert d["value"] > 0 and d["redone_steps"] >= 0
