"""The multi-rank code path of bench.py (barriers, max-over-ranks timing, the two gathers of results to rank 0, the
rank-0 line) rehearsed with two ranks that share the one GPU of the test box over gloo.  Its numbers are not a
measurement (the line says so); what is checked is that both ranks' blocks arrive on rank 0 in order and intact."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_rank_rehearsal_of_the_bench():
    env = dict(os.environ)
    env.update({"VSLAM_BENCH_SHARE_GPU": "1", "VSLAM_BENCH_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-dense", "--no-cpu-baseline"]
    # a fresh child process (never an exec from this one, which has initialised the GPU)
    run = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert "rehearsal" in out and "not a measurement" in out["rehearsal"]
    assert out["results_gathered_to_root"] is True
    filt = out["filter_results_gathered_to_root"]
    assert filt["filters_on_root"] == 32 and filt["filters_per_rank"] == 16
    assert filt["own_block_round_trip_exact"] is True
    assert filt["doubles_per_filter"] == 462 + 462 * 463 // 2
    kp = out["corner"]["keypoints_gathered_to_root"]
    assert kp["own_block_round_trip_exact"] is True and kp["frames_on_root"] == 2 * out["corner"]["frames_per_gpu_per_step"]
    # 2 ranks x 512 instances x 150 landmarks x 3 steps went into the headline
    assert out["value"] * out["ms_per_step"] * 1e-3 * 3 == pytest.approx(2 * 512 * 150 * 3, rel=1e-9)
