"""GPU tier, SURVEY.md 8e / BASELINE configs 3 and 5: the driver's N = 8 bench line, proven on ONE GPU before it meets eight.
``tools/scale_smoke.sh`` runs ``bench.py --gpus 8`` (self-launch through torch.distributed.run, rendezvous on 127.0.0.1) with the
data-path collectives staged through the host (T2L_DIST_BACKEND=gloo: RCCL refuses two ranks on one device) and asserts
ranks_seen == 8, merged ids == the float64 oracle, and the three N > 1 side lines (weak_scaling_point, config5_coarse_plus_fine,
alt_query_sharded) present and error-free."""
import os
import os.path as osp
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REPO = osp.dirname(osp.dirname(osp.abspath(__file__)))


def test_bench_gpus_8_plumbing_on_one_gpu(tmp_path):
    # (eight bench processes on one box: without a cap every one of them starts a torch / BLAS thread per core — on an 8-core GPU box this
    # test took 65-145 s instead of 10)
    env = dict(os.environ, SCALE_SMOKE_NS="8", OMP_NUM_THREADS="2", MKL_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
    r = subprocess.run([osp.join(REPO, "tools", "scale_smoke.sh"), str(tmp_path)], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "scale_smoke N=8" in r.stdout and "-> OK" in r.stdout, r.stdout
