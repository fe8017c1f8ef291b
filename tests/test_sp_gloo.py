"""CPU, world_size 2, gloo: the sequence-parallel orchestration (magcache_amd/parallel.py) is
correct by construction -- sharded forward == unsharded oracle forward for FULL / SKIP / CALIB."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sequence_parallel_orchestration_gloo(tmp_path):
    out = tmp_path / "res.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "tests", "gloo_sp_worker.py"), str(out)]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.load(open(out))
    assert len(res) == 2
    for x in res:
        assert x["rel_full"] < 1e-2, x
        assert x["rel_skip"] < 1e-2, x
        assert x["calib_err"] < 1e-3, x
