"""RCCL on the one GPU the test box has: world size 1, backend "nccl".  See rccl_world1_worker.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _env(port):
    return dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_rccl_world_of_one_runs_the_sequence_parallel_collectives(tmp_path):
    out = tmp_path / "rccl.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py"), str(out)], env=_env(29581),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.load(open(out))
    assert res["backend"] == "nccl" and res["world"] == 1 and res["allreduce_ones"] == 1.0
    assert res["inplace_selftest"] is True
    assert res["inplace_async_gather_beside_attention_bit_identical"] is True
    assert res["head_token_gather_ok"] and res["calib_allreduce_ok"]
    # the collective inside the library: ONE C call per sharded forward on the engine's own RCCL communicator
    assert res["c_path_attached"] and "rccl" in res["c_path_info"] and "1 ranks" in res["c_path_info"], res
    assert res["c_path_chunks"] == [4, 4]
    for name, err in res["c_path_rel"].items():
        assert err < 1e-3, (name, err)               # same kernels; the attention output takes one extra lse round trip
    assert res["c_path_calib_err"] < 5e-4
    assert res["c_path_equals_callback_path"] is True
    assert res["sp_wait_pairs"] == 2 * 4             # layers x rounds


def test_bench_line_through_rccl_at_world_one():
    """bench.py with MC_BENCH_FORCE_DIST=1: the N > 1 communicator code (init on the device, all-reduce of ones, barrier
    between the timed regions, teardown) runs on RCCL with one rank; the line reports rccl_world = 1."""
    env = dict(_env(29583), MC_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "0",
                        "--no_cpu_baseline", "--no_kernels"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rccl_world"] == 1 and line["comm_backend"] == "nccl" and line["n_gpus"] == 1
    assert line["value"] > 0 and line["forwards_total"] == 20
