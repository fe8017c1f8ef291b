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


def test_bench_whole_multi_gpu_flow_on_one_gpu_through_the_library_side_rccl():
    """bench.py with MC_BENCH_FORCE_DIST=1 MC_BENCH_FORCE_SP=1: everything the driver's --gpus 8 launch runs -- layout, a sharded
    engine (the sequence-parallel phase path with a world of one), the start-up self-check of overlapped vs serialised
    forwards, the library-side RCCL communicator with ONE mc_forward_sp_rccl per forward, the N > 1 report (sp_collective,
    sp_gather_wait, rank 0's launch classes, per-GPU roofline) -- on the one GPU there is, with the real collective library.
    The final latents equal the plain single-GPU run's."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "0", "--no_cpu_baseline",
            "--no_table"]
    r = subprocess.run(base, env=dict(_env(29585), MC_BENCH_FORCE_DIST="1", MC_BENCH_FORCE_SP="1"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    one = subprocess.run(base, env=_env(29587), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-3000:]
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert got["rccl_world"] == 1 and got["comm_backend"] == "nccl" and got["layout"] == "sp"
    assert "RCCL communicator inside the library" in got["sp_collective"] and "rccl" in got["sp_collective"], got["sp_collective"]
    assert got["sp_chunks"] == 4 and got["sp_rounds"] == 4 and got["sp_overlap"] is True and got["sp_selfcheck_rel"] <= 3e-3
    assert got["forwards_skipped"] == ref["forwards_skipped"] > 0 and got["skipped_steps"] == ref["skipped_steps"]
    for which in ("magcache", "nocache"):
        a, b = got["final_latent_probe"][which], ref["final_latent_probe"][which]
        assert abs(a["l2"] - b["l2"]) < 2e-3 * b["l2"] and max(abs(x - y) for x, y in zip(a["samples"], b["samples"])) < 3e-2 * b["rms"]
    k = got["kernels_live_rank0"]["classes"]
    # 3 live steps x 2 forwards x 30 layers: four gather rounds each (of the rank's own rows), ONE attention launch (a world of
    # one: the local shard is every key, the rounds have nothing left to attend)
    assert k["sp_wait"]["pairs"] == 6 * 30 * 4 and k["attn_self"]["pairs"] == 6 * 30
    w = got["sp_gather_wait"]
    assert w["waits_per_layer"] == 4 and 0.0 <= w["frac_of_forward_wall"] < 0.2
    assert got["roofline"]["bound"] == "mfma" and 0.3 < got["roofline"]["frac"] < 1.0 and "kernels_live" not in got
    # the sharded path costs a world of one little: the same kernels + the chain's merges and an all-gather of its own rows
    assert got["nocache_steps_per_s"] > 0.85 * ref["nocache_steps_per_s"]
