"""bench.py's own multi-GPU entry point, driven on CPU: `python bench.py --gpus 2` with no WORLD_SIZE must start two fresh ranks
(torch.distributed.run on 127.0.0.1), run the barrier / broadcast / max-over-ranks plumbing on gloo, and relay ONE JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_self_launch_world2_gloo():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "selftest", "--backend", "gloo",
                        "--steps", "4", "--warmup", "0", "--batch", "7"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["rccl_ranks"] == [0, 1] and len(out["per_rank_ms"]) == 2
    assert out["cpu_baseline"] is None and "N>1" in out["cpu_baseline_note"]
    assert out["config"]["items_per_rank"] == 4        # rank 0's share of 7 items
    assert out["value"] > 0


def test_launch_children_function_and_external_torchrun():
    sys.path.insert(0, ROOT)
    import bench
    rc, line = bench.launch_children(2, ["--gpus", "2", "--workload", "selftest", "--backend", "gloo", "--steps", "2"], extra_env=_env(), timeout=300)
    assert rc == 0 and line is not None
    assert json.loads(line)["rccl_ranks"] == [0, 1]


def test_single_rank_selftest_in_process():
    sys.path.insert(0, ROOT)
    import bench
    env_backup = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK") if k in os.environ}
    try:
        assert bench.main(["--gpus", "1", "--workload", "selftest", "--backend", "gloo", "--steps", "2"]) == 0
    finally:
        os.environ.update(env_backup)


def test_world_mismatch_is_an_error():
    env = _env()
    env.update({"WORLD_SIZE": "1", "RANK": "0"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "selftest", "--backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
