"""The bench line the driver records (a 22 KB line gave `BENCH_r05.parsed: null`). The LAST stdout line must be a compact strict-JSON
object (< 6 KB) that still carries the contract's keys, `roofline` and `cpu_baseline` — and the ONLY thing on stdout; the full object goes to stderr (prefixed) and a file."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _strict(line):
    def no_const(name):
        raise ValueError(f"non-strict JSON constant {name}")
    return json.loads(line, parse_constant=no_const)


def test_compact_line_of_the_round5_object():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_final_bench.json")))   # (the round-5 object: a fixed input for the compactor)
    assert len(json.dumps(full)) > 20000                      # the object that could not be recorded
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert "\n" not in line and len(line.encode()) < 6144, len(line)
    c = _strict(line)
    for k in CONTRACT_KEYS:
        assert k in c, k
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"] and c["metric"] == full["metric"]
    assert "workload" in c["config"] and "model" not in c["config"]
    r = c["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_executed", "avg_launch_ms", "launches_per_step", "traffic", "held_clock_ghz",
              "frac_executed_at_held_clock"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    b = c["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in b, k
    for name in ("acoustic", "semantic_m", "semantic_s"):
        w = c["workloads"][name]
        assert w["value"] == full[name]["value"] and w["ms_per_step"] == full[name]["ms_per_step"] and w["token_checksum"] == full[name]["token_checksum"]
    assert c["workloads"]["acoustic_decode"]["ms_per_step"] == full["acoustic_decode"]["ms_per_step"]
    assert c["fallback_batches"] == 0 and c["rccl_ranks"] == [0] and c["per_rank_ms"] == [full["ms_per_step"]]
    assert len(c["files"]) == len(full["files"]["legs"])


def test_compact_line_sheds_optional_blocks_rather_than_exceed_the_limit():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_final_bench.json")))   # (the round-5 object: a fixed input for the compactor)
    full["files"]["legs"] = full["files"]["legs"] * 40       # a bloated optional block
    line = bench.compact_line(full)
    c = _strict(line)
    assert len(line) <= bench.COMPACT_LIMIT and "files" not in c and "roofline" in c and "cpu_baseline" in c


def test_non_finite_numbers_become_null():
    import bench
    out = bench._finite({"a": float("nan"), "b": [1.0, float("inf")], "c": {"d": -float("inf"), "e": 2}})
    assert out == {"a": None, "b": [1.0, None], "c": {"d": None, "e": 2}}


def test_stdout_is_the_compact_object_alone(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    detail = str(tmp_path / "detail.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "selftest", "--backend", "gloo", "--steps", "3",
                        "--detail-out", detail], env=dict(env, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                                   # nothing but the line: whatever the driver keeps of stdout, it keeps this
    last = _strict(lines[0])
    assert len(lines[0]) < 6144 and last["n_gpus"] == 2 and last["rccl_ranks"] == [0, 1] and "roofline" in last and "cpu_baseline" in last
    details = [ln for ln in p.stderr.splitlines() if ln.startswith("BENCH_DETAIL ")]
    assert len(details) == 1                                           # rank 0's full object, on stderr
    full = _strict(details[0][len("BENCH_DETAIL "):])
    assert full["value"] == last["value"]
    assert _strict(open(detail).read())["value"] == last["value"]     # ... and in the file


def test_full_line_flag_keeps_the_tools_format():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--workload", "selftest", "--backend", "gloo", "--steps", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and "breakdown" in _strict(lines[0])


def test_compact_line_never_raises_on_long_texts():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_final_bench.json")))
    full["config"]["workload"] = "w" * 9000
    full["dtype"] = "d" * 3000
    line = bench.compact_line(full)
    c = _strict(line)
    assert len(line) <= bench.COMPACT_LIMIT and c["value"] == full["value"] and c["roofline"]["frac"] == full["roofline"]["frac"] and c["cpu_baseline"]["cores"] == 16
