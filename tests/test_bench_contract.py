"""The driver's contract with bench.py: one JSON line on stdout with the agreed keys (GPU), and the
entry points the driver imports (CPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config"}


def test_graft_entry_exposes_build_and_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g

    assert callable(g.build) and callable(g.smoke)


def test_bench_refuses_multi_gpu_without_a_launcher():
    env = dict(os.environ, WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 2 and "torchrun" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("force_comm", ["0", "1"])
def test_bench_prints_exactly_one_json_line(force_comm):
    """force_comm=1 keeps a live 1-rank RCCL group: RCCL's start-up banner must not reach stdout."""
    env = dict(os.environ, DTF_FORCE_COMM=force_comm, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["dtype"] == "bf16" and d["scaling"] == "weak" and d["unit"] == "images/sec"
    assert d["config"]["global_batch"] == 256 and "resnet50" in d["config"]["model"]
    assert d["value"] > 1000 and abs(d["value"] - 256 * 1e3 / d["ms_per_step"]) / d["value"] < 0.01
    assert abs(d["vs_baseline"] - d["value"] / 219.0) < 0.01
