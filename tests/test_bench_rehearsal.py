"""``bench.py`` itself at 2 / 4 / 8 ranks (gloo, CPU): the driver's multi-GPU launch line, the all-reduce MAX of the
step time, the rank-0-only JSON line and every ShardedPS option the bench exposes are exercised before the first
8-GPU run.  The numbers are meaningless (fp32 on a CPU, tiny images); the CONTRACT is what is checked."""
import json
import os
import subprocess
import sys

import pytest

from conftest import free_port

pytestmark = pytest.mark.slow
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(world, *extra, timeout=400):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--device", "cpu", "--batch", "2",
           "--image-size", "32"] + list(extra)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %r" % lines
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_contract_at_n_ranks(world):
    out = run_bench(world, "--wire-dtype", "bf16")
    assert out["n_gpus"] == world and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["global_batch"] == 2 * world
    assert out["config"]["parallelism"] == "dp%d+ps-shard%d" % (world, world)
    assert out["config"]["wire_dtype"] == "bfloat16"
    assert out["value"] > 0 and abs(out["value"] - 2 * world * 2 / (out["ms_per_step"] * 2 / 1e3)) < 0.05 * out["value"]
    assert out["vs_baseline"] == pytest.approx(out["value"] / {2: 422.0, 4: 852.0, 8: 1734.0}[world], rel=1e-2, abs=6e-4)   # rounded to 3 decimals


@pytest.mark.parametrize("extra", [("--push", "all_to_all"), ("--num-ps", "1"),
                                   ("--update-mode", "stale", "--staleness", "2"),
                                   ("--model", "bert", "--model-arg", "seq_len=32", "--wire-dtype", "bf16")])
def test_bench_options_at_2_ranks(extra):
    out = run_bench(2, *extra)
    assert out["n_gpus"] == 2 and out["value"] > 0
    if "--num-ps" in extra:
        assert out["config"]["parallelism"] == "dp2+ps-shard1" and out["config"]["pull"] == "broadcast"
