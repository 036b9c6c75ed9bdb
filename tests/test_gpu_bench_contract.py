"""bench.py's contract on the small shape (C1, seconds): the single-GPU JSON line carries every key the driver and the
judge read (roofline, cpu_baseline, kernels ...), and the N > 1 host code -- shard bounds, per-rank synthetic columns, the
sharded engine, MAX-over-ranks timing, one JSON line from rank 0, orderly teardown -- runs with two ranks sharing the test
box's GPU (PVAE_BENCH_BACKEND=gloo: a rehearsal, marked as such in `data`).  RCCL between devices needs a node."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_keys():
    p = subprocess.run([sys.executable, "bench.py", "--workload", "C1", "--steps", "12", "--warmup", "3"], cwd=ROOT,
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["value"] > 0 and d["vs_baseline"] is None
    assert abs(d["value"] - 12 * 32 / (d["ms_per_step"] * 12e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert np.isfinite(d["loss_timed_mean"])


def test_two_ranks_on_one_gpu_rehearsal():
    env = dict(os.environ, PVAE_BENCH_BACKEND="gloo")
    port = 29700 + int(np.random.default_rng().integers(0, 2000))
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--workload",
                        "C1", "--steps", "10", "--warmup", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong" and "REHEARSAL" in d["data"]
    assert "snp-shard x2" in d["config"]["parallelism"] and np.isfinite(d["loss_timed_mean"])
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
