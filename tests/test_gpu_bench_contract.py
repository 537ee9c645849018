"""bench.py contract on the GPU box: one JSON line with the required keys (small scene so it runs in seconds), and the
N > 1 code path (scene broadcast, weight broadcast, image sharding, pose gather, max-over-ranks timing) with two ranks
sharing the one GPU over gloo -- the driver runs the same path over RCCL at 2/4/8 GPUs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}


def _run(cmd, env=None, timeout=400):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]
    return json.loads(lines[0])


@pytest.mark.timeout(500)
def test_bench_single_gpu_json_contract():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run([sys.executable, "-W", "ignore", "bench.py", "--gaussians", "8000", "--steps", "2", "--warmup", "1", "--cpu-sample-rays", "100000"])
    assert REQUIRED <= set(d) and "cpu_baseline" in d
    assert d["metric"] == "poses/sec" and d["unit"] == "poses/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches"] >= 2 and r["avg_launch_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.timeout(500)
def test_bench_two_ranks_code_path():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run([sys.executable, "-W", "ignore", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
              "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2", "--gaussians", "8000", "--steps", "2", "--warmup", "1"],
             env={"SIXDGS_BENCH_BACKEND": "gloo", "SIXDGS_BENCH_FORCE_DEVICE": "0"})
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d       # baseline only on rank 0 at N = 1
    assert abs(d["value"] - 2 * 4 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-2 * d["value"]   # whole-job aggregate over both ranks
