"""GPU: the bench.py contract the round driver depends on -- exactly ONE line on stdout, a JSON record with the agreed fields,
`roofline` and `cpu_baseline` objects, whatever --steps / --warmup say (a small workload keeps this to a few seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["timed_steps"] % 40 == 0 and d["timed_steps"] >= 200          # whole major-iteration periods, past the initial phase
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-3)
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], abs=1e-3) and 0.0 < rf["frac"] < 1.0
    if "PYTEST_XDIST_WORKER" not in os.environ:  # (two timings of the same run: not under the contention soak, where the load changes between them)
        assert rf["avg_launch_ms"] < d["ms_per_step"]
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu and cpu["unit"] == d["unit"]
    if "PYTEST_XDIST_WORKER" not in os.environ:  # (a rate comparison: not under the contention soak, where four processes share the GPU)
        assert d["value"] > cpu["value"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` needs no wrapper: it starts torch.distributed.run itself and relays rank 0's record.  With one
    GPU the launcher path is forced by --self-launch (one RCCL rank); asking for more GPUs than the box has ends, AFTER the
    launcher ran, in "N GPUs requested, V visible" and a non-zero exit code."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--self-launch", "--workload", "tiny", "--steps", "20",
                        "--warmup", "5", "--no-cpu-baseline", "--no-convergence-run"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_nranks"] == 1 and d["value"] > 0
    from cuopt_amd import capi
    visible = capi.device_count()
    if visible < 16:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(visible + 1), "--workload", "tiny"],
                           capture_output=True, text=True, cwd=ROOT, timeout=900)
        assert r.returncode != 0 and not r.stdout.strip()
        assert "%d GPUs requested, %d visible" % (visible + 1, visible) in r.stderr
