"""bench.py end to end at a small size: the default one-GPU run (build steps, two builders in flight, CPU baseline, query
leg) and the N > 1 step (exchange + merge, one after the other and on the worker thread) with one rank over the real
backend.  One JSON line with the contract's keys; a broken leg must not reach the driver."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def _run(extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--docs", "30000", "--steps", "3", "--warmup", "1", "--queries", "300",
                        "--and3", "200"] + extra, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert KEYS <= set(out), sorted(KEYS - set(out))
    assert out["value"] > 0 and out["steps"] == 3 and out["warmup"] == 1 and out["n_gpus"] == 1
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and rf["achieved"] > 0 and rf["peak"] == 8000.0
    return out


def test_default_run_prints_the_contract_line():
    out = _run(["--cpu-docs", "2000"], 29731)
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["value"] > 0
    assert out["queries"]["single_term"]["queries_per_s"] > 0 and out["queries"]["and3"]["queries_per_s"] > 0
    assert out["builders_in_flight"]["builders"] == 2
    assert out["config"]["postings"] > 0 and out["config"]["workload"].startswith("configs[1]")


def test_lean_events_run():
    out = _run(["--cpu-docs", "0", "--queries", "0", "--inflight", "0", "--lean-events", "1"], 29735)
    assert "dominant kernel only" in out["events"] and out["roofline"]["avg_launch_ms"] > 0
    assert out["device_ms_per_step"]["total"] > 0          # (from the instrumented warm-up step)


@pytest.mark.parametrize("pipeline", [0, 1])
def test_exchange_step_with_one_rank(pipeline):
    out = _run(["--cpu-docs", "0", "--force-exchange", "--pipeline", str(pipeline)], 29732 + pipeline)
    assert out["exchange_merge_ms_per_step_rank0"] > 0
    assert ("worker" in out["config"]["exchange"] or "second host thread" in out["config"]["exchange"]) == bool(pipeline)
