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


SMALL = ["--docs", "30000", "--steps", "3", "--warmup", "1", "--queries", "300", "--and3", "200", "--total-docs", "70000", "--c4-queries", "300",
         "--c5-queries", "120", "--query-chunk", "128", "--cpu-docs-all", "4000"]


def _run(extra, port, env_extra=None, n_gpus=1, expect_rc=0):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    if expect_rc:
        assert r.returncode != 0
        return r
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert KEYS <= set(out), sorted(KEYS - set(out))
    assert out["value"] > 0 and out["steps"] == 3 and out["warmup"] == 1 and out["n_gpus"] == n_gpus
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and rf["achieved"] > 0 and rf["peak"] == 8000.0
    # the whole build on the SURVEY 8(d) bytes, beside the dominant-kernel figure
    ba = rf["build_algorithmic"]
    assert 0 < rf["frac_build_algorithmic"] == ba["frac"] < rf["frac"] and ba["bytes_per_step_per_gpu"] > out["config"]["text_bytes_per_gpu"]
    return out


def _check_big(big, n_gpus=1):
    assert big["index"]["docs"] == 70000 and big["index"]["n_gpus"] == n_gpus and big["index"]["text_GB_per_s"] > 0
    for leg, nq in (("c4_and3", 300), ("c5_phrase", 120)):
        r = big[leg]
        assert 0 < r["queries"] <= nq and r["queries_per_s"] > 0 and r["calls"] >= 1
        rf = r["roofline"]
        assert rf["algorithmic_bytes"] == 2 * r["units_in"] + 4 * r["positions_out"] + 8 * r["pages_out"] and 0 < rf["frac"] < 1
    assert big["c5_phrase"]["pages_out"] >= big["c5_phrase"]["queries"]      # every sampled phrase occurs where it was taken from


def test_default_run_prints_the_contract_line():
    out = _run(["--cpu-docs", "2000"], 29731)
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert cb["all_cores"]["cores"] >= 2 and cb["all_cores"]["value"] > 0
    q = out["queries"]
    assert q["single_term"]["queries_per_s"] > 0 and q["and3"]["queries_per_s"] > 0 and q["and3"]["roofline"]["frac"] > 0
    assert q["and3"]["cpu_baseline"]["queries_per_s"] > 0
    assert out["builders_in_flight"]["builders"] == 2
    assert out["general_pages"]["GB_per_s"] > 0 and out["general_pages"]["ms_per_build_device"] > 0
    assert out["config"]["postings"] > 0 and out["config"]["workload"].startswith("configs[1]") and out["config"]["shard_only"] is False
    _check_big(out["configs234"])
    assert out["configs234"]["c4_and3"]["cpu_baseline"]["queries_per_s"] > 0


def test_gpus_2_without_two_gpus_fails_loudly():
    """`python bench.py --gpus 2` (how the driver calls it) must never measure one GPU and label it two."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    r = _run(["--gpus", "2", "--cpu-docs", "0"], 29741, expect_rc=1)
    assert "only 1 GPU" in (r.stderr + r.stdout)


def test_gpus_2_spawns_its_ranks_rehearsal():
    """--gpus 2 with no launcher: bench.py starts its two ranks as a child (here both on the one GPU, over gloo:
    DOCODO_BENCH_REHEARSAL) and forwards ONE line that says n_gpus 2: weak-scaled timed steps with the exchange + merge,
    then configs[2..4] strong-scaled: shard per rank -> balanced owners -> replica all-gather -> queries sharded by query id."""
    out = _run(["--gpus", "2", "--cpu-docs", "0", "--inflight", "0"], 29743, env_extra={"DOCODO_BENCH_REHEARSAL": "1"}, n_gpus=2)
    assert out["config"]["shard_only"] is True and out["config"]["world_size"] == 2 and out["config"]["backend"] == "gloo"
    assert "exchange" in out["config"] and out["exchange_merge_ms_per_step_rank0"] > 0
    _check_big(out["configs234"], 2)
    assert "2 replicas" in out["configs234"]["c4_and3"]["sharding"]


def test_lean_events_run():
    out = _run(["--cpu-docs", "0", "--queries", "0", "--inflight", "0", "--lean-events", "1", "--total-docs", "0"], 29735)
    assert "dominant kernel only" in out["events"] and out["roofline"]["avg_launch_ms"] > 0
    assert out["device_ms_per_step"]["total"] > 0          # (from the instrumented warm-up step)


@pytest.mark.parametrize("pipeline", [0, 1])
def test_exchange_step_with_one_rank(pipeline):
    out = _run(["--cpu-docs", "0", "--force-exchange", "--pipeline", str(pipeline)], 29732 + pipeline)
    assert out["config"]["shard_only"] is True and "shard-only" in out["config"]["workload"]
    assert out["exchange_merge_ms_per_step_rank0"] > 0
    assert ("worker" in out["config"]["exchange"] or "second host thread" in out["config"]["exchange"]) == bool(pipeline)
