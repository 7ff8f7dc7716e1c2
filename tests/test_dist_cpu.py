"""N > 1 plumbing on CPU: world_size-2 and -3 gloo runs of the exchange step (docodo_amd.dist.exchange) and of the replica
all-gather (docodo_amd.dist.allgather_export)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("world,port", [(2, 29613), (3, 29617)])
def test_exchange_and_allgather_gloo(world, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "helpers", "gloo_exchange_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert all("rank %d ok" % k in r.stdout for k in range(world))
