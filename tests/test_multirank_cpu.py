"""world_size-2 (and 4) gloo tests on CPU: the N>1 host path that exists without a GPU — Comm::setup geometry of
every rank, partner symmetry, and real messages through the transport along the swap pattern."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(nproc, argv, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py")] + argv
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_swap_pattern(world, port, tmp_path):
    out = str(tmp_path / "geo.json")
    r = launch(world, ["geometry", out], port)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    assert res["errors"] == [] and res["world"] == world
    assert res["procgrid"][0] * res["procgrid"][1] * res["procgrid"][2] == world
