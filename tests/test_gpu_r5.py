"""`-m gpu` parity tests, part 3 (round 5): option combinations the step loop must survive, the launcher contract of the drop-in
executable (ref/run_one_test:50: `${MPISTART} -np N ./exe ...`), direct borders (ref/comm.cpp:700-883 as one exchange)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import free_port
from test_gpu_parity import mm, rows_close

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_every_tile_read_form_runs_to_the_last_step(prec):
    """option tile_read 0..3 of the LJ full-list tile kernel (one reciprocal per pair / per four pairs, paired / separate LDS reads): the last step of a
    run carries finalIntegrate inside the force launch (fuse_final), which needs the (tile_read, FUSE = 2) instantiation — tile_read 0 had none (round-4
    advisor finding) and Integrate::run failed on its last step. Every form runs 47 steps in slices (each slice ends with such a step) and lands on the
    same thermo rows (the forms differ in rounding only: ref/force_lj.cpp:366-449 computes the same pair terms)."""
    rows = []
    for rd in (3, 2, 1, 0):
        s = mm().Sim(["-s", "10", "-n", "47", "--half_neigh", "0"], precision=prec)
        s.handle.set_option("tile_read", rd)
        s.initial()
        for c in (1, 6, 20, 20):
            s.run_steps(c)
        s.handle.force_compute(1)
        d = s.handle.download()
        rows.append((s.rows(), d["x"][:d["nlocal"]].copy()))
        s.close()
    for other in rows[1:]:
        assert np.allclose(rows[0][1], other[1], rtol=0, atol=1e-9 if prec == "dp" else 2e-3)
